// ccmp_flat_newton.h — the Newton routine of the latency kernels: one sample per 128-thread block, every residual
// evaluation of an iteration in ONE round.  Included by ccmp_kernels_flat.hip (the projector's latency kernel) and by
// ccmp_kernels_geo.hip (the extend step), which are built with different register budgets and code-motion flags
// (build.py); everything here sits in the including unit's anonymous namespace.  Canonical (bit-reproducible) rounding
// model: both units are built -ffp-contract=off -DCCMP_USE_FMA like ccmp_kernels_fd.hip.
#ifndef CCMP_FLAT_NEWTON_H
#define CCMP_FLAT_NEWTON_H
// The FP64 literals of sine/cosine and arctangent come from an LDS table in these units (ccmp_detmath.h: CCMP_K): the
// compiler re-materialises a literal with two move instructions per use (machine LICM is off here, and hoisting them
// costs the registers that eight blocks per CU live on), one 16-byte LDS read brings two.  -DCCMP_FLAT_LITERALS: as before.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CCMP_FLAT_LITERALS)
static __shared__ __attribute__((aligned(16))) double ccmp_ktab[36]; // internal linkage: three translation units include this header
#define CCMP_K(i, v) (ccmp_ktab[i])
#define CCMP_KTAB_IN_LDS 1
#endif
#include "ccmp_fd_common.h"

using namespace ccmp;

namespace {

constexpr int kConstsDoubles = (int)((sizeof(ccmp_consts) + 7) / 8);
#ifdef CCMP_GEO_TRACE
// per-edge / per-sample timeline of one launch (tools/exp_r3.py geo_trace, flat_trace; never defined in the product build):
// start, end (100 MHz wall clock), block << 32 | ticket
__device__ unsigned long long g_geo_trace[3 * 65536];
#endif
#ifndef CCMP_FLAT_MIN_WAVES
#define CCMP_FLAT_MIN_WAVES 4 // waves per SIMD the register budget must allow (A/B: -DCCMP_FLAT_MIN_WAVES=2)
#endif

// ------------------------------------------------------------------------------------------------
// "Flat" Newton: one sample per 128-thread block, every evaluation of an iteration in ONE round.
// Wave w of the block owns arm w.  42 of its lanes own the 42 stencil points of that arm's 7 Jacobian columns and each
// runs the WHOLE 7-joint chain of the arm with its joint perturbed — the same operations in the same order as re-entering
// at a cached prefix frame, hence the same bits; one lane runs the arm's chain at x; seven lanes compute the
// sines/cosines of the arm's 7 joints.  function(x) and the 84 evaluations of jacobian(x) are therefore computed side by
// side instead of one after the other (the Jacobian of the final iterate is computed and dropped).  The arm is
// wave-uniform (no role selects); the per-joint constants come from an LDS step table read one joint ahead (scalar loads
// from the kernarg segment were 15 % slower here: one or two waves per SIMD cannot hide the scalar-cache round trips).
//
// Lane map (round 3): a column's six stencil points sit in ONE 16-lane row — row r < 3 holds columns 2r and 2r + 1 in
// its lanes 0..5 and 6..11 (point p = +h, +2h, +3h, -h, -2h, -3h), row 3 holds column 6 in lanes 0..5, the chain at x
// in lane 6 (wave lane 54) and the seven sine/cosine lanes in 7..13 (wave lanes 55..61) — so that the stencil is combined
// INSIDE the wave with row shifts (DPP): lane p < 3 takes its minus-point partner from lane p + 3 and forms
// m_s = (t1 - t2) / (y1 - y2), lane 0 of the column takes m_2, m_3 from lanes 1, 2 and forms 1.5 m1 - 0.6 m2 + 0.1 m3 —
// no trip through LDS, no extra barrier, and wave 1 combines its own arm's columns instead of waiting for wave 0.
// Then the min-norm solve runs on wave 0 with lane c < 14 owning column c of the 2 x 14 system: the three serial sums
// of a Jacobi sweep are formed in the oracle's order by three lanes — one sum each, from the operand pairs the columns
// are stored as — and handed round by readlane; the rotation, the step and the update of x touch only the lane's own
// column (3 operations instead of 84 per sweep).
//
// What a round costs a lone block is its NUMBER of instructions (one wavefront issues one per ~4.25 cycles, whatever it
// is: tools/ubench/), so the routine is written to keep moves, selects, exec-mask regions and branches out of the
// round: lanes find "their" joint, their (sin, cos) slot and every joint's (sin, cos) through per-lane LDS addresses
// computed once; literals of the elementary functions come from an LDS table (CCMP_K).
#ifdef CCMP_FLAT_TIMING
// phase timing of thread 0 of block 0 (tools/exp_r3.py phases; never defined in the product build)
__device__ unsigned long long g_flat_timing[8];
__device__ __forceinline__ void flat_tick(int k, unsigned long long &prev)
{
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned long long now = __builtin_readcyclecounter();
    g_flat_timing[k] += now - prev;
    prev = now;
  }
}
#define FLAT_TICK(k) flat_tick(k, tprev)
#else
#define FLAT_TICK(k) do { } while (0)
#endif

// record of one sample in LDS (doubles): x 14 | joint rotations at x 14 x 12 | tool poses at x 24 | f(x) 2 | J | flag |
// one joint-rotation slot per evaluation lane (its own perturbed joint).
// A joint-rotation slot: (sin, cos | the nine entries of Rot(axis, angle), ccmp_kin.h rot_sc | pad) — formed ONCE per round
// and joint by the lane that takes the joint's sine and cosine, instead of by every lane of the chain for every joint.
// J: six doubles per column c, (J0c, J0c | J1c, J1c | J0c, J1c) — the operand pairs of the three serial sums of a Jacobi
// sweep (sum J0c^2, sum J1c^2, sum J0c J1c), so that a lane forms ONE of them from one 16-byte read per column.
// CCMP_SUMS_IN_LANE (the build with machine LICM, which measured 4-7 % slower with the split sums): two doubles per
// column, every lane forms all three sums.
#ifdef CCMP_SUMS_IN_LANE
constexpr int kJCol = 2, kJPair = 0;
#else
constexpr int kJCol = 6, kJPair = 4;
#endif
constexpr int kRot = 14;  // doubles per joint-rotation slot: 11 used; 28 dwords apart, sixteen lanes' 16-byte accesses fall on distinct LDS banks
constexpr int kEvLanes = 42; // evaluation lanes per wavefront
constexpr int fX = 0, fSC = 14, fEE = fSC + 14 * kRot, fF = fEE + 24, fJ = fF + 2, fV = fJ + 14 * kJCol, fOwn = fV + 2,
              fDummy = fOwn + 2 * kEvLanes * kRot, fRec = fDummy + kRot;
static_assert(fSC % 2 == 0 && fEE % 2 == 0 && fJ % 2 == 0 && fOwn % 2 == 0 && fDummy % 2 == 0, "16-byte slots");

// column (v0, v1) of J into its slot
__device__ __forceinline__ void store_column(double *slot, double v0, double v1)
{
  double2 p;
#ifndef CCMP_SUMS_IN_LANE
  p.x = v0; p.y = v0;
  *reinterpret_cast<double2 *>(slot) = p;
  p.x = v1; p.y = v1;
  *reinterpret_cast<double2 *>(slot + 2) = p;
#endif
  p.x = v0; p.y = v1;
  *reinterpret_cast<double2 *>(slot + kJPair) = p;
}
// one of the three serial sums over the 14 columns, chosen by the lane's slot offset (0: row 0 squared, 2: row 1 squared,
// 4: the product), in column order — solve_minnorm's order (ccmp_solve.h)
__device__ __forceinline__ double column_sum(const double *jx)
{
  // all fourteen reads in flight before the first FMA (the scheduler otherwise — seen — serialises read, wait, FMA through
  // one register quad: fourteen exposed LDS round trips on the critical path of every round)
  double2 uv[14];
#pragma unroll
  for (int k = 0; k < 14; k++) uv[k] = *reinterpret_cast<const double2 *>(jx + kJCol * k);
  __builtin_amdgcn_sched_barrier(0);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 14; k++) acc = CCMP_FMA(uv[k].x, uv[k].y, acc);
  return acc;
}
// the value lane N of the row holds, in every lane of the row (the three sums are formed in lanes 0, 1, 2 of EVERY row, so
// all 64 lanes end up with the same numbers).  By readlane into a scalar pair, or — CCMP_SUM_BCAST_DPP, the build with
// machine LICM, whose scalar registers are spilled already — by a DPP row broadcast that stays in vector registers.
// all three in every lane (CCMP_SUMS_IN_LANE: the build with machine LICM measured 5 % slower with the split sums)
__device__ __forceinline__ void column_sums(const double *jbase, double &a, double &d, double &b)
{
  a = 0.0; d = 0.0; b = 0.0;
  double2 pp[14];
#pragma unroll
  for (int k = 0; k < 14; k++) pp[k] = *reinterpret_cast<const double2 *>(jbase + kJCol * k + kJPair);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 14; k++) {
    const double2 p = pp[k];
    a = CCMP_FMA(p.x, p.x, a);
    d = CCMP_FMA(p.y, p.y, d);
    b = CCMP_FMA(p.x, p.y, b);
  }
}
template <int N>
__device__ __forceinline__ double lane_value(double v)
{
#ifdef CCMP_SUM_BCAST_DPP
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + N, 0xf, 0xf, false); // row_newbcast:N
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
#else
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), N), __builtin_amdgcn_readlane(__double2loint(v), N));
#endif
}
constexpr int kXLane = 54, kScLane0 = 55; // chain at x; first sine/cosine lane

// value of the lane CTRL says: 0x100 + n = row_shl:n (lane i takes lane i + n of its 16-lane row)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_lds_fence()
{
  // LDS operations of one wave execute in order: a fence keeps the compiler (and the waitcnt insertion) honest, no s_barrier
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0); // and nothing is scheduled across (an s_barrier used to stand here: the max-ilp
                                     // scheduler otherwise hoists the next phase's loads to the top and spills)
}

// Per-joint constants of both arms in chain order, 14 doubles per joint: offset 3 + pad (read by the chain, one joint ahead
// of its use) | axis 3, axis products 6 + pad (read by the lane that forms the joint's rotation).
constexpr int kStepDoubles = 14, kStepAxis = 4, kStepTab = 2 * 7 * kStepDoubles;

__device__ __forceinline__ void stage_step_table(const ccmp_consts &K, double *tab, int tid)
{
#ifdef CCMP_KTAB_IN_LDS
  if (tid == 64) ccmp_fill_ktab(ccmp_ktab); // one thread of wave 1; the caller's barrier publishes it
#endif
  for (int k = tid; k < kStepTab; k += 128) {
    const int a = k / (7 * kStepDoubles), r = k - a * 7 * kStepDoubles, i = r / kStepDoubles, c = r - i * kStepDoubles;
    tab[k] = c < 3 ? K.offset[a][i][c] : (c < 4 ? 0.0 : (c < 7 ? K.axis[a][i][c - 4] : (c < 13 ? K.aprod[a][i][c - 7] : 0.0)));
  }
}

// joints I..6 of the chain of arm W with compile-time joint indices (STOCK: the exact-zero structure of the uncalibrated
// Panda is known to the compiler, ccmp_kin.h); the offset and the rotation slot of joint I+1 are read while joint I is
// computed.  A lane finds joint I's rotation at rec[slot]: the arm's table at x, or — for the lane's own perturbed joint —
// the lane's slot (an address select per joint).  A stock z joint needs the slot's (sin, cos) only.
template <int W, bool STOCK, int I>
__device__ __forceinline__ void flat_chain_from(const double2 *tab, const double *rec, int j, int own, const double2 *off_cur, const double2 *rot_cur,
                                                double *R, double *o)
{
  if constexpr (I < 7) {
    double2 off_nxt[2] = {off_cur[0], off_cur[1]}, rot_nxt[6];
#pragma unroll
    for (int k = 0; k < 6; k++) rot_nxt[k] = rot_cur[k];
    if constexpr (I < 6) {
      off_nxt[0] = tab[(kStepDoubles / 2) * (I + 1)];
      off_nxt[1] = tab[(kStepDoubles / 2) * (I + 1) + 1];
      const double2 *slot = reinterpret_cast<const double2 *>(rec + ((I + 1 == j) ? own : fSC + kRot * (W * 7 + I + 1)));
#pragma unroll
      for (int k = 0; k < 6; k++) rot_nxt[k] = slot[k];
    }
    double Rn[9];
    const double off[3] = {off_cur[0].x, off_cur[0].y, off_cur[1].x};
    mulvec_acc_nz<STOCK ? kStockOff[I] : 7>(R, off, o);
    if constexpr (STOCK && kStockZ[I] != 0) {
      mul_zrot(R, rot_cur[0].x, rot_cur[0].y, Rn);
    } else {
      const double Rj[9] = {rot_cur[1].x, rot_cur[1].y, rot_cur[2].x, rot_cur[2].y, rot_cur[3].x, rot_cur[3].y, rot_cur[4].x, rot_cur[4].y, rot_cur[5].x};
      mul33(R, Rj, Rn);
    }
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
    flat_chain_from<W, STOCK, I + 1>(tab, rec, j, own, off_nxt, rot_nxt, R, o);
  }
}

// ---- B: the chain of arm W, joint j at (s, c), the others at x; the x lane leaves the arm's pose at x in the record -------
template <int W, bool STOCK>
__device__ __forceinline__ void flat_chain_pose(const ccmp_consts &KC, const double *steptab, double *rec, int lane, int j, int own,
                                                double *Tw)
{
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
  const double2 *tab = reinterpret_cast<const double2 *>(steptab + W * 7 * kStepDoubles);
  double2 off0[2] = {tab[0], tab[1]}, rot0[6];
  {
    const double2 *slot = reinterpret_cast<const double2 *>(rec + ((j == 0) ? own : fSC + kRot * W * 7));
#pragma unroll
    for (int k = 0; k < 6; k++) rot0[k] = slot[k];
  }
  flat_chain_from<W, STOCK, 0>(tab, rec, j, own, off0, rot0, R, o);
  tool_pose_t<STOCK>(KC, W, R, o, &Tw[0], &Tw[9]);
  if (lane == kXLane) {
#pragma unroll
    for (int k = 0; k < 12; k++) rec[fEE + W * 12 + k] = Tw[k];
  }
}

// ---- C, D: residuals against the partner arm's pose at x (the x lane of wave 0 yields f(x)), then OMPL's stencil inside the
// wave: m_s = (t1 - t2) / (y1[j] - y2[j]) on the plus lane of each pair (its minus partner sits three lanes up the row),
// J[.][col] = 1.5 m1 - 0.6 m2 + 0.1 m3 on the column's first lane.  Lanes that own no plus point compute on whatever their row
// neighbours hold and are not stored.  Behind a block barrier (both arms' poses at x are in the record).
template <int W, bool STOCK>
__device__ __forceinline__ void flat_residual_stencil(const ccmp_consts &K, double *rec, int lane, int j, bool head, double y, const double *Tw)
{
  double To[12], tt[2];
#pragma unroll
  for (int k = 0; k < 12; k++) To[k] = rec[fEE + (1 - W) * 12 + k];
  if (W == 0) chain_residual(K, &Tw[0], &Tw[9], &To[0], &To[9], tt, nullptr, nullptr);
  else chain_residual(K, &To[0], &To[9], &Tw[0], &Tw[9], tt, nullptr, nullptr);
  const double t0m = dpp_f64<0x103>(tt[0]), t1m = dpp_f64<0x103>(tt[1]), ym = dpp_f64<0x103>(y);
  const double den = y - ym;
  const double m0 = (tt[0] - t0m) / den, m1 = (tt[1] - t1m) / den;
  const double m0b = dpp_f64<0x101>(m0), m0c = dpp_f64<0x102>(m0);
  const double m1b = dpp_f64<0x101>(m1), m1c = dpp_f64<0x102>(m1);
  if (head) {
    double2 col;
    col.x = CCMP_FMA(0.1, m0c, CCMP_FMA(-0.6, m0b, 1.5 * m0));
    col.y = CCMP_FMA(0.1, m1c, CCMP_FMA(-0.6, m1b, 1.5 * m1));
    store_column(rec + fJ + kJCol * (W * 7 + j), col.x, col.y);
  } else if (W == 0 && lane == kXLane) {
    rec[fF] = tt[0];
    rec[fF + 1] = tt[1];
  }
}

template <int W, bool STOCK>
__device__ __forceinline__ void flat_chain_and_residual(const ccmp_consts &K, const ccmp_consts &KC, const double *steptab,
                                                        double *rec, int lane, int j, int own, bool head, double y,
                                                        unsigned long long &tprev)
{
  (void)tprev; // only the -DCCMP_FLAT_TIMING build ticks it
  double Tw[12];
  flat_chain_pose<W, STOCK>(KC, steptab, rec, lane, j, own, Tw);
  __syncthreads();
  FLAT_TICK(1);
  flat_residual_stencil<W, STOCK>(K, rec, lane, j, head, y, Tw);
}

// what a lane of the wavefront of arm `w` does in a round, as LDS slots (doubles) and flags computed once: the joint value it
// starts from, where it stores the (sin, cos | rotation) it computes — evaluation lanes own a rotation slot each (42 per
// wavefront), the sine/cosine lanes fill the arm's table at x, the rest write a dummy slot — and where "its" joint's axis sits
// in the step table
struct FlatLane {
  int j, own, x_at, sc_to, ax_at, nstep;
  bool ev, head, plus;
};
__device__ __forceinline__ FlatLane flat_lane(int w, int lane)
{
  FlatLane L;
  const int row = lane >> 4, rl = lane & 15;
  L.ev = row < 3 ? rl < 12 : rl < 6;                               // stencil evaluation
  const bool sc_lane = lane >= kScLane0 && lane < kScLane0 + 7;  // sincos of joint lane - kScLane0 of arm w
  const int hi6 = (row < 3 && rl >= 6) ? 1 : 0;
  L.j = L.ev ? 2 * row + hi6 : -1;
  const int pt = L.ev ? rl - 6 * hi6 : 0;
  L.head = L.ev && pt == 0;
  L.plus = pt < 3;
  L.nstep = (L.plus ? pt : pt - 3) + 1;
  const int my_joint = L.ev ? L.j : (sc_lane ? lane - kScLane0 : 0);
  L.own = L.ev ? fOwn + kRot * (w * kEvLanes + row * 12 + rl) : fDummy;
  L.x_at = fX + w * 7 + my_joint;
  L.sc_to = sc_lane ? fSC + kRot * (w * 7 + lane - kScLane0) : L.own;
  L.ax_at = (w * 7 + my_joint) * kStepDoubles + kStepAxis; // axis + axis products of "my" joint in the step table
  return L;
}

// ---- A: angles.  y1[j] += h / y2[j] -= h by sequential adds as OMPL does; one sincos per lane, without a branch: every lane
// reads the joint it is concerned with (x_at), steps it (evaluation lanes) or not (sine/cosine lanes), and stores (sin, cos)
// and the joint's rotation (ccmp_kin.h rot_sc: the chain's operations, once) to its slot — the arm's table at x for the
// sine/cosine lanes, the lane's own slot otherwise.  Returns the angle the lane evaluated at.
__device__ __forceinline__ double flat_angles(const double *steptab, double *rec, const FlatLane &L)
{
  double s, c;
  const double xj = rec[L.x_at];
  const double axj = ccmp_abs(xj);
  const double h = 1.4901161193847656e-08 * (axj >= 1 ? axj : 1);
  const double hh = L.plus ? h : -h;
  const double y1 = xj + hh, y2 = y1 + hh, y3 = y2 + hh;
  const double ys = L.nstep == 1 ? y1 : (L.nstep == 2 ? y2 : y3);
  const double y = L.ev ? ys : xj;
  ccmp_sincos(y, &s, &c);
  const double2 *ac = reinterpret_cast<const double2 *>(steptab + L.ax_at);
  const double2 a01 = ac[0], a2p0 = ac[1], p12 = ac[2], p34 = ac[3], p5 = ac[4];
  const double axv[3] = {a01.x, a01.y, a2p0.x};
  const double apv[6] = {a2p0.y, p12.x, p12.y, p34.x, p34.y, p5.x};
  double Rj[9];
  rot_sc(axv, apv, s, c, Rj);
  double2 *slot = reinterpret_cast<double2 *>(rec + L.sc_to);
  double2 v;
  v.x = s; v.y = c; slot[0] = v;
  v.x = Rj[0]; v.y = Rj[1]; slot[1] = v;
  v.x = Rj[2]; v.y = Rj[3]; slot[2] = v;
  v.x = Rj[4]; v.y = Rj[5]; slot[3] = v;
  v.x = Rj[6]; v.y = Rj[7]; slot[4] = v;
  v.x = Rj[8]; v.y = 0.0; slot[5] = v;
  return y;
}

template <bool STOCK>
__device__ __forceinline__ bool flat_newton(const ccmp_consts &K, const ccmp_consts &KC, const double *steptab, double *rec, int tid,
                                            int &iter, int &updates, double &norm1, double &norm2, int max_iter)
{
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform arm
  const int lane = tid & 63;
  const int rl = lane & 15;
  const FlatLane L = flat_lane(w, lane);
  const int j = L.j, own = L.own;
  const bool head = L.head;
  unsigned long long tprev = __builtin_readcyclecounter();
  for (;;) {
    const double y = flat_angles(steptab, rec, L);
    // an arm's sines and cosines are written and read by the arm's own wave: no block barrier
    wave_lds_fence();
    FLAT_TICK(0);
    if (w == 0) flat_chain_and_residual<0, STOCK>(K, KC, steptab, rec, lane, j, own, head, y, tprev);
    else flat_chain_and_residual<1, STOCK>(K, KC, steptab, rec, lane, j, own, head, y, tprev);
    __syncthreads();
    FLAT_TICK(2);
    const double f0 = rec[fF], f1 = rec[fF + 1];
    // ---- loop condition of ConstraintFunction.h:68 (block-uniform) ------------------------------------------
    // (norm1 = f0 > tol1) || (norm2 = f1) > tol2: norm2 is assigned only when the first test fails; iter++ only when the
    // residual test holds — written as selects, one branch
    const bool c1 = f0 > K.tol_pos;
    const bool resid = c1 || (f1 > K.tol_rot);
    norm1 = c1 ? 1.0 : 0.0;
    norm2 = c1 ? norm2 : f1;
    const bool cont = resid && iter < max_iter;
    iter += resid ? 1 : 0;
    if (!cont) return (norm1 < K.tol_pos) && (norm2 < K.tol_rot);
    // ---- E on wave 0 alone: minimum-norm step (solve_minnorm's operations in its order, ccmp_solve.h), lane c < 14
    // owning column c.  A second wave repeating it would only take issue slots from other blocks; wave 1 waits at the
    // barrier below.
    if (w == 0) {
#ifdef CCMP_FLAT_SOLVE_PRIO
      __builtin_amdgcn_s_setprio(CCMP_FLAT_SOLVE_PRIO); // the serial solve is every block's critical path: first in line on its SIMD
#endif
      // lanes 0, 1 and 2+ of every 16-lane row form sum J0c^2, sum J1c^2 and sum J0c J1c; lane_value hands them round
#ifndef CCMP_SUMS_IN_LANE
      const double *jx = rec + fJ + 2 * (rl < 2 ? rl : 2);
#endif
      const int me = lane < 14 ? lane : 13;
      double2 mine = *reinterpret_cast<const double2 *>(rec + fJ + kJCol * me + kJPair);
      double g0 = f0, g1 = f1, a, d, b;
#pragma unroll
      for (int sweep = 0; sweep < 2; sweep++) {
#ifdef CCMP_SUMS_IN_LANE
        column_sums(rec + fJ, a, d, b);
#else
        const double acc = column_sum(jx);
        a = lane_value<0>(acc);
        d = lane_value<1>(acc);
        b = lane_value<2>(acc);
#endif
        if (b != 0.0) { // the same value in every lane
          const double zeta = (d - a) / (2.0 * b);
          double t = 1.0 / (ccmp_abs(zeta) + ccmp_sqrt(CCMP_FMA(zeta, zeta, 1.0)));
          if (zeta < 0.0) t = -t;
          const double cr = 1.0 / ccmp_sqrt(CCMP_FMA(t, t, 1.0));
          const double sr = cr * t;
          const double v0 = mine.x, v1 = mine.y;
          mine.x = CCMP_FMA(cr, v0, -(sr * v1));
          mine.y = CCMP_FMA(sr, v0, cr * v1);
          const double h0 = g0, h1 = g1;
          g0 = CCMP_FMA(cr, h0, -(sr * h1));
          g1 = CCMP_FMA(sr, h0, cr * h1);
          if (lane < 14) store_column(rec + fJ + kJCol * lane, mine.x, mine.y);
          wave_lds_fence();
        }
      }
#ifdef CCMP_SUMS_IN_LANE
      column_sums(rec + fJ, a, d, b);
      const double s0 = ccmp_sqrt(a), s1 = ccmp_sqrt(d);
      const double smax = s0 > s1 ? s0 : s1;
      double thr = smax * (2.0 * 2.220446049250313e-16);
      if (thr < 2.2250738585072014e-308) thr = 2.2250738585072014e-308;
      const double k0 = s0 > thr ? g0 / a : 0.0;
      const double k1 = s1 > thr ? g1 / d : 0.0;
#else
      // singular values and quotients of the two rows side by side: lane 0 takes row 0 (sqrt(a), g0 / a), lane 1 row 1
      const double acc = column_sum(jx);
      const double root = ccmp_sqrt(acc), quot = (rl == 0 ? g0 : g1) / acc;
      const double s0 = lane_value<0>(root), s1 = lane_value<1>(root);
      const double smax = s0 > s1 ? s0 : s1;
      double thr = smax * (2.0 * 2.220446049250313e-16);
      if (thr < 2.2250738585072014e-308) thr = 2.2250738585072014e-308;
      const double k0 = s0 > thr ? lane_value<0>(quot) : 0.0; // g0 / a where the row counts (ccmp_solve.h), 0 otherwise
      const double k1 = s1 > thr ? lane_value<1>(quot) : 0.0;
#endif
      const double dxm = CCMP_FMA(k1, mine.y, k0 * mine.x);
      if (lane < 14) rec[fX + lane] = CCMP_FMA(-K.step, dxm, rec[fX + lane]);
#ifdef CCMP_FLAT_SOLVE_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
#ifdef CCMP_FLAT_PROBE_DUMMY
    // probe (never in the product): wave 1 burns CCMP_FLAT_PROBE_DUMMY dependent-free FP64 instructions while it would wait
    // for the solve — issue load without any effect on the block's critical path.  If the loaded throughput falls, the
    // kernel is bound by issue slots; if it does not, by the latency of the rounds.
    else {
      // eight independent accumulators, multiplier and addend in registers: nothing but v_fma_f64, 4 cycles each
      double a0 = f0, a1 = f1, a2 = f0 + 1.0, a3 = f1 + 1.0, a4 = f0 + 2.0, a5 = f1 + 2.0, a6 = f0 + 3.0, a7 = f1 + 3.0;
      const double m = 1.0 + f0 * 1e-9, c0 = f1 * 1e-9;
#pragma unroll
      for (int k = 0; k < CCMP_FLAT_PROBE_DUMMY / 8; k++) {
        a0 = CCMP_FMA(a0, m, c0); a1 = CCMP_FMA(a1, m, c0); a2 = CCMP_FMA(a2, m, c0); a3 = CCMP_FMA(a3, m, c0);
        a4 = CCMP_FMA(a4, m, c0); a5 = CCMP_FMA(a5, m, c0); a6 = CCMP_FMA(a6, m, c0); a7 = CCMP_FMA(a7, m, c0);
      }
      if (((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) == 12345.678) rec[fV] = a0; // keeps the chain alive; never true
    }
#endif
    updates++;
    __syncthreads();
    FLAT_TICK(4);
  }
}

// jointValid(x) of the iterate in rec[fX..] (ConstraintFunction.h:43-55); every thread gets the result through LDS
__device__ __forceinline__ bool flat_joint_valid(const ccmp_consts &KL, double *rec, int tid)
{
  bool bad = false;
  if (tid < 14) {
    const double v = rec[fX + tid];
    const int jj = tid < 7 ? tid : tid - 7;
    if (v < KL.lbe[jj]) bad = true;
    if (v > KL.ube[jj]) bad = true;
  }
  const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull; // threads 0..13 sit in wave 0
  if (tid == 0) rec[fV] = any_bad ? 1.0 : 0.0; // its own slot: a slower wave may still be reading f(x) from fF
  __syncthreads();
  const bool jv = rec[fV] == 0.0;
  __syncthreads();
  return jv;
}


} // namespace
#endif // CCMP_FLAT_NEWTON_H
