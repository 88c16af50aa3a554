// ccmp_kernels_flat.hip — the latency kernel of the reference-arithmetic path: one sample per 128-thread block,
// every residual evaluation of a Newton iteration in ONE round.  Canonical (bit-reproducible) rounding model:
// built -ffp-contract=off -DCCMP_USE_FMA like ccmp_kernels_fd.hip, and like it with -mllvm -disable-machine-licm.
// __launch_bounds__(128, 4): four waves per SIMD, eight blocks per CU (107-123 VGPRs, no scratch; with a budget of 256
// registers the max-ilp scheduler takes 181 and halves the occupancy).
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
// Then the min-norm solve runs on wave 0 with lane c < 14 owning column c of the 2 x 14 system: the three serial dot
// products of a Jacobi sweep are formed by every lane from an LDS copy of the rows (the oracle's order), the rotation,
// the step and the update of x touch only the lane's own column (3 operations instead of 84 per sweep).
#ifdef CCMP_FLAT_TIMING
// phase timing of thread 0 of block 0 (tools/time_phases.py; never defined in the product build)
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

// record of one sample in LDS (doubles): x 14 | sin,cos 28 | tool poses at x 24 | f(x) 2 | J as [column][row] 28 | flag
constexpr int fX = 0, fSC = 14, fEE = 42, fF = 66, fJ = 68, fV = 96, fRec = 98;
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

// Per-joint constants of both arms in chain order, 12 doubles per joint (offset 3, axis 3, axis products 6), so that a
// joint's constants are six 16-byte LDS reads that can be issued one joint ahead of their use.
constexpr int kStepDoubles = 12, kStepTab = 2 * 7 * kStepDoubles;

__device__ __forceinline__ void stage_step_table(const ccmp_consts &K, double *tab, int tid)
{
  for (int k = tid; k < kStepTab; k += 128) {
    const int a = k / (7 * kStepDoubles), r = k - a * 7 * kStepDoubles, i = r / kStepDoubles, c = r - i * kStepDoubles;
    tab[k] = c < 3 ? K.offset[a][i][c] : (c < 6 ? K.axis[a][i][c - 3] : K.aprod[a][i][c - 6]);
  }
}

// joints I..6 of the chain of arm W with compile-time joint indices (STOCK: the exact-zero structure of the uncalibrated
// Panda is known to the compiler, ccmp_kin.h); the constants of joint I+1 are read while joint I is computed
template <int W, bool STOCK, int I>
__device__ __forceinline__ void flat_chain_from(const double2 *tab, const double2 *sct, int j, double s, double c, const double2 *cur,
                                                double2 sc_cur, double *R, double *o)
{
  if constexpr (I < 7) {
    double2 nxt[6], sc_nxt = sc_cur;
    if constexpr (I < 6) {
#pragma unroll
      for (int k = 0; k < 6; k++) nxt[k] = tab[6 * (I + 1) + k];
      sc_nxt = sct[I + 1];
    }
    double Rn[9];
    const double si = (I == j) ? s : sc_cur.x;
    const double ci = (I == j) ? c : sc_cur.y;
    const double off[3] = {cur[0].x, cur[0].y, cur[1].x};
    const double ax[3] = {cur[1].y, cur[2].x, cur[2].y};
    const double ap[6] = {cur[3].x, cur[3].y, cur[4].x, cur[4].y, cur[5].x, cur[5].y};
    chain_step<I, STOCK>(off, ax, ap, si, ci, R, Rn, o);
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
    flat_chain_from<W, STOCK, I + 1>(tab, sct, j, s, c, nxt, sc_nxt, R, o);
  }
}

template <int W, bool STOCK>
__device__ __forceinline__ void flat_chain_and_residual(const ccmp_consts &K, const ccmp_consts &KC, const double *steptab,
                                                        double *rec, int lane, int j, bool head, double s, double c, double y,
                                                        unsigned long long &tprev)
{
  (void)tprev; // only the -DCCMP_FLAT_TIMING build ticks it
  // ---- B: the chain of arm W, joint j at (s, c), the others at x --------------------------------------------------
  double Tw[12];
  {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
    const double2 *tab = reinterpret_cast<const double2 *>(steptab + W * 7 * kStepDoubles);
    const double2 *sct = reinterpret_cast<const double2 *>(rec + fSC + 2 * W * 7);
    double2 cur[6];
#pragma unroll
    for (int k = 0; k < 6; k++) cur[k] = tab[k];
    flat_chain_from<W, STOCK, 0>(tab, sct, j, s, c, cur, sct[0], R, o);
    tool_pose_t<STOCK>(KC, W, R, o, &Tw[0], &Tw[9]);
    if (lane == kXLane) {
#pragma unroll
      for (int k = 0; k < 12; k++) rec[fEE + W * 12 + k] = Tw[k];
    }
  }
  __syncthreads();
  FLAT_TICK(1);
  // ---- C: residuals against the partner arm's pose at x; the x lane of wave 0 yields f(x) ------------------------
  double To[12], tt[2];
#pragma unroll
  for (int k = 0; k < 12; k++) To[k] = rec[fEE + (1 - W) * 12 + k];
  if (W == 0) chain_residual(K, &Tw[0], &Tw[9], &To[0], &To[9], tt, nullptr, nullptr);
  else chain_residual(K, &To[0], &To[9], &Tw[0], &Tw[9], tt, nullptr, nullptr);
  // ---- D: OMPL's stencil inside the wave.  m_s = (t1 - t2) / (y1[j] - y2[j]) on the plus lane of each pair (its minus
  // partner sits three lanes up the row), J[.][col] = 1.5 m1 - 0.6 m2 + 0.1 m3 on the column's first lane.  Lanes that own
  // no plus point compute on whatever their row neighbours hold and are not stored.
  const double t0m = dpp_f64<0x103>(tt[0]), t1m = dpp_f64<0x103>(tt[1]), ym = dpp_f64<0x103>(y);
  const double den = y - ym;
  const double m0 = (tt[0] - t0m) / den, m1 = (tt[1] - t1m) / den;
  const double m0b = dpp_f64<0x101>(m0), m0c = dpp_f64<0x102>(m0);
  const double m1b = dpp_f64<0x101>(m1), m1c = dpp_f64<0x102>(m1);
  if (head) {
    double2 col;
    col.x = CCMP_FMA(0.1, m0c, CCMP_FMA(-0.6, m0b, 1.5 * m0));
    col.y = CCMP_FMA(0.1, m1c, CCMP_FMA(-0.6, m1b, 1.5 * m1));
    *reinterpret_cast<double2 *>(rec + fJ + 2 * (W * 7 + j)) = col;
  } else if (W == 0 && lane == kXLane) {
    rec[fF] = tt[0];
    rec[fF + 1] = tt[1];
  }
}

// the 2 x 14 rows as every lane needs them for the serial dot products (the oracle's order): fourteen 16-byte reads
__device__ __forceinline__ void load_rows(const double *rec, double2 *P)
{
#pragma unroll
  for (int k = 0; k < 14; k++) P[k] = *reinterpret_cast<const double2 *>(rec + fJ + 2 * k);
}

template <bool STOCK>
__device__ __forceinline__ bool flat_newton(const ccmp_consts &K, const ccmp_consts &KC, const double *steptab, double *rec, int tid,
                                            int &iter, int &updates, double &norm1, double &norm2, int max_iter)
{
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform arm
  const int lane = tid & 63;
  const int row = lane >> 4, rl = lane & 15;
  const bool ev = row < 3 ? rl < 12 : rl < 6;                    // stencil evaluation
  const bool sc_lane = lane >= kScLane0 && lane < kScLane0 + 7;  // sincos of joint lane - kScLane0 of arm w
  const int hi6 = (row < 3 && rl >= 6) ? 1 : 0;
  const int j = ev ? 2 * row + hi6 : -1, pt = ev ? rl - 6 * hi6 : 0;
  const bool head = ev && pt == 0;
  const bool plus = pt < 3;
  const int nstep = (plus ? pt : pt - 3) + 1;
  unsigned long long tprev = __builtin_readcyclecounter();
  for (;;) {
    // ---- A: angles.  y1[j] += h / y2[j] -= h by sequential adds as OMPL does; one sincos per lane -----------
    double y = 0.0, s, c;
    if (ev) {
      const double xj = rec[fX + w * 7 + j];
      const double axj = ccmp_abs(xj);
      const double h = 1.4901161193847656e-08 * (axj >= 1 ? axj : 1);
      const double hh = plus ? h : -h;
      y = xj + hh;
      if (nstep >= 2) y = y + hh;
      if (nstep >= 3) y = y + hh;
    } else if (sc_lane) {
      y = rec[fX + w * 7 + lane - kScLane0];
    }
    ccmp_sincos(y, &s, &c);
    if (sc_lane) {
      rec[fSC + 2 * (w * 7 + lane - kScLane0)] = s;
      rec[fSC + 2 * (w * 7 + lane - kScLane0) + 1] = c;
    }
    // an arm's sines and cosines are written and read by the arm's own wave: no block barrier
    wave_lds_fence();
    FLAT_TICK(0);
    if (w == 0) flat_chain_and_residual<0, STOCK>(K, KC, steptab, rec, lane, j, head, s, c, y, tprev);
    else flat_chain_and_residual<1, STOCK>(K, KC, steptab, rec, lane, j, head, s, c, y, tprev);
    __syncthreads();
    FLAT_TICK(2);
    const double f0 = rec[fF], f1 = rec[fF + 1];
    // ---- loop condition of ConstraintFunction.h:68 (block-uniform) ------------------------------------------
    bool cont = false;
    {
      const bool c1 = f0 > K.tol_pos;
      norm1 = c1 ? 1.0 : 0.0;
      bool resid = c1;
      if (!c1) { norm2 = f1; resid = f1 > K.tol_rot; }
      if (resid) { cont = iter < max_iter; iter++; }
    }
    if (!cont) return (norm1 < K.tol_pos) && (norm2 < K.tol_rot);
    // ---- E on wave 0 alone: minimum-norm step (solve_minnorm's operations in its order, ccmp_solve.h), lane c < 14
    // owning column c.  A second wave repeating it would only take issue slots from other blocks; wave 1 waits at the
    // barrier below.
    if (w == 0) {
#ifdef CCMP_FLAT_SOLVE_PRIO
      __builtin_amdgcn_s_setprio(CCMP_FLAT_SOLVE_PRIO); // the serial solve is every block's critical path: first in line on its SIMD
#endif
      double2 P[14];
      load_rows(rec, P);
      const int me = lane < 14 ? lane : 13;
      double2 mine = *reinterpret_cast<const double2 *>(rec + fJ + 2 * me);
      double g0 = f0, g1 = f1, a, d, b;
#pragma unroll
      for (int sweep = 0; sweep < 2; sweep++) {
        a = 0; d = 0; b = 0;
#pragma unroll
        for (int k = 0; k < 14; k++) {
          a = CCMP_FMA(P[k].x, P[k].x, a);
          d = CCMP_FMA(P[k].y, P[k].y, d);
          b = CCMP_FMA(P[k].x, P[k].y, b);
        }
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
          if (lane < 14) *reinterpret_cast<double2 *>(rec + fJ + 2 * lane) = mine;
          wave_lds_fence();
          load_rows(rec, P);
        }
      }
      a = 0; d = 0;
#pragma unroll
      for (int k = 0; k < 14; k++) { a = CCMP_FMA(P[k].x, P[k].x, a); d = CCMP_FMA(P[k].y, P[k].y, d); }
      const double s0 = ccmp_sqrt(a), s1 = ccmp_sqrt(d);
      const double smax = s0 > s1 ? s0 : s1;
      double thr = smax * (2.0 * 2.220446049250313e-16);
      if (thr < 2.2250738585072014e-308) thr = 2.2250738585072014e-308;
      const double k0 = s0 > thr ? g0 / a : 0.0;
      const double k1 = s1 > thr ? g1 / d : 0.0;
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

// SRC 0: q_in, SRC 1: ambient sampler, SRC 2: straggler pool.  queue == nullptr: static striding (one block per
// sample launches need no queue reset).
template <int SRC, bool STOCK>
__global__ __launch_bounds__(128, CCMP_FLAT_MIN_WAVES) void project_fd_flat_kernel(
    const ccmp_consts K, const double *__restrict__ q_in, double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
    uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient, unsigned long long B, unsigned long long *queue,
    unsigned long long seed, unsigned long long first_index, const double *__restrict__ pool,
    const unsigned long long *__restrict__ pool_count, int wrap_output, unsigned int *done_flag, unsigned int done_seq,
    unsigned long long pool_records)
{
  __shared__ __attribute__((aligned(16))) double lds[fRec];
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ __attribute__((aligned(16))) double steptab[kStepTab];
  __shared__ unsigned long long ticket;
  const int tid = threadIdx.x;
  {
    const double *src = reinterpret_cast<const double *>(&K);
    for (int k = tid; k < kConstsDoubles; k += 128) ktab[k] = src[k];
  }
  stage_step_table(K, steptab, tid);
  __syncthreads();
  const ccmp_consts &KL = *reinterpret_cast<const ccmp_consts *>(ktab);
  double *rec = lds;
  // the pool is filled from both ends (ccmp_kernels_fd.hip): pool_count[0] records predicted long at the front — taken
  // first — and pool_count[5] others from the back (pool_records == 0: front only)
  const unsigned long long n_front = (SRC == 2) ? pool_count[0] : 0ull;
  const unsigned long long total = (SRC == 2) ? n_front + (pool_records ? pool_count[5] : 0ull) : B;
  unsigned long long t = blockIdx.x;

  for (;;) {
    if (queue) {
      if (tid == 0) ticket = atomicAdd(queue, 1ull);
      __syncthreads();
      t = ticket;
    }
    if (t >= total) break;
    unsigned long long idx;
    int iter = 0, updates = 0;
    double norm1 = 0.0, norm2 = 0.0;
    if (SRC == 2) {
      const double *ent = pool + (t < n_front ? t : pool_records - 1ull - (t - n_front)) * kPoolEntry;
      idx = (unsigned long long)__double_as_longlong(ent[14]);
      iter = __double2hiint(ent[15]);
      updates = __double2loint(ent[15]);
      norm1 = ent[16];
      norm2 = ent[17];
      if (tid < 14) rec[fX + tid] = ent[tid];
    } else {
      idx = t;
      if (tid < 14) {
        double v;
        if (SRC == 0) v = q_in[idx * 14 + tid];
        else {
          v = ambient_uniform(KL, seed, first_index + idx, tid);
          if (q_ambient) q_ambient[idx * 14 + tid] = v;
        }
        rec[fX + tid] = v;
      }
    }
    __syncthreads();
#ifdef CCMP_GEO_TRACE
    const int updates_in = updates;
    if (tid == 0 && t < 65536) { g_geo_trace[3 * t] = wall_clock64(); g_geo_trace[3 * t + 2] = ((unsigned long long)blockIdx.x << 32) | (unsigned)updates_in; }
#endif
    const bool conv = flat_newton<STOCK>(K, KL, steptab, rec, tid, iter, updates, norm1, norm2, K.max_iter);
#ifdef CCMP_GEO_TRACE
    if (tid == 0 && t < 65536) { g_geo_trace[3 * t + 1] = wall_clock64(); g_geo_trace[3 * t + 2] |= (unsigned long long)(unsigned)(updates - updates_in) << 16; }
#endif
    const bool jv = flat_joint_valid(KL, rec, tid);
    if (tid < 14) {
      const double v = rec[fX + tid];
      q_out[idx * 14 + tid] = wrap_output ? wrap_pi(v) : v;
    }
    if (tid == 0) {
      ok_out[idx] = (uint8_t)(jv && conv);
      if (iters_out) iters_out[idx] = (uint16_t)updates;
    }
    __syncthreads();
    if (!queue) t += gridDim.x;
  }
  // single-state calls through the host entry point (one block): the results sit in pinned host memory; publish a
  // sequence number behind them so that the host can poll a word instead of waiting for the stream's completion signal
  if (done_flag != nullptr && tid == 0) {
    __threadfence_system();
    __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------------------------------------
// geodesic_flat_kernel — jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:32-96), one
// 128-thread block per edge (from -> to): interpolate a step of delta towards `to` (KinematicChainSpace::interpolate,
// KinematicChain.h:145-171), project it (flat_newton), apply the reference's four break tests, record the state.
// The StateValidityChecker (MoveIt collision) stays on the host: the kernel runs as the reference does with
// interpolate == true and the host truncates the list at the first invalid state, which is what the reference's
// break would have produced.
//
// Round 3.  Blocks are persistent and take edges from an atomic ticket (queue != NULL), optionally through a processing
// order (long edges first, geodesic_order_kernel below): the launch ends on short edges instead of on whichever long
// one the dispatcher happened to start last.  An edge that fills its list stops (n_states = max_states + 1) and leaves
// what a continuation needs in carry_out — the running length `total` BEFORE the step that did not fit and the bound
// lambda * dist(from, to); a later call with carry_in resumes it from its last stored state (passed as `from`): `dist`
// is then the distance of that state to the target, recomputed from the same operands as the value the first call held,
// so first call + continuation produce the states, flags and counts of one uninterrupted traversal bit for bit.
constexpr int gPrev = fRec, gTo = fRec + 14, gRec = fRec + 28;


// RealVectorStateSpace::distance over the 14 joints (plain Euclidean, KinematicChainSpace does not override it),
// summed serially in the canonical order; every thread computes it from LDS.
__device__ __forceinline__ double lds_distance(const double *a, const double *b)
{
  double dist = 0.0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    const double diff = a[i] - b[i];
    dist = CCMP_FMA(diff, diff, dist);
  }
  return ccmp_sqrt(dist);
}

template <bool STOCK>
__global__ __launch_bounds__(128, CCMP_FLAT_MIN_WAVES) void geodesic_flat_kernel(
    const ccmp_consts K, const double delta, const double lambda, const double *__restrict__ from,
    const double *__restrict__ to, unsigned long long E, int max_states, double *__restrict__ states,
    int *__restrict__ n_states, uint8_t *__restrict__ ok_out, int *__restrict__ newton_iters, int check_target,
    unsigned long long *queue, const unsigned int *__restrict__ order, const double *__restrict__ carry_in,
    double *__restrict__ carry_out, int round_budget)
{
  __shared__ __attribute__((aligned(16))) double lds[gRec];
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ __attribute__((aligned(16))) double steptab[kStepTab];
  __shared__ unsigned long long ticket;
  const int tid = threadIdx.x;
  {
    const double *src = reinterpret_cast<const double *>(&K);
    for (int k = tid; k < kConstsDoubles; k += 128) ktab[k] = src[k];
  }
  stage_step_table(K, steptab, tid);
  __syncthreads();
  const ccmp_consts &KL = *reinterpret_cast<const ccmp_consts *>(ktab);
  double *rec = lds;
  const double pi = 3.14159265358979323846;

  unsigned long long tk = blockIdx.x;
  for (;;) {
    if (queue) {
      if (tid == 0) ticket = atomicAdd(queue, 1ull);
      __syncthreads();
      tk = ticket;
    }
    if (tk >= E) break;
    const unsigned long long t = order ? (unsigned long long)order[tk] : tk;
#ifdef CCMP_GEO_TRACE
    if (tid == 0 && t < 65536) { g_geo_trace[3 * t] = wall_clock64(); g_geo_trace[3 * t + 2] = ((unsigned long long)blockIdx.x << 32) | tk; }
#endif
    double *out = states + t * (unsigned long long)max_states * 14ull;
    if (tid < 14) {
      const double a = from[t * 14 + tid];
      rec[gPrev + tid] = a;
      rec[gTo + tid] = to[t * 14 + tid];
      if (max_states > 0) out[tid] = a; // geodesic->push_back(cloneState(from))
    }
    __syncthreads();
    int n = 1, its = 0, rounds = 0;
    bool suspended = false; // the edge used up the call's budget of Newton rounds: it stops between two states (ok = 2)
    bool fits = true; // false: an accepted state found the list full — the edge stops there and reports max_states + 1
    bool target_ok = true;
    if (check_target) {
      // ConstrainedMotionValidator::checkMotion (src/planner/stefanBiPRM.cpp:397-398): isSatisfied(s2) first —
      // function(to) through one evaluation pass of the Newton routine (iteration cap 0: no update), then
      // KinematicChainConstraint::isSatisfied's test (finite, f0 <= tol1, f1 <= tol2; ConstraintFunction.h:114-120)
      if (tid < 14) rec[fX + tid] = rec[gTo + tid];
      __syncthreads();
      int iter0 = 0, upd0 = 0;
      double n1 = 0.0, n2 = 0.0;
      (void)flat_newton<STOCK>(K, KL, steptab, rec, tid, iter0, upd0, n1, n2, 0);
      const double f0 = rec[fF], f1 = rec[fF + 1];
      target_ok = (f0 - f0 == 0.0) && (f1 - f1 == 0.0) && f0 <= K.tol_pos && f1 <= K.tol_rot;
      __syncthreads();
    }
    double dist = lds_distance(rec + gPrev, rec + gTo), total = 0.0, total_before = 0.0;
    double maxd = dist * lambda;
    // a continuation is in the middle of the reference's do-while: it re-enters on the loop's own condition
    // (dist >= delta) with the running length and the bound of the first call
    bool enter = dist > delta;
    if (carry_in) {
      total = carry_in[2 * t];
      maxd = carry_in[2 * t + 1];
      enter = dist >= delta;
    }
    if (target_ok && enter) {
      // Between two projections every thread does the reference's bookkeeping for itself, in ONE pass over the 14 joints
      // and without a barrier (round 3; before: jointValid through a ballot and two barriers, then the distances one
      // after the other — per state about as long as a Newton round): jointValid(x), step = |previous - x| and
      // newDist = |x - to| are accumulated side by side (two independent serial sums, the canonical order each), the
      // tests then run in the reference's order.  The joints' owners (tid < 14) keep x, previous and to in registers and
      // write the next interpolated state themselves: two block barriers per state instead of five.
      double x_own = 0.0, to_own = 0.0; // this thread's joint of the accepted state / of the target (tid < 14)
      if (tid < 14) { x_own = rec[gPrev + tid]; to_own = rec[gTo + tid]; }
      for (int guard = 0; guard < 1000000; guard++) { // the reference loop ends by itself; guard bounds a non-finite input
        if (tid < 14) { // WrapperStateSpace::interpolate(previous, to, delta_ / dist, scratch)
          const double tt = delta / dist;
          const double fr = x_own;
          double diff = to_own - fr, v;
          if (ccmp_abs(diff) <= pi) v = CCMP_FMA(diff, tt, fr);
          else {
            if (diff > 0.0) diff = 2.0 * pi - diff;
            else diff = -2.0 * pi - diff;
            v = CCMP_FMA(-diff, tt, fr);
            if (v > pi) v -= 2.0 * pi;
            else if (v < -pi) v += 2.0 * pi;
          }
          rec[fX + tid] = v;
          rec[gPrev + tid] = fr; // previous := the accepted state (unchanged on the first pass)
        }
        __syncthreads();
        int iter = 0, updates = 0;
        double norm1 = 0.0, norm2 = 0.0;
        const bool conv = flat_newton<STOCK>(K, KL, steptab, rec, tid, iter, updates, norm1, norm2, K.max_iter);
        its += updates;
        rounds += updates + 1;
        // flat_newton leaves through a block barrier behind which nobody writes x any more: every thread reads the final
        // iterate, previous and the target straight from LDS
        bool jv = true;
        double s_acc = 0.0, d_acc = 0.0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
          const double xi = rec[fX + i];
          const int jj = i < 7 ? i : i - 7;
          if (xi < K.lbe[jj]) jv = false; // KinematicChainConstraint::jointValid (ConstraintFunction.h:43-55)
          if (xi > K.ube[jj]) jv = false;
          const double ds = rec[gPrev + i] - xi, dd = xi - rec[gTo + i];
          s_acc = CCMP_FMA(ds, ds, s_acc); // distance(previous, scratch)
          d_acc = CCMP_FMA(dd, dd, d_acc); // distance(scratch, to)
        }
        if (!(conv && jv)) break;                        // not on manifold
        const double step = ccmp_sqrt(s_acc), newDist = ccmp_sqrt(d_acc);
        if (step > lambda * delta) break;                // deviated
        total_before = total;
        total += step;
        if (total > maxd) break;                         // wandered too far
        if (newDist >= dist) break;                      // no closer than before
        // an edge that creeps (hundreds of accepted states, each a hair closer: seen at 1 in 16384 near-neighbour edges,
        // 952 states) must not hold the whole launch: when the list is full the edge stops and says so
        // (its running length and Newton count go back to what they were before this state: a continuation projects it again)
        if (n >= max_states) { fits = false; n = max_states + 1; total = total_before; its -= updates; break; }
        dist = newDist;
        if (tid < 14) {
          x_own = rec[fX + tid];
          out[(unsigned long long)n * 14ull + tid] = x_own;
        }
        n++;
        if (!(dist >= delta)) break;
        // A call bounds the serial work it spends on one edge: past round_budget Newton rounds the edge stops HERE — between
        // two states, where the reference's do-while has just found dist >= delta — and reports ok = 2; a continuation
        // from its last stored state with carry_out goes on exactly where this one stops (nothing is projected twice).
        // 16 384 near-neighbour edges, lists of 16: everything but one edge is through after 1.46 ms, that one creeping
        // edge needs 545 rounds for its 15 states and held the launch until 2.1 ms (profiles/r03_extend_timeline.log).
        if (round_budget > 0 && rounds >= round_budget) { suspended = true; break; }
        __syncthreads(); // everybody has read x and previous: their owners may overwrite them (top of the loop)
      }
    }
    if (tid == 0) {
      n_states[t] = n;
      ok_out[t] = suspended ? (uint8_t)2 : (uint8_t)(target_ok && fits && dist <= delta);
      if (newton_iters) newton_iters[t] = its;
      if (carry_out) { carry_out[2 * t] = total; carry_out[2 * t + 1] = maxd; }
#ifdef CCMP_GEO_TRACE
      if (t < 65536) g_geo_trace[3 * t + 1] = wall_clock64();
#endif
    }
    __syncthreads();
    if (!queue) tk += gridDim.x;
  }
}

// Processing order for a large batch of edges: those longer than `long_dist` first (they need the most states, and the
// edges that creep — joint values on either side of the +-pi wrap — are among them), the rest behind, so that the launch
// ends on short edges.  One thread per edge; the long edges fill the order from the front, the others from the back
// (two atomic counters); the order inside a class is whatever the atomics give and changes no result.  The distance is a
// scheduling hint only: plain arithmetic, no claim on its bits.
__global__ void geodesic_order_kernel(const double *__restrict__ from, const double *__restrict__ to, unsigned long long E,
                                      double long_dist2, unsigned int *__restrict__ counters, unsigned int *__restrict__ order)
{
  const unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  double d2 = 0.0;
#pragma unroll
  for (int k = 0; k < 14; k++) {
    const double d = to[e * 14 + k] - from[e * 14 + k];
    d2 += d * d;
  }
  // ballot-aggregated atomics: one per wave and class
  const bool is_long = !(d2 <= long_dist2); // NaN counts as long
  const unsigned long long mL = __builtin_amdgcn_ballot_w64(is_long), mS = __builtin_amdgcn_ballot_w64(!is_long);
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned int baseL = 0, baseS = 0;
  if (lane == 0) {
    if (mL) baseL = atomicAdd(&counters[0], (unsigned int)__builtin_popcountll(mL));
    if (mS) baseS = atomicAdd(&counters[1], (unsigned int)__builtin_popcountll(mS));
  }
  baseL = __builtin_amdgcn_readfirstlane(baseL);
  baseS = __builtin_amdgcn_readfirstlane(baseS);
  if (is_long) order[baseL + (unsigned int)__builtin_popcountll(mL & below)] = (unsigned int)e;
  else order[(unsigned int)(E - 1) - (baseS + (unsigned int)__builtin_popcountll(mS & below))] = (unsigned int)e;
}

} // namespace

extern "C" {

// one sample per 128-thread block, all evaluations of an iteration in one round; queue_head may be NULL (static striding)
hipError_t ccmp_launch_project_flat(const ccmp_consts *K, int src, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue_head,
                                    unsigned long long seed, unsigned long long first, const double *pool,
                                    const unsigned long long *pool_count, int wrap_output, int nblocks, unsigned int *done_flag,
                                    unsigned int done_seq, size_t pool_records, hipStream_t st)
{
  if (nblocks != 1) done_flag = nullptr; // the completion word is written by the one block of a single-state call
#define CCMP_LAUNCH_FLAT(SRC, STOCK)                                                                                             \
  hipLaunchKernelGGL((project_fd_flat_kernel<SRC, STOCK>), dim3(nblocks), dim3(128), 0, st, *K, q_in, q_out, ok, iters, q_ambient, \
                     (unsigned long long)B, queue_head, seed, first, pool, pool_count, wrap_output, done_flag, done_seq, \
                     (unsigned long long)pool_records)
  if (src == 0) {
    if (K->stock) CCMP_LAUNCH_FLAT(0, true);
    else CCMP_LAUNCH_FLAT(0, false);
  } else if (src == 1) {
    if (K->stock) CCMP_LAUNCH_FLAT(1, true);
    else CCMP_LAUNCH_FLAT(1, false);
  } else {
    if (K->stock) CCMP_LAUNCH_FLAT(2, true);
    else CCMP_LAUNCH_FLAT(2, false);
  }
#undef CCMP_LAUNCH_FLAT
  return hipGetLastError();
}

hipError_t ccmp_launch_geodesic(const ccmp_consts *K, double delta, double lambda, const double *from, const double *to,
                                size_t E, int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters,
                                int check_target, int nblocks, unsigned long long *queue, const unsigned int *order,
                                const double *carry_in, double *carry_out, int round_budget, hipStream_t st)
{
  if (K->stock)
    hipLaunchKernelGGL(geodesic_flat_kernel<true>, dim3(nblocks), dim3(128), 0, st, *K, delta, lambda, from, to, (unsigned long long)E,
                       max_states, states, n_states, ok, newton_iters, check_target, queue, order, carry_in, carry_out, round_budget);
  else
    hipLaunchKernelGGL(geodesic_flat_kernel<false>, dim3(nblocks), dim3(128), 0, st, *K, delta, lambda, from, to, (unsigned long long)E,
                       max_states, states, n_states, ok, newton_iters, check_target, queue, order, carry_in, carry_out, round_budget);
  return hipGetLastError();
}

// counters: two zeroed words; order: E words
hipError_t ccmp_launch_geodesic_order(const double *from, const double *to, size_t E, double long_dist, unsigned int *counters,
                                      unsigned int *order, hipStream_t st)
{
  hipLaunchKernelGGL(geodesic_order_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, from, to, (unsigned long long)E,
                     long_dist * long_dist, counters, order);
  return hipGetLastError();
}

#ifdef CCMP_GEO_TRACE
hipError_t ccmp_debug_geo_trace(unsigned long long *out, size_t n_edges)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_geo_trace), 3 * n_edges * sizeof(unsigned long long));
}
#endif

#ifdef CCMP_FLAT_TIMING
hipError_t ccmp_debug_flat_timing(unsigned long long *out8, int reset)
{
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_flat_timing), 8 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_flat_timing), z, sizeof z);
  }
  return e;
}
#endif

} // extern "C"
