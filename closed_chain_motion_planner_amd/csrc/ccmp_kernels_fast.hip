// ccmp_kernels_fast.hip — analytic-Jacobian projector for gfx950 (jacobian_mode = CCMP_JAC_ANALYTIC).
//
// Same Newton iteration, stopping rule and quirks as KinematicChainConstraint::project
// (include/closed_chain_motion_planner/base/constraints/ConstraintFunction.h:57-82), but the 2x14 Jacobian is the exact
// derivative of the residual — G(2x6) times the 6x14 closed-loop geometric Jacobian, whose single-arm half the reference
// itself carries as PandaModel::getJacobianMatrix (src/kinematics/panda_rbdl.cpp:9-22, never called there) — instead of
// OMPL's 84-evaluation finite-difference stencil, and the step is SURVEY.md §7.3's x -= 0.30 J^T (J J^T)^-1 f on the 2x2
// Gram matrix (the SVD-equivalent routine of the reference arithmetic only where the two rows are nearly parallel).
// Built in the canonical rounding model (-ffp-contract=off -DCCMP_USE_FMA): bit-identical to the CPU oracle run with
// ORC_JAC_ANALYTIC (oracle/ccmp_oracle.c: orc_jacobian_analytic + orc_solve_gram restate this file's operation order),
// which is how the mode is verified.  NOT bit-comparable with the REFERENCE arithmetic: the reference iteration amplifies
// 1e-8 Jacobian differences along the trajectory, so this mode lands on a different point of the same
// manifold for ~20 % of uniform samples.  Opt-in; the default mode is the FD-faithful kernel in ccmp_kernels_fd.hip.
//
// Layout (round 6): ONE SAMPLE PER LANE PAIR, 32 samples per wavefront.  The even lane carries arm 0, the odd lane arm 1:
// each runs its own 7-joint chain (sines, cosines, frames, joint axes z_i and origins o_i in the arm's base frame) and its
// seven Jacobian columns; the two tool poses cross inside the pair by DPP quad_perm broadcasts, the three Gram sums by a
// quad_perm swap.  Half the per-lane state of a one-sample-per-lane layout (round 5: 256 VGPRs + 198 AGPRs of spill
// space, one wavefront per SIMD, 34 % VALU issue): 142 registers, three wavefronts per SIMD, no scratch, no AGPR traffic.
// Lanes whose sample is done are refilled from ticket queues while their neighbours keep iterating (iteration counts
// spread 15..250); when the queues are dry a wavefront that is mostly empty hands its live samples over (x, index,
// counters, at the loop top) to a pool that the LATENCY kernel below — sixteen lanes per sample, 2.7 us per Newton round
// where this layout needs 4 — finishes; that kernel also takes small batches alone.  The extend step in this mode is a step
// loop around these projectors (end of the file).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccmp_kin.h"
#include "ccmp_solve.h"

using namespace ccmp;

namespace {

constexpr int kConstsDoubles = (int)((sizeof(ccmp_consts) + 7) / 8);
constexpr int kFastQueues = 64; // ticket words of the lane-pair kernel: one per lane of a wavefront (ccmp_ctx.h: kAnalyticWords)
constexpr int kPoolEntry = 18;  // hand-over record: x[14], idx, (iter, updates), norm1, norm2 (as ccmp_fd_common.h)

// value of the pair's even (arm 0) / odd (arm 1) lane in both lanes; the partner's value
__device__ __forceinline__ double pair_even(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xA0 /* quad_perm:[0,0,2,2] */, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xA0, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pair_odd(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xF5 /* quad_perm:[1,1,3,3] */, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xF5, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pair_swap(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// A value the optimiser cannot see through: what is derived from it inside the Newton loop is recomputed where it is used
// instead of being hoisted in front of the loop and held (or spilled) across it.
__device__ __forceinline__ int opaque(int v)
{
  asm volatile("" : "+v"(v));
  return v;
}
// KinematicChainSpace::enforceBounds (ccmp_kin.h: wrap_pi): fmod(q, 2 pi) is q itself when |q| < 2 pi — where a Newton
// iterate practically always is — and the library call's ~120 instructions run only for the lanes beyond (same bits).
__device__ __forceinline__ double wrap_pi_near(double q)
{
  const double pi = 3.14159265358979323846;
  double v = q;
  if (!(ccmp_abs(q) < 2.0 * pi)) v = __builtin_fmod(q, 2.0 * pi);
  if (v < -pi) v += 2.0 * pi;
  else if (v >= pi) v -= 2.0 * pi;
  return v;
}
__device__ __forceinline__ int pair_swap_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false); }

// Forward chain of this lane's arm in the arm's base frame, keeping every joint's axis z_i and origin o_i; joint indices are
// compile-time so that the STOCK instantiation skips the products with the stock Panda's exact zeros (ccmp_kin.h).  `ac` is
// the arm whose constants are read: the lane's own arm, or 0 for twin arms (one LDS address for the whole wavefront).
// The axes go to this lane's column of an LDS array (zs[k * 64], k = 3 i + component: consecutive lanes, consecutive words)
// — 42 more live registers across the chain otherwise; the origins stay in registers (with the stock structure joints 1
// and 5 share the origin of the joint before them, and joint 0's is a constant).
template <bool STOCK, int I>
__device__ __forceinline__ void chain_frames_from(const ccmp_consts &K, const int ac, const double *q, double *zs, double (*oj)[3],
                                                  double *R, double *o)
{
  if constexpr (I < 7) {
    asm volatile("" ::: "memory"); // the joint's constants are read from LDS here, not hoisted out of the Newton loop into registers
    // one joint after the other: the angle is made to depend on the frame the joint before left behind (no instruction
    // is emitted), or all seven sines and cosines are formed up front and the chain is interleaved across joints at
    // a cost of ~150 registers
    double qi = q[I];
    if constexpr (I > 0) asm volatile("" : "+v"(qi) : "v"(R[0]), "v"(R[4]), "v"(R[8]), "v"(o[0]), "v"(o[1]), "v"(o[2]));
    double s, c;
    ccmp_sincos(qi, &s, &c);
    mulvec_acc_nz<STOCK ? kStockOff[I] : 7>(R, K.offset[ac][I], o);
    const double *a = K.axis[ac][I];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      zs[(3 * I + k) * 64] = (STOCK && kStockZ[I]) ? R[3 * k + 2] : dot3(R[3 * k], a[0], R[3 * k + 1], a[1], R[3 * k + 2], a[2]);
      oj[I][k] = o[k];
    }
    double Rn[9];
    chain_rot<I, STOCK>(a, K.aprod[ac][I], s, c, R, Rn);
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
    chain_frames_from<STOCK, I + 1>(K, ac, q, zs, oj, R, o);
  }
}

// Sources: 0 q_in; 1 the ambient sampler (sampleUniform fused, output wrapped).
// STOCK: both arms carry the stock Panda's exact zeros; TWIN: and bit-identical chain constants with diag(+-1) base frames.
template <bool STOCK, bool TWIN>
__global__ __launch_bounds__(64, 3) void project_pair_kernel(const ccmp_consts K_arg, const int srcmode, const double *__restrict__ q_in,
                                                          double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
                                                          uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient,
                                                          unsigned long long B, unsigned long long *queue, unsigned long long seed,
                                                          unsigned long long first_index, double *__restrict__ pool_out,
                                                          unsigned long long *pool_out_count, const int dump_below)
{
  // the constants as an LDS copy read by broadcast: with compile-time joint indices the compiler would otherwise hoist
  // every scalar load of the kernarg copy out of the Newton loop and spill hundreds of SGPRs into VGPR lanes; the compiler
  // barriers in chain_frames_from keep the LDS reads at their joints
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ double zpark[21 * 64];
  {
    const double *srcp = reinterpret_cast<const double *>(&K_arg);
    for (int k = threadIdx.x; k < kConstsDoubles; k += 64) ktab[k] = srcp[k];
  }
  __syncthreads();
  const ccmp_consts &K = *reinterpret_cast<const ccmp_consts *>(ktab);
  const int src = srcmode & 15;          // where the samples come from
  const bool wrap = (srcmode >> 4) == 1; // the call is a fused sampleUniform: enforceBounds on the way out
  const int lane = threadIdx.x;
  const int arm = lane & 1;
  const int ac = TWIN ? 0 : arm;
  double x[7];
  unsigned long long idx = 0;
  int iter = 0, updates = 0;
  double norm1 = 0.0, norm2 = 0.0;
  bool active = false, drained = false;
  // Work distribution: one atomic per wavefront and refill event takes as many tickets as pairs are free.  kFastQueues
  // ticket words, each owning a contiguous slice of the batch (or of the pool), keep the atomics off a single address — a
  // same-address atomic costs ~12 ns chip-wide on this part (tools/ubench/atomic_rate.hip); a wavefront starts on its own
  // word and moves on to the next ones when that slice is used up.
  const unsigned long long total = B;
  static_assert(kFastQueues == 64, "one ticket word per lane");
  int qk = blockIdx.x % kFastQueues;
  const unsigned long long below_pair = (1ull << (lane & ~1)) - 1ull;
  constexpr unsigned long long kEven = 0x5555555555555555ull;

  for (;;) {
    // ---- hand-over --------------------------------------------------------------------------------------------------
    // When the tickets are used up and at most dump_below pairs of this wavefront still carry a sample, all of them leave and
    // the wavefront retires: the latency kernel launched behind finishes them, four to a wavefront.  State at the loop top:
    // function(x) and the loop test come next.
    if (pool_out != nullptr && drained) {
      const unsigned long long lv = __builtin_amdgcn_ballot_w64(active) & kEven;
      if (lv != 0ull && __builtin_popcountll(lv) <= dump_below) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(pool_out_count, (unsigned long long)__builtin_popcountll(lv));
        base = __shfl(base, 0);
        if (active) {
          const int a7 = 7 * (opaque(lane) & 1);
          double *ent = pool_out + (base + (unsigned long long)__builtin_popcountll(lv & below_pair)) * kPoolEntry + a7;
#pragma unroll
          for (int e = 0; e < 7; e++) ent[e] = x[e];
          if (a7 == 0) {
            ent[14] = __longlong_as_double((long long)idx);
            ent[15] = __hiloint2double(updates, iter);
            ent[16] = norm1;
            ent[17] = norm2;
          }
          active = false;
        }
      }
    }
    // ---- refill -------------------------------------------------------------------------------------------------------
    unsigned long long need = __builtin_amdgcn_ballot_w64(!active && !drained) & kEven;
    while (need != 0ull) {
      const int n = __builtin_popcountll(need);
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(queue + qk, (unsigned long long)n);
      base = __shfl(base, 0); // ticket of the first free pair, relative to the word's share
      const bool mine = (need >> (lane & ~1)) & 1ull;
      const unsigned long long lo = total * (unsigned long long)qk / kFastQueues;
      const unsigned long long hi = total * (unsigned long long)(qk + 1) / kFastQueues;
      const unsigned long long t = lo + base + (unsigned long long)__builtin_popcountll(need & below_pair);
      if (mine && t < hi) {
        active = true;
        const int a7 = 7 * (opaque(lane) & 1);
        idx = t;
        iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
#pragma unroll
        for (int e = 0; e < 7; e++) {
          if (src == 0) x[e] = q_in[idx * 14 + a7 + e];
          else {
            x[e] = ambient_uniform_at(K, seed, first_index + idx, a7 + e, e);
            if (q_ambient) q_ambient[idx * 14 + a7 + e] = x[e];
          }
        }
      }
      need = __builtin_amdgcn_ballot_w64(!active && !drained) & kEven;
      if (need != 0ull) {
        // this slice is used up.  One load brings all kFastQueues (= 64, one per lane) ticket words: go on with the next slice
        // that still has tickets; a word only grows, so when none has any left the batch is handed out for good.  (Trying
        // the words one by one cost every wavefront 64 serial atomic round trips at the end of a launch.)
        const unsigned long long taken = __hip_atomic_load(queue + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long lo_l = total * (unsigned long long)lane / kFastQueues, hi_l = total * (unsigned long long)(lane + 1) / kFastQueues;
        const unsigned long long avail = __builtin_amdgcn_ballot_w64(lo_l + taken < hi_l);
        if (avail == 0ull) { drained = true; break; }
        const int r = (qk + 1) & 63;
        const unsigned long long rot = r ? ((avail >> r) | (avail << (64 - r))) : avail;
        qk = (r + __builtin_ctzll(rot)) & 63;
      }
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;

    // ---- function(x): this lane's chain with its joint frames kept for the Jacobian, tool poses crossed in the pair -----
    double oj[7][3], Rw[9], pw[3];
    double *const zs = zpark + lane;
    {
      double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
      chain_frames_from<STOCK, 0>(K, ac, x, zs, oj, R, o);
      asm volatile("" ::: "memory");
      // tool_pose_t with this lane's base frame (ccmp_kin.h): TWIN bases are diag(+-1) — d_r * Rf[r][c] and
      // fma(d_r, pf[r], base_p[r]), to which the general product only adds exact zeros
      double pf[3] = {o[0], o[1], o[2]}, Rf[9];
      mulvec_acc_nz<STOCK ? kStockEe : 7>(R, K.ee[ac], pf);
      mul33(R, K.R_tool[ac], Rf);
      if (TWIN) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const double d = K.base_R[arm][4 * r];
#pragma unroll
          for (int c = 0; c < 3; c++) Rw[3 * r + c] = d * Rf[3 * r + c];
          pw[r] = CCMP_FMA(d, pf[r], K.base_p[arm][r]);
        }
      } else {
        mul33(K.base_R[arm], Rf, Rw);
        pw[0] = K.base_p[arm][0]; pw[1] = K.base_p[arm][1]; pw[2] = K.base_p[arm][2];
        mulvec_acc(K.base_R[arm], pf, pw);
      }
    }
    double R1[9], p1[3], R2[9], p2[3];
#pragma unroll
    for (int k = 0; k < 9; k++) { R1[k] = pair_even(Rw[k]); R2[k] = pair_odd(Rw[k]); }
#pragma unroll
    for (int k = 0; k < 3; k++) { p1[k] = pair_even(pw[k]); p2[k] = pair_odd(pw[k]); }
    double f[2], dq[4], pc[3];
    chain_residual(K, R1, p1, R2, p2, f, dq, pc);

    // ---- loop condition of ConstraintFunction.h:68, quirks included (both lanes of a pair decide alike) -------------------
    bool cont = false;
    if (active) {
      const bool c1 = f[0] > K.tol_pos;
      norm1 = c1 ? 1.0 : 0.0;
      bool resid = c1;
      if (!c1) { norm2 = f[1]; resid = f[1] > K.tol_rot; }
      if (resid) { cont = iter < K.max_iter; iter++; }
    }
    {
      const bool fin = active && !cont;
      int good = 1;
      const int a7 = 7 * (opaque(lane) & 1);
      if (fin) {
        double *row = q_out + idx * 14 + a7;
#pragma unroll
        for (int e = 0; e < 7; e++) {
          if (x[e] < K.lbe[e]) good = 0;
          if (x[e] > K.ube[e]) good = 0;
          row[e] = wrap ? wrap_pi_near(x[e]) : x[e];
        }
      }
      good &= pair_swap_i(good);
      if (fin && a7 == 0) {
        ok_out[idx] = (uint8_t)(good && (norm1 < K.tol_pos) && (norm2 < K.tol_rot));
        if (iters_out) iters_out[idx] = (uint16_t)updates;
      }
      if (fin) active = false;
    }
    if (__builtin_amdgcn_ballot_w64(cont) == 0ull) continue;

    // ---- analytic Jacobian, this lane's seven columns ----------------------------------------------------------------------
    // u = dp/|dp| (chain frame), n = axis of R_c R_0^T with w >= 0; both taken to the world frame through R_2, then into
    // this lane's arm base frame through base_R^T (oracle/ccmp_oracle.c: orc_jacobian_analytic).
    double J0[7], J1[7];
    {
      double u[3] = {0, 0, 0}, n[3] = {0, 0, 0};
      if (f[0] > 0.0) {
        const double inv = 1.0 / f[0];
#pragma unroll
        for (int k = 0; k < 3; k++) u[k] = (pc[k] - K.init_p[k]) * inv;
      }
      const double vn = ccmp_sqrt(dot3(dq[0], dq[0], dq[1], dq[1], dq[2], dq[2]));
      if (vn > 0.0) {
        const double sg = (dq[3] < 0.0 ? -1.0 : 1.0) / vn;
#pragma unroll
        for (int k = 0; k < 3; k++) n[k] = dq[k] * sg;
      }
      double aw[3], bw[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        aw[k] = dot3(R2[3 * k], u[0], R2[3 * k + 1], u[1], R2[3 * k + 2], u[2]);
        bw[k] = dot3(R2[3 * k], n[0], R2[3 * k + 1], n[1], R2[3 * k + 2], n[2]);
      }
      double al[3], bl[3], pl[3], dp[3];
#pragma unroll
      for (int k = 0; k < 3; k++) dp[k] = p1[k] - K.base_p[arm][k];
      mulTvec(K.base_R[arm], aw, al);
      mulTvec(K.base_R[arm], bw, bl);
      mulTvec(K.base_R[arm], dp, pl);
      asm volatile("" ::: "memory");
      const double sgn = (opaque(lane) & 1) ? -1.0 : 1.0; // + arm 0, - arm 1
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const double zi[3] = {zs[(3 * i) * 64], zs[(3 * i + 1) * 64], zs[(3 * i + 2) * 64]};
        const double r0 = pl[0] - oj[i][0], r1 = pl[1] - oj[i][1], r2 = pl[2] - oj[i][2];
        const double cx = CCMP_FMA(zi[1], r2, -(zi[2] * r1));
        const double cy = CCMP_FMA(zi[2], r0, -(zi[0] * r2));
        const double cz = CCMP_FMA(zi[0], r1, -(zi[1] * r0));
        J0[i] = sgn * dot3(al[0], cx, al[1], cy, al[2], cz);
        J1[i] = sgn * dot3(bl[0], zi[0], bl[1], zi[1], bl[2], zi[2]);
      }
    }
    // ---- Newton step on the Gram matrix (orc_solve_gram): per-arm partial sums, added across the pair ---------------------
    double dx[7];
    {
      double pa = 0.0, pd = 0.0, pb = 0.0;
#pragma unroll
      for (int j = 0; j < 7; j++) {
        pa = CCMP_FMA(J0[j], J0[j], pa);
        pd = CCMP_FMA(J1[j], J1[j], pd);
        pb = CCMP_FMA(J0[j], J1[j], pb);
      }
      // arm 0's sum + arm 1's sum: IEEE addition commutes, both lanes hold the same bits
      const double a = pa + pair_swap(pa), d = pd + pair_swap(pd), b = pb + pair_swap(pb);
      double y0, y1;
      const bool well = gram_coeffs(a, d, b, f[0], f[1], y0, y1);
#pragma unroll
      for (int j = 0; j < 7; j++) dx[j] = CCMP_FMA(y1, J1[j], y0 * J0[j]);
      if (__builtin_amdgcn_ballot_w64(cont && !well) != 0ull) {
        // nearly parallel rows (or a NaN): the reference arithmetic's SVD-equivalent solve on the full rows, both lanes alike
        double Jf[28], dxf[14];
#pragma unroll
        for (int j = 0; j < 7; j++) {
          Jf[j] = pair_even(J0[j]); Jf[7 + j] = pair_odd(J0[j]);
          Jf[14 + j] = pair_even(J1[j]); Jf[21 + j] = pair_odd(J1[j]);
        }
        solve_minnorm(Jf, f[0], f[1], dxf);
        if (!well) {
#pragma unroll
          for (int j = 0; j < 7; j++) dx[j] = (opaque(lane) & 1) ? dxf[7 + j] : dxf[j];
        }
      }
    }
    if (cont) {
#pragma unroll
      for (int e = 0; e < 7; e++) x[e] = CCMP_FMA(-K.step, dx[e], x[e]);
      updates++;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// project_row16_kernel — the analytic mode's LATENCY kernel: sixteen lanes (one DPP row) per sample, four samples per
// wavefront, one wavefront per block.  A wavefront that has its SIMD to itself pays ~4.6 cycles for ANY instruction
// (tools/ubench/), so a Newton round of the lane-pair kernel (~2 100 instructions) lasts ~5 us however few samples are
// left — 250 rounds of a sample that never converges: 1.2 ms.  Here a round is ~900 instructions on the critical path:
//   lane l < 14 owns joint l (arm l / 7): its angle, sine / cosine and joint rotation (rot_sc), its Jacobian column,
//               its component of the Newton step — the joint's constants stay in the lane's registers;
//   lane l < 6 also carries row l % 3 of arm l / 3's chain frame: per joint one row of R * Rot, its component of the
//               joint's axis z_i = R axis_i and of the joint's origin (the decomposition of the reference-arithmetic
//               throughput kernel's phase 1, ccmp_kernels_fd.hip), then one row of the tool pose;
//   all lanes   the residual, the probe vectors and the 2x2 Gram step (every lane needs their results).
// Joint rotations, axes / origins, the two tool poses and the Jacobian pass through the sample's LDS record; with one
// wavefront per block the LDS queue orders writes and reads, no barrier is needed.  Every value is produced by the same
// operations on the same operands as in the lane-pair kernel and in the oracle (general formulas: what the STOCK
// instantiations skip are products with exact zeros): bit-identical.  DIAG: both base frames are diag(+-1) (every shipped
// t_wb); otherwise the base-frame product is formed from the whole hand poses by every lane.
// Sources: 0 q_in, 1 ambient sampler, 2 the hand-over pool of the lane-pair kernel launched in front.
constexpr int qRJ = 0, qZO = 126, qT = 210, qJ = 234, qDX = 262, qRec = 277; // doubles per sample: Rot[14][9], (z, o)[14][6], poses[2][12], J[28], fallback dx[14]; odd stride

template <bool DIAG>
__global__ __launch_bounds__(64, 2) void project_row16_kernel(const ccmp_consts K_arg, const int srcmode, const double *__restrict__ q_in,
                                                              double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
                                                              uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient,
                                                              unsigned long long B, unsigned long long *queue, unsigned long long seed,
                                                              unsigned long long first_index, const double *__restrict__ pool_in,
                                                              const unsigned long long *__restrict__ pool_in_count)
{
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ double lds[4 * qRec];
  {
    const double *srcp = reinterpret_cast<const double *>(&K_arg);
    for (int k = threadIdx.x; k < kConstsDoubles; k += 64) ktab[k] = srcp[k];
  }
  __syncthreads();
  const ccmp_consts &K = *reinterpret_cast<const ccmp_consts *>(ktab);
  const int src = srcmode & 15;
  const bool wrap = (srcmode >> 4) == 1;
  const int lane = threadIdx.x, l = lane & 15;
  double *const rec = lds + (lane >> 4) * qRec;
  // joint role (lanes 14, 15 shadow joint 13 and never store)
  const bool jl = l < 14;
  const int lj = jl ? l : 13;
  const int aj = lj >= 7 ? 1 : 0, ij = lj - 7 * aj;
  double ax[3], ap[6];
#pragma unroll
  for (int k = 0; k < 3; k++) ax[k] = K.axis[aj][ij][k];
#pragma unroll
  for (int k = 0; k < 6; k++) ap[k] = K.aprod[aj][ij][k];
  const double lbe = K.lbe[ij], ube = K.ube[ij];
  const double sgn = aj ? -1.0 : 1.0;
  // chain role: lanes 6..15 shadow the rows of lanes 0..5 — the same operands, the same results, stored to the same words
  const int ac = (l / 3) & 1, rc = l % 3;
  double cee[3], cRt[9]; // this lane's arm: loop-invariant, kept in registers
#pragma unroll
  for (int k = 0; k < 3; k++) cee[k] = K.ee[ac][k];
#pragma unroll
  for (int k = 0; k < 9; k++) cRt[k] = K.R_tool[ac][k];
  const double cd = K.base_R[ac][4 * rc], cbp = K.base_p[ac][rc];
  const unsigned long long total = src == 2 ? *pool_in_count : B;

  double x = 0.0;
  unsigned long long idx = 0;
  int iter = 0, updates = 0;
  double norm1 = 0.0, norm2 = 0.0;
  bool active = false, drained = false;

  for (;;) {
    // ---- refill: rows without a sample take the next ticket ------------------------------------------------------------
    {
      const bool want = !active && !drained;
      unsigned long long t = 0;
      if (want && l == 0) t = atomicAdd(queue, 1ull);
      t = __shfl(t, lane & ~15);
      if (want) {
        if (t < total) {
          active = true;
          if (src == 2) {
            const double *ent = pool_in + t * kPoolEntry;
            idx = (unsigned long long)__double_as_longlong(ent[14]);
            iter = __double2hiint(ent[15]);
            updates = __double2loint(ent[15]);
            norm1 = ent[16];
            norm2 = ent[17];
            x = ent[lj];
          } else {
            idx = t;
            iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
            if (src == 0) x = q_in[idx * 14 + lj];
            else {
              x = ambient_uniform_at(K, seed, first_index + idx, lj, ij);
              if (q_ambient && jl) q_ambient[idx * 14 + lj] = x;
            }
          }
        } else drained = true;
      }
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;

    // ---- function(x).  Joint lanes: sine, cosine, joint rotation (lanes 14, 15 shadow lane 13: same values, same words) ------
    {
      double s, c, Rj[9];
      ccmp_sincos(x, &s, &c);
      rot_sc(ax, ap, s, c, Rj);
#pragma unroll
      for (int k = 0; k < 9; k++) rec[qRJ + 9 * lj + k] = Rj[k];
    }
    // ---- chain lanes: one row of the arm's frame through the seven joints, then one row of the hand pose ---------------------
    {
      double R0 = rc == 0 ? 1.0 : 0.0, R1 = rc == 1 ? 1.0 : 0.0, R2 = rc == 2 ? 1.0 : 0.0, o = 0.0;
      // the arm's joint rotations in two batches (four joints, then three): two LDS latencies instead of seven
#pragma unroll
      for (int h = 0; h < 2; h++) {
        constexpr int kFirst[2] = {0, 4}, kCount[2] = {4, 3};
        double Rj[4][9];
#pragma unroll
        for (int i = 0; i < kCount[h]; i++)
#pragma unroll
          for (int k = 0; k < 9; k++) Rj[i][k] = rec[qRJ + 9 * (ac * 7 + kFirst[h] + i) + k];
#pragma unroll
        for (int ii = 0; ii < kCount[h]; ii++) {
          const int i = kFirst[h] + ii;
          const double *off = K.offset[ac][i], *a = K.axis[ac][i];
          o = dot3acc(o, R0, off[0], R1, off[1], R2, off[2]);
          const double zr = dot3(R0, a[0], R1, a[1], R2, a[2]);
          rec[qZO + (ac * 7 + i) * 6 + rc] = zr;
          rec[qZO + (ac * 7 + i) * 6 + 3 + rc] = o;
          const double n0 = dot3(R0, Rj[ii][0], R1, Rj[ii][3], R2, Rj[ii][6]);
          const double n1 = dot3(R0, Rj[ii][1], R1, Rj[ii][4], R2, Rj[ii][7]);
          const double n2 = dot3(R0, Rj[ii][2], R1, Rj[ii][5], R2, Rj[ii][8]);
          R0 = n0; R1 = n1; R2 = n2;
        }
      }
      // the hand frame in the arm's base frame (getTranslation / getRotation), one row
      const double pf = dot3acc(o, R0, cee[0], R1, cee[1], R2, cee[2]);
      const double f0 = dot3(R0, cRt[0], R1, cRt[3], R2, cRt[6]);
      const double f1 = dot3(R0, cRt[1], R1, cRt[4], R2, cRt[7]);
      const double f2 = dot3(R0, cRt[2], R1, cRt[5], R2, cRt[8]);
      double *T = rec + qT + 12 * ac;
      if (DIAG) { // t_wb * T with t_wb.linear() = diag(d): d_r * Rf[r][c], fma(d_r, pf[r], base_p[r]) (the general product adds exact zeros)
        T[3 * rc] = cd * f0;
        T[3 * rc + 1] = cd * f1;
        T[3 * rc + 2] = cd * f2;
        T[9 + rc] = CCMP_FMA(cd, pf, cbp);
      } else {
        T[3 * rc] = f0;
        T[3 * rc + 1] = f1;
        T[3 * rc + 2] = f2;
        T[9 + rc] = pf;
      }
    }
    // ---- all lanes: the two world poses, the residual, the loop condition ------------------------------------------------------
    double T0[12], T1[12], f[2], dq[4], pc[3];
    if (DIAG) {
#pragma unroll
      for (int k = 0; k < 12; k++) { T0[k] = rec[qT + k]; T1[k] = rec[qT + 12 + k]; }
    } else {
#pragma unroll
      for (int arm = 0; arm < 2; arm++) {
        double Rf[9], pf[3], *Tw = arm ? T1 : T0;
#pragma unroll
        for (int k = 0; k < 9; k++) Rf[k] = rec[qT + 12 * arm + k];
#pragma unroll
        for (int k = 0; k < 3; k++) pf[k] = rec[qT + 12 * arm + 9 + k];
        mul33(K.base_R[arm], Rf, Tw);
        Tw[9] = K.base_p[arm][0]; Tw[10] = K.base_p[arm][1]; Tw[11] = K.base_p[arm][2];
        mulvec_acc(K.base_R[arm], pf, Tw + 9);
      }
    }
    chain_residual(K, &T0[0], &T0[9], &T1[0], &T1[9], f, dq, pc);
    bool cont = false;
    if (active) { // ConstraintFunction.h:68, quirks included; the sixteen lanes of a row decide alike
      const bool c1 = f[0] > K.tol_pos;
      norm1 = c1 ? 1.0 : 0.0;
      bool resid = c1;
      if (!c1) { norm2 = f[1]; resid = f[1] > K.tol_rot; }
      if (resid) { cont = iter < K.max_iter; iter++; }
    }
    {
      const bool fin = active && !cont;
      bool bad = false;
      if (fin && jl) {
        if (x < lbe) bad = true;
        if (x > ube) bad = true;
        q_out[idx * 14 + lj] = wrap ? wrap_pi_near(x) : x;
      }
      const unsigned long long badmask = __builtin_amdgcn_ballot_w64(bad);
      if (fin && l == 0) {
        const bool rbad = ((badmask >> (lane & ~15)) & 0xFFFFull) != 0ull;
        ok_out[idx] = (uint8_t)((!rbad) && (norm1 < K.tol_pos) && (norm2 < K.tol_rot));
        if (iters_out) iters_out[idx] = (uint16_t)updates;
      }
      if (fin) active = false;
    }
    if (__builtin_amdgcn_ballot_w64(cont) == 0ull) continue;
    // ---- analytic Jacobian: probes (every lane, its joint's arm), then this lane's column --------------------------------------
    double J0, J1;
    {
      double u[3] = {0, 0, 0}, n[3] = {0, 0, 0}, aw[3], bw[3];
      if (f[0] > 0.0) {
        const double inv = 1.0 / f[0];
#pragma unroll
        for (int k = 0; k < 3; k++) u[k] = (pc[k] - K.init_p[k]) * inv;
      }
      const double vn = ccmp_sqrt(dot3(dq[0], dq[0], dq[1], dq[1], dq[2], dq[2]));
      if (vn > 0.0) {
        const double sg = (dq[3] < 0.0 ? -1.0 : 1.0) / vn;
#pragma unroll
        for (int k = 0; k < 3; k++) n[k] = dq[k] * sg;
      }
#pragma unroll
      for (int k = 0; k < 3; k++) {
        aw[k] = dot3(T1[3 * k], u[0], T1[3 * k + 1], u[1], T1[3 * k + 2], u[2]);
        bw[k] = dot3(T1[3 * k], n[0], T1[3 * k + 1], n[1], T1[3 * k + 2], n[2]);
      }
      double al[3], bl[3], pl[3], dp[3];
#pragma unroll
      for (int k = 0; k < 3; k++) dp[k] = T0[9 + k] - K.base_p[aj][k];
      mulTvec(K.base_R[aj], aw, al);
      mulTvec(K.base_R[aj], bw, bl);
      mulTvec(K.base_R[aj], dp, pl);
      const double *zo = rec + qZO + 6 * lj;
      const double z0 = zo[0], z1 = zo[1], z2 = zo[2];
      const double r0 = pl[0] - zo[3], r1 = pl[1] - zo[4], r2 = pl[2] - zo[5];
      const double cx = CCMP_FMA(z1, r2, -(z2 * r1));
      const double cy = CCMP_FMA(z2, r0, -(z0 * r2));
      const double cz = CCMP_FMA(z0, r1, -(z1 * r0));
      J0 = sgn * dot3(al[0], cx, al[1], cy, al[2], cz);
      J1 = sgn * dot3(bl[0], z0, bl[1], z1, bl[2], z2);
      rec[qJ + lj] = J0;
      rec[qJ + 14 + lj] = J1;
    }
    // ---- Newton update on the Gram matrix (orc_solve_gram): every lane the three sums, its own component of the step ----------
    {
      double Jr[28], a, d, b, y0, y1;
#pragma unroll
      for (int k = 0; k < 28; k++) Jr[k] = rec[qJ + k];
      gram_sums(Jr, a, d, b);
      const bool well = gram_coeffs(a, d, b, f[0], f[1], y0, y1);
      double dx = CCMP_FMA(y1, J1, y0 * J0);
      if (__builtin_amdgcn_ballot_w64(cont && !well) != 0ull) { // nearly parallel rows (or a NaN): the SVD-equivalent solve, through LDS
        double dxf[14];
        solve_minnorm(Jr, f[0], f[1], dxf);
        if (l == 0) {
#pragma unroll
          for (int k = 0; k < 14; k++) rec[qDX + k] = dxf[k];
        }
        const double own = rec[qDX + lj];
        if (!well) dx = own;
      }
      if (cont) {
        x = CCMP_FMA(-K.step, dx, x);
        updates++;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// The extend step in analytic mode (round 6): jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:32-96)
// as a STEP LOOP around the batched projector — per step one launch that writes every live edge's interpolated state, the
// analytic-mode projection of all of them (the kernels above, whatever the policy picks for E states), one launch that does
// the reference's bookkeeping between two projections (the four break tests, the state list, the running lengths) — instead
// of a traversal kernel of its own.  max_states steps at most (an edge appends one state per step), no host synchronisation:
// an edge that has ended projects a state that is already on the manifold (the problem's start_joint: zero iterations).
// Operation for operation oracle/ccmp_oracle.c: orc_discrete_geodesic_ex with interpolate = true (the host truncates the list
// at the first state its StateValidityChecker rejects — where the reference's loop breaks), resumable through carry_in /
// carry_out like the reference-arithmetic kernel; a round budget is not enforced in this mode (ok is never 2).
struct geo_an_ws { // the step loop's per-edge state (device workspace of the context)
  double *prev, *scr, *dtm; // [E][14] last accepted state, [E][14] the state under projection, [E][3] dist / total / max
  uint16_t *itp;            // [E] iterations of the step's projection
  uint8_t *okp, *live;      // [E] its result; is the edge still under way
};

__device__ __forceinline__ double geo_distance(const double *a, const double *b)
{ // RealVectorStateSpace::distance: sqrt of the squares summed left to right (oracle: orc_distance)
  double d = 0.0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    const double diff = a[i] - b[i];
    d = CCMP_FMA(diff, diff, d);
  }
  return ccmp_sqrt(d);
}

__global__ void geo_an_init_kernel(const double delta, const double lambda, const double *__restrict__ from, const double *__restrict__ to,
                                   unsigned long long E, int max_states, double *__restrict__ states, int32_t *__restrict__ n_states,
                                   uint8_t *__restrict__ ok, int32_t *__restrict__ newton_iters, const double *__restrict__ carry_in,
                                   double *__restrict__ carry_out, const uint8_t *__restrict__ target_ok, geo_an_ws W)
{
  const unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  double a[14], b[14];
#pragma unroll
  for (int i = 0; i < 14; i++) { a[i] = from[e * 14 + i]; b[i] = to[e * 14 + i]; }
  if (max_states > 0) {
#pragma unroll
    for (int i = 0; i < 14; i++) states[e * (unsigned long long)max_states * 14 + i] = a[i];
  }
  const double dist = geo_distance(a, b);
  double total = 0.0, mx = dist * lambda;
  if (carry_in) { total = carry_in[2 * e]; mx = carry_in[2 * e + 1]; }
  if (carry_out) { carry_out[2 * e] = total; carry_out[2 * e + 1] = mx; }
  newton_iters[e] = 0;
  n_states[e] = 1;
  bool live = true;
  if (target_ok && !target_ok[e]) { ok[e] = 0; live = false; } // checkMotion: isSatisfied(to) failed — false, only `from` in the list
  else if (carry_in ? !(dist >= delta) : dist <= delta) { ok[e] = (uint8_t)(dist <= delta); live = false; }
  W.live[e] = live ? 1 : 0;
  W.dtm[3 * e] = dist; W.dtm[3 * e + 1] = total; W.dtm[3 * e + 2] = mx;
#pragma unroll
  for (int i = 0; i < 14; i++) W.prev[e * 14 + i] = a[i];
}

struct geo_an_start { double q[14]; };

// the next state of every live edge: WrapperStateSpace::interpolate(previous, to, delta / dist) (KinematicChain.h:145-171; oracle:
// orc_interpolate); an edge that has ended gets the problem's start state (f = 0 exactly: the projector leaves at once)
__global__ void geo_an_prepare_kernel(const double delta, const double *__restrict__ to, unsigned long long E, const geo_an_start S, geo_an_ws W)
{
  const unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const double pi = 3.14159265358979323846;
  if (!W.live[e]) {
#pragma unroll
    for (int i = 0; i < 14; i++) W.scr[e * 14 + i] = S.q[i];
    return;
  }
  const double t = delta / W.dtm[3 * e];
#pragma unroll
  for (int i = 0; i < 14; i++) {
    const double f = W.prev[e * 14 + i];
    double diff = to[e * 14 + i] - f, v;
    if (ccmp_abs(diff) <= pi) v = CCMP_FMA(diff, t, f);
    else {
      if (diff > 0.0) diff = 2.0 * pi - diff;
      else diff = -2.0 * pi - diff;
      v = CCMP_FMA(-diff, t, f);
      if (v > pi) v -= 2.0 * pi;
      else if (v < -pi) v += 2.0 * pi;
    }
    W.scr[e * 14 + i] = v;
  }
}

// between two projections: jy_ProjectedStateSpace.cpp:65-92 (the break tests, the list, the running lengths)
__global__ void geo_an_book_kernel(const double delta, const double lambda, const double *__restrict__ to, unsigned long long E, int max_states,
                                   double *__restrict__ states, int32_t *__restrict__ n_states, uint8_t *__restrict__ ok,
                                   int32_t *__restrict__ newton_iters, double *__restrict__ carry_out, geo_an_ws W)
{
  const unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E || !W.live[e]) return;
  double prev[14], sc[14], tg[14];
#pragma unroll
  for (int i = 0; i < 14; i++) { prev[i] = W.prev[e * 14 + i]; sc[i] = W.scr[e * 14 + i]; tg[i] = to[e * 14 + i]; }
  double dist = W.dtm[3 * e], total = W.dtm[3 * e + 1];
  const double mx = W.dtm[3 * e + 2];
  const int it = (int)W.itp[e];
  int its = newton_iters[e] + it, n = n_states[e];
  bool go_on = W.okp[e] != 0; // !project(scratch) -> break (interpolate = true: no validity test on the device)
  double step = 0.0;
  if (go_on) { step = geo_distance(prev, sc); go_on = !(step > lambda * delta); }
  const double total_before = total;
  if (go_on) { total += step; go_on = !(total > mx); }
  double new_dist = dist;
  if (go_on) { new_dist = geo_distance(sc, tg); go_on = !(new_dist >= dist); }
  if (go_on && n >= max_states) { // the accepted state finds the list full: the continuation projects it again
    n_states[e] = max_states + 1;
    newton_iters[e] = its - it;
    if (carry_out) { carry_out[2 * e] = total_before; carry_out[2 * e + 1] = mx; }
    ok[e] = 0;
    W.live[e] = 0;
    return;
  }
  if (go_on) {
    dist = new_dist;
#pragma unroll
    for (int i = 0; i < 14; i++) {
      W.prev[e * 14 + i] = sc[i];
      states[(e * (unsigned long long)max_states + (unsigned long long)n) * 14 + i] = sc[i];
    }
    n++;
    go_on = dist >= delta; // } while (dist >= tolerance)
  }
  n_states[e] = n;
  newton_iters[e] = its;
  W.dtm[3 * e] = dist; W.dtm[3 * e + 1] = total;
  if (!go_on) {
    if (carry_out) { carry_out[2 * e] = total; carry_out[2 * e + 1] = mx; }
    ok[e] = (uint8_t)(dist <= delta);
    W.live[e] = 0;
  }
}

} // namespace

extern "C" hipError_t ccmp_launch_clear_words(void *words, size_t n_u32, hipStream_t st); // ccmp_kernels_fd.hip

// One call of the analytic mode on one stream: the lane-pair kernel with pair_blocks wavefronts (0: none) and behind it, or
// alone, the latency kernel with latency_blocks wavefronts (0: none).  With both, a wavefront of the lane-pair kernel whose
// tickets are used up and that holds at most dump_below samples hands them over through `pool` (kPoolEntry doubles per
// sample; the fill count is read on the device: surplus wavefronts of the latency kernel exit at once).  queue: kFastQueues
// ticket words of the lane-pair kernel, the pool's fill count, the ticket word of the latency kernel.
extern "C" hipError_t ccmp_launch_project_analytic(const ccmp_consts *K, int mode, const double *q_in, double *q_out, uint8_t *ok,
                                                   uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue,
                                                   unsigned long long seed, unsigned long long first, int pair_blocks, int dump_below,
                                                   int latency_blocks, double *pool, hipStream_t st)
{
  if (pair_blocks <= 0 && latency_blocks <= 0) return hipErrorInvalidValue;
  hipError_t e = ccmp_launch_clear_words(queue, (kFastQueues + 2) * 2, st); // a kernel, so that a stream capture replays it
  if (e != hipSuccess) return e;
  unsigned long long *count = queue + kFastQueues, *lat_tickets = queue + kFastQueues + 1;
  const bool both = pair_blocks > 0 && latency_blocks > 0;
  if (pair_blocks > 0) {
#define CCMP_LAUNCH_PAIR(STOCK, TWIN)                                                                                                  \
  hipLaunchKernelGGL((project_pair_kernel<STOCK, TWIN>), dim3(pair_blocks), dim3(64), 0, st, *K, mode | (mode << 4), q_in, q_out, ok, iters, \
                     q_ambient, (unsigned long long)B, queue, seed, first, both ? pool : nullptr, count, dump_below)
    if (K->twin_arms) CCMP_LAUNCH_PAIR(true, true);
    else if (K->stock) CCMP_LAUNCH_PAIR(true, false);
    else CCMP_LAUNCH_PAIR(false, false);
#undef CCMP_LAUNCH_PAIR
  }
  if (latency_blocks > 0) {
    // srcmode: low nibble = where the samples come from (2: the pool), high nibble = the call's mode (a sample of a fused
    // sampleUniform is wrapped by whichever kernel finishes it)
    const int srcmode = (both ? 2 : mode) | (mode << 4);
    if (K->base_diag == 3)
      hipLaunchKernelGGL((project_row16_kernel<true>), dim3(latency_blocks), dim3(64), 0, st, *K, srcmode, q_in, q_out, ok, iters, q_ambient,
                         (unsigned long long)B, lat_tickets, seed, first, pool, count);
    else
      hipLaunchKernelGGL((project_row16_kernel<false>), dim3(latency_blocks), dim3(64), 0, st, *K, srcmode, q_in, q_out, ok, iters, q_ambient,
                         (unsigned long long)B, lat_tickets, seed, first, pool, count);
  }
  return hipGetLastError();
}

// the extend step's step loop in analytic mode: which = 0 init, 1 prepare, 2 bookkeeping (ws: seven device pointers of the
// context's workspace — prev, scr, dtm, itp, okp, live; start14: the problem's start_joint)
extern "C" hipError_t ccmp_launch_geodesic_analytic_step(int which, double delta, double lambda, const double *from, const double *to, size_t E,
                                                         int max_states, double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters,
                                                         const double *carry_in, double *carry_out, const uint8_t *target_ok, void *const *ws,
                                                         const double *start14, hipStream_t st)
{
  geo_an_ws W{(double *)ws[0], (double *)ws[1], (double *)ws[2], (uint16_t *)ws[3], (uint8_t *)ws[4], (uint8_t *)ws[5]};
  const dim3 grid((unsigned)((E + 127) / 128)), block(128);
  if (which == 0)
    hipLaunchKernelGGL(geo_an_init_kernel, grid, block, 0, st, delta, lambda, from, to, (unsigned long long)E, max_states, states, n_states, ok,
                       newton_iters, carry_in, carry_out, target_ok, W);
  else if (which == 1) {
    geo_an_start S;
    for (int i = 0; i < 14; i++) S.q[i] = start14[i];
    hipLaunchKernelGGL(geo_an_prepare_kernel, grid, block, 0, st, delta, to, (unsigned long long)E, S, W);
  } else
    hipLaunchKernelGGL(geo_an_book_kernel, grid, block, 0, st, delta, lambda, to, (unsigned long long)E, max_states, states, n_states, ok,
                       newton_iters, carry_out, W);
  return hipGetLastError();
}
