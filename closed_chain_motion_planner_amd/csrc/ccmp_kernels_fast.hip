// ccmp_kernels_fast.hip — analytic-Jacobian projector for gfx950 (jacobian_mode = CCMP_JAC_ANALYTIC).
//
// Same Newton iteration, stopping rule and quirks as KinematicChainConstraint::project
// (include/closed_chain_motion_planner/base/constraints/ConstraintFunction.h:57-82), but the 2x14
// Jacobian is the exact derivative of the residual instead of OMPL's 84-evaluation
// finite-difference stencil: ~35x less arithmetic per iteration.  Built in the canonical rounding model
// (-ffp-contract=off -DCCMP_USE_FMA, like the reference-arithmetic units): bit-identical to the CPU oracle run with
// ORC_JAC_ANALYTIC (oracle/ccmp_oracle.c: orc_jacobian_analytic restates this file's operation order), which is
// how the mode is verified.  NOT bit-comparable with the REFERENCE arithmetic: the reference iteration amplifies
// 1e-8 Jacobian differences along the trajectory, so this mode lands on a different
// point of the same manifold for ~20 % of uniform samples.  It is an opt-in fast mode; the default
// mode is the FD-faithful kernel in ccmp_kernels_fd.hip.
//
// Decomposition: one sample per lane, state in registers; lanes that finish a sample are refilled together from
// wave-level ticket queues while their neighbours keep iterating (iteration counts spread 15..250).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccmp_kin.h"
#include "ccmp_solve.h"

using namespace ccmp;

namespace {

constexpr int kConstsDoubles = (int)((sizeof(ccmp_consts) + 7) / 8);
constexpr int kFastQueues = 64; // queue words of the analytic kernel (ccmp_api.cpp allocates and clears as many)
constexpr int kPoolEntry = 18;  // hand-over record: x[14], idx, (iter, updates), norm1, norm2 (as ccmp_fd_common.h)

// Forward chain of one arm in its own base frame, keeping every joint's axis z_i and origin o_i; joint indices are
// compile-time so that the STOCK instantiation can skip the products with the stock Panda's exact zeros (ccmp_kin.h).
template <bool STOCK, int I>
__device__ __forceinline__ void chain_frames_from(const ccmp_consts &K, const int arm, const double *q, double (*z)[3], double (*oj)[3],
                                                  double *R, double *o)
{
  if constexpr (I < 7) {
    asm volatile("" ::: "memory"); // the joint's constants are read from LDS here, not hoisted out of the Newton loop into registers
    double s, c;
    ccmp_sincos(q[I], &s, &c);
    mulvec_acc_nz<STOCK ? kStockOff[I] : 7>(R, K.offset[arm][I], o);
    const double *a = K.axis[arm][I];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      z[I][k] = (STOCK && kStockZ[I]) ? R[3 * k + 2] : dot3(R[3 * k], a[0], R[3 * k + 1], a[1], R[3 * k + 2], a[2]);
      oj[I][k] = o[k];
    }
    double Rn[9];
    chain_rot<I, STOCK>(a, K.aprod[arm][I], s, c, R, Rn);
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
    chain_frames_from<STOCK, I + 1>(K, arm, q, z, oj, R, o);
  }
}
template <bool STOCK>
__device__ __forceinline__ void chain_frames(const ccmp_consts &K, const int arm, const double *q, double (*z)[3],
                                             double (*oj)[3], double *R, double *o)
{
  R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
  o[0] = 0; o[1] = 0; o[2] = 0;
  chain_frames_from<STOCK, 0>(K, arm, q, z, oj, R, o);
}

template <int MODE, bool STOCK>
__global__ __launch_bounds__(64) void project_fast_kernel(const ccmp_consts K_arg, const double *__restrict__ q_in,
                                                          double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
                                                          uint16_t *__restrict__ iters_out,
                                                          double *__restrict__ q_ambient, unsigned long long B,
                                                          unsigned long long *queue, unsigned long long seed,
                                                          unsigned long long first_index, double *__restrict__ pool,
                                                          unsigned long long *pool_count, int cap_iter,
                                                          const unsigned int *__restrict__ order,
                                                          const unsigned int *__restrict__ split_ptr)
{
  // the constants as an LDS copy read by broadcast: with compile-time joint indices the compiler would otherwise hoist
  // every scalar load of the kernarg copy out of the Newton loop and spill ~300 SGPRs into VGPR lanes (1170 v_readlane /
  // v_writelane of 4600 vector instructions); the compiler barriers in chain_frames_from keep the LDS reads at their joints (they also keep the
  // scheduler from interleaving all fourteen joints at once, which costs 512 registers and scratch)
  __shared__ double ktab[kConstsDoubles + 1];
  {
    const double *src = reinterpret_cast<const double *>(&K_arg);
    for (int k = threadIdx.x; k < kConstsDoubles; k += 64) ktab[k] = src[k];
  }
  __syncthreads();
  const ccmp_consts &K = *reinterpret_cast<const ccmp_consts *>(ktab);
  double x[14];
  unsigned long long idx = 0;
  int iter = 0, updates = 0;
  double norm1 = 0.0, norm2 = 0.0;
  bool active = false, drained = false;
  // Work distribution: iteration counts spread 15..250, so a fixed list of samples per lane leaves most lanes idle
  // while the unluckiest one works through its list (at 262144 samples: 23 % lane utilisation at 8 wavefronts per CU).
  // Lanes that need a sample are served together: one atomic per wavefront and refill event takes as many tickets as
  // lanes are free.  kFastQueues queue words, each owning a contiguous slice of the batch, keep the atomics off a
  // single address; a wavefront starts on its own word and moves on to the next ones when that slice is used up.
  const int lane = threadIdx.x;
  int qk = blockIdx.x % kFastQueues, tried = 0;
  // With a processing order (longest predicted first, ccmp_kernels_scout.hip) this kernel takes positions [split, B) of it
  // — the front goes to the six-lane kernel on a second stream — and the queue words own INTERLEAVED positions
  // (word k: split + k, split + k + 64, …), so that every wavefront, whichever word it starts on, begins with the longest
  // samples left and the launch ends on the shortest.
  const unsigned long long split = (order != nullptr && split_ptr != nullptr) ? (unsigned long long)*split_ptr : 0ull;
  const unsigned long long n_mine = B > split ? B - split : 0ull;

  for (;;) {
    // ---- hand-over: this kernel runs a sample on ONE lane at ~7 us per iteration whatever the occupancy, so the serial
    // chain of its longest sample bounds the launch (250 iterations: 1.85 ms).  A sample that has used cap_iter iterations
    // leaves its state (x, index, counters — at the loop top, where function(x) and the loop test come next) in the pool
    // for the six-lanes-per-sample kernel below, which iterates ~3x faster per sample.
    if (pool != nullptr && active && iter >= cap_iter) {
      const unsigned long long slot = atomicAdd(pool_count, 1ull);
      double *ent = pool + slot * kPoolEntry;
#pragma unroll
      for (int e = 0; e < 14; e++) ent[e] = x[e];
      ent[14] = __longlong_as_double((long long)idx);
      ent[15] = __hiloint2double(updates, iter);
      ent[16] = norm1;
      ent[17] = norm2;
      active = false; // the refill below gives the lane its next sample in this same pass
    }
    unsigned long long need = __builtin_amdgcn_ballot_w64(!active && !drained);
    while (need != 0ull) {
      const int n = __builtin_popcountll(need);
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(queue + qk, (unsigned long long)n);
      base = __shfl(base, 0);  // ticket of the first free lane, relative to the word's share
      const bool mine = (need >> lane) & 1ull;
      const unsigned long long j = base + (unsigned long long)__builtin_popcountll(need & ((1ull << lane) - 1ull));
      unsigned long long t, hi;
      if (order != nullptr) { // interleaved positions of the order
        t = (unsigned long long)qk + kFastQueues * j;
        hi = n_mine;
      } else { // a contiguous slice of the batch
        const unsigned long long lo = B * (unsigned long long)qk / kFastQueues;
        t = lo + j;
        hi = B * (unsigned long long)(qk + 1) / kFastQueues;
      }
      if (mine && t < hi) {
        idx = order != nullptr ? (unsigned long long)order[split + t] : t;
        active = true; iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
#pragma unroll
        for (int e = 0; e < 14; e++) {
          if (MODE == 0) x[e] = q_in[idx * 14 + e];
          else {
            x[e] = ambient_uniform(K_arg, seed, first_index + idx, e);
            if (q_ambient) q_ambient[idx * 14 + e] = x[e];
          }
        }
      }
      need = __builtin_amdgcn_ballot_w64(!active && !drained);
      if (need != 0ull) { // this slice is used up: try the next one; after a full round every slice is dry
        if (++tried >= kFastQueues) { drained = true; break; }
        qk = (qk + 1) % kFastQueues;
      }
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;

    // ---- function(x) with joint frames kept for the Jacobian --------------------------------
    double z[2][7][3], oj[2][7][3], Rw[2][9], pw[2][3];
#pragma unroll
    for (int arm = 0; arm < 2; arm++) {
      double R[9], o[3];
      chain_frames<STOCK>(K, arm, x + 7 * arm, z[arm], oj[arm], R, o);
      tool_pose_t<STOCK>(K, arm, R, o, Rw[arm], pw[arm]);
    }
    double f[2], dq[4], pc[3];
    chain_residual(K, Rw[0], pw[0], Rw[1], pw[1], f, dq, pc);

    bool cont = false;
    if (active) {
      const bool c1 = f[0] > K.tol_pos;
      norm1 = c1 ? 1.0 : 0.0;
      bool resid = c1;
      if (!c1) { norm2 = f[1]; resid = f[1] > K.tol_rot; }
      if (resid) { cont = iter < K.max_iter; iter++; }
    }
    if (active && !cont) {
      bool good = true;
#pragma unroll
      for (int e = 0; e < 14; e++) {
        if (x[e] < K.lbe[e % 7]) good = false;
        if (x[e] > K.ube[e % 7]) good = false;
        q_out[idx * 14 + e] = (MODE == 1) ? wrap_pi(x[e]) : x[e];
      }
      ok_out[idx] = (uint8_t)(good && (norm1 < K.tol_pos) && (norm2 < K.tol_rot));
      if (iters_out) iters_out[idx] = (uint16_t)updates;
      active = false;
    }
    if (__builtin_amdgcn_ballot_w64(cont) == 0ull) continue;

    // ---- analytic 2x14 Jacobian ----------------------------------------------------------------
    // u = dp/|dp| (chain frame), n = axis of R_c R_0^T with w >= 0; both taken to the world frame
    // through R_2, then into each arm's base frame through base_R^T.
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
      aw[k] = dot3(Rw[1][3 * k], u[0], Rw[1][3 * k + 1], u[1], Rw[1][3 * k + 2], u[2]);
      bw[k] = dot3(Rw[1][3 * k], n[0], Rw[1][3 * k + 1], n[1], Rw[1][3 * k + 2], n[2]);
    }
    double J[28];
#pragma unroll
    for (int arm = 0; arm < 2; arm++) {
      // probe vectors in this arm's base frame: a' = Rb^T a, b' = Rb^T b, p' = Rb^T (p1 - pb)
      double al[3], bl[3], pl[3], dp[3];
#pragma unroll
      for (int k = 0; k < 3; k++) dp[k] = pw[0][k] - K.base_p[arm][k];
      mulTvec(K.base_R[arm], aw, al);
      mulTvec(K.base_R[arm], bw, bl);
      mulTvec(K.base_R[arm], dp, pl);
      const double sgn = arm == 0 ? 1.0 : -1.0;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const double *zi = z[arm][i];
        const double r0 = pl[0] - oj[arm][i][0], r1 = pl[1] - oj[arm][i][1], r2 = pl[2] - oj[arm][i][2];
        const double cx = CCMP_FMA(zi[1], r2, -(zi[2] * r1));
        const double cy = CCMP_FMA(zi[2], r0, -(zi[0] * r2));
        const double cz = CCMP_FMA(zi[0], r1, -(zi[1] * r0));
        J[arm * 7 + i] = sgn * dot3(al[0], cx, al[1], cy, al[2], cz);
        J[14 + arm * 7 + i] = sgn * dot3(bl[0], zi[0], bl[1], zi[1], bl[2], zi[2]);
      }
    }
    double dx[14];
    solve_minnorm(J, f[0], f[1], dx);
    if (cont) {
#pragma unroll
      for (int e = 0; e < 14; e++) x[e] = CCMP_FMA(-K.step, dx[e], x[e]);
      updates++;
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// project_fast_rows_kernel — the analytic mode's LATENCY kernel: six lanes per sample, ten samples per wavefront.
// Lane r of a group carries row r % 3 of arm r / 3's chain frame (the decomposition of the reference-arithmetic
// throughput kernel's phase 1, ccmp_kernels_fd.hip): per joint the lane computes the joint's rotation matrix (13
// operations, all lanes alike) and ONE row of the frame product, its component of the joint axis z_i = R axis_i and of the
// joint origin; axes and origins of all fourteen joints, the two tool poses and the Jacobian pass through LDS.  An
// iteration is ~1.35 k instructions per lane instead of ~3.6 k (2.3 us instead of 7.4 us when a wave has its SIMD to
// itself), at 2.5x the SIMD-cycles per sample-iteration: used for the samples the one-lane kernel hands over, and alone for
// small batches.  Every value is produced by the same operations on the same operands as in the one-lane kernel and in
// the oracle (orc_jacobian_analytic): bit-identical.  Twin stock arms and diag(+-1) base frames only (K.twin_arms).
// SRC 0: q_in; SRC 1: ambient sampler; SRC 2: the hand-over pool.
constexpr int rGroup = 6, rGroups = 10;
constexpr int rX = 0, rSC = 14, rZO = 42, rT = 126, rJ = 150, rRec = 179; // doubles per group: x, sin/cos, (z, o)[2][7], poses, J; odd stride

template <int I>
__device__ __forceinline__ void rows_chain_from(const ccmp_consts &K, double *rec, const double *sc, int arm, int row, bool live,
                                                double &R0, double &R1, double &R2, double &o)
{
  if constexpr (I < 7) {
    asm volatile("" ::: "memory");
    constexpr int NZ = kStockOff[I];
    if (NZ & 1) o = CCMP_FMA(R0, K.offset[0][I][0], o);
    if (NZ & 2) o = CCMP_FMA(R1, K.offset[0][I][1], o);
    if (NZ & 4) o = CCMP_FMA(R2, K.offset[0][I][2], o);
    const double *a = K.axis[0][I];
    // this lane's component of z_i = R axis_i (a stock z joint: the third column of R) and of the joint origin
    const double zr = kStockZ[I] ? R2 : dot3(R0, a[0], R1, a[1], R2, a[2]);
    if (live) {
      rec[rZO + (arm * 7 + I) * 6 + row] = zr;
      rec[rZO + (arm * 7 + I) * 6 + 3 + row] = o;
    }
    const double s = sc[2 * I], c = sc[2 * I + 1];
    double n0, n1, n2;
    if constexpr (kStockZ[I] != 0) { // mul_zrot, one row
      const double t = 1.0 - c;
      const double w = t + c;
      const double ns = -s;
      n0 = CCMP_FMA(R1, s, R0 * c);
      n1 = CCMP_FMA(R1, c, R0 * ns);
      n2 = R2 * w;
    } else { // rot_sc + one row of mul33
      double Rj[9];
      rot_sc(a, K.aprod[0][I], s, c, Rj);
      n0 = dot3(R0, Rj[0], R1, Rj[3], R2, Rj[6]);
      n1 = dot3(R0, Rj[1], R1, Rj[4], R2, Rj[7]);
      n2 = dot3(R0, Rj[2], R1, Rj[5], R2, Rj[8]);
    }
    R0 = n0; R1 = n1; R2 = n2;
    rows_chain_from<I + 1>(K, rec, sc, arm, row, live, R0, R1, R2, o);
  }
}

template <int SRC>
__global__ __launch_bounds__(64, 2) void project_fast_rows_kernel(
    const ccmp_consts K_arg, const double *__restrict__ q_in, double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
    uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient, unsigned long long B, unsigned long long *queue,
    unsigned long long seed, unsigned long long first_index, const double *__restrict__ pool,
    const unsigned long long *__restrict__ pool_count, int wrap_output, const unsigned int *__restrict__ order,
    const unsigned int *__restrict__ split_ptr)
{
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ double lds[rGroups * rRec];
  {
    const double *src = reinterpret_cast<const double *>(&K_arg);
    for (int k = threadIdx.x; k < kConstsDoubles; k += 64) ktab[k] = src[k];
  }
  __syncthreads();
  const ccmp_consts &K = *reinterpret_cast<const ccmp_consts *>(ktab);
  const int lane = threadIdx.x;
  const int g = lane / rGroup, r = lane - rGroup * g;
  const bool live = g < rGroups;
  const int leader = live ? rGroup * g : 0;
  double *rec = lds + (live ? g : 0) * rRec; // idle lanes alias group 0 for reads, never write
  const int arm = r < 3 ? 0 : 1, row = r < 3 ? r : r - 3;
  const double d_lane = K.base_R[arm][4 * row], bp_lane = K.base_p[arm][row];
  // SRC 0 / 1 with a processing order: the first *split_ptr positions of it (the samples predicted longest)
  const unsigned long long total = (SRC == 2) ? *pool_count : ((order != nullptr && split_ptr != nullptr) ? (unsigned long long)*split_ptr : B);

  unsigned long long idx = 0;
  int iter = 0, updates = 0;
  double norm1 = 0.0, norm2 = 0.0;
  bool active = false, drained = false;

  for (;;) {
    // ---- refill: groups without a sample pull the next ticket ---------------------------------------------------
    {
      const bool want = live && !active && !drained;
      unsigned long long t = 0;
      if (want && r == 0) t = atomicAdd(queue, 1ull);
      t = __shfl(t, leader);
      if (want) {
        if (t < total) {
          active = true;
          if (SRC == 2) {
            const double *ent = pool + t * kPoolEntry;
            idx = (unsigned long long)__double_as_longlong(ent[14]);
            iter = __double2hiint(ent[15]);
            updates = __double2loint(ent[15]);
            norm1 = ent[16];
            norm2 = ent[17];
            for (int e = r; e < 14; e += rGroup) rec[rX + e] = ent[e];
          } else {
            idx = order != nullptr ? (unsigned long long)order[t] : t;
            iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
            for (int e = r; e < 14; e += rGroup) {
              double v;
              if (SRC == 0) v = q_in[idx * 14 + e];
              else {
                v = ambient_uniform(K, seed, first_index + idx, e);
                if (q_ambient) q_ambient[idx * 14 + e] = v;
              }
              rec[rX + e] = v;
            }
          }
        } else drained = true;
      }
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
    __syncthreads();
    // ---- function(x): sines / cosines (6 lanes, 3 rounds), both chains by rows, tool poses by rows ------------------
    for (int e = r; e < 14; e += rGroup) {
      double s, c;
      ccmp_sincos(rec[rX + e], &s, &c);
      if (live) { rec[rSC + 2 * e] = s; rec[rSC + 2 * e + 1] = c; }
    }
    __syncthreads();
    {
      double R0 = row == 0 ? 1.0 : 0.0, R1 = row == 1 ? 1.0 : 0.0, R2 = row == 2 ? 1.0 : 0.0, o = 0.0;
      rows_chain_from<0>(K, rec, rec + rSC + 14 * arm, arm, row, live, R0, R1, R2, o);
      asm volatile("" ::: "memory");
      double pf = o; // tool_pose_t<true>, diag(+-1) base frame, one row
      if (kStockEe & 1) pf = CCMP_FMA(R0, K.ee[0][0], pf);
      if (kStockEe & 2) pf = CCMP_FMA(R1, K.ee[0][1], pf);
      if (kStockEe & 4) pf = CCMP_FMA(R2, K.ee[0][2], pf);
      const double *Rt = K.R_tool[0];
      const double f0 = dot3(R0, Rt[0], R1, Rt[3], R2, Rt[6]);
      const double f1 = dot3(R0, Rt[1], R1, Rt[4], R2, Rt[7]);
      const double f2 = dot3(R0, Rt[2], R1, Rt[5], R2, Rt[8]);
      if (live) {
        double *T = rec + rT + 12 * arm;
        T[3 * row] = d_lane * f0;
        T[3 * row + 1] = d_lane * f1;
        T[3 * row + 2] = d_lane * f2;
        T[9 + row] = CCMP_FMA(d_lane, pf, bp_lane);
      }
    }
    __syncthreads();
    double T0[12], T1[12], f[2], dq[4], pc[3];
#pragma unroll
    for (int k = 0; k < 12; k++) { T0[k] = rec[rT + k]; T1[k] = rec[rT + 12 + k]; }
    chain_residual(K, &T0[0], &T0[9], &T1[0], &T1[9], f, dq, pc);
    // ---- loop condition of ConstraintFunction.h:68, quirks included ----------------------------------------------
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
      bool bad = false;
      if (fin) {
        for (int e = r; e < 14; e += rGroup) {
          const double v = rec[rX + e];
          const int jj = e < 7 ? e : e - 7;
          if (v < K.lbe[jj]) bad = true;
          if (v > K.ube[jj]) bad = true;
          q_out[idx * 14 + e] = wrap_output ? wrap_pi(v) : v;
        }
      }
      const unsigned long long badmask = __builtin_amdgcn_ballot_w64(bad);
      if (fin && r == 0) {
        const bool gbad = ((badmask >> leader) & 0x3Full) != 0ull;
        ok_out[idx] = (uint8_t)((!gbad) && (norm1 < K.tol_pos) && (norm2 < K.tol_rot));
        if (iters_out) iters_out[idx] = (uint16_t)updates;
      }
      if (fin) active = false;
    }
    if (__builtin_amdgcn_ballot_w64(cont) == 0ull) continue;
    // ---- analytic Jacobian: probes (every lane, its arm), then this lane's 2-3 joints of its arm ---------------------
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
      // base_R[arm] is diag(d0, d1, d2) here, but the probes take the general transposed product (the oracle's
      // m3t_vec): exact zeros are added, the bits are those of the one-lane kernel
      double al[3], bl[3], pl[3], dp[3];
#pragma unroll
      for (int k = 0; k < 3; k++) dp[k] = T0[9 + k] - K.base_p[arm][k];
      mulTvec(K.base_R[arm], aw, al);
      mulTvec(K.base_R[arm], bw, bl);
      mulTvec(K.base_R[arm], dp, pl);
      const double sgn = arm == 0 ? 1.0 : -1.0;
#pragma unroll
      for (int nn = 0; nn < 3; nn++) {
        const int i = row + 3 * nn; // joints row, row + 3, row + 6 of this lane's arm
        if (i < 7 && live) {
          const double *zo = rec + rZO + (arm * 7 + i) * 6;
          const double z0 = zo[0], z1 = zo[1], z2 = zo[2];
          const double r0 = pl[0] - zo[3], r1 = pl[1] - zo[4], r2 = pl[2] - zo[5];
          const double cx = CCMP_FMA(z1, r2, -(z2 * r1));
          const double cy = CCMP_FMA(z2, r0, -(z0 * r2));
          const double cz = CCMP_FMA(z0, r1, -(z1 * r0));
          rec[rJ + arm * 7 + i] = sgn * dot3(al[0], cx, al[1], cy, al[2], cz);
          rec[rJ + 14 + arm * 7 + i] = sgn * dot3(bl[0], z0, bl[1], z1, bl[2], z2);
        }
      }
    }
    __syncthreads();
    // ---- Newton update: x -= 0.30 * J.jacobiSvd().solve(f) ---------------------------------------------------------
    {
      double Jr[28], dx[14];
#pragma unroll
      for (int k = 0; k < 28; k++) Jr[k] = rec[rJ + k];
      solve_minnorm(Jr, f[0], f[1], dx);
      if (cont) {
#pragma unroll
        for (int e = 0; e < 14; e++)
          if (e % rGroup == r) rec[rX + e] = CCMP_FMA(-K.step, dx[e], rec[rX + e]);
        updates++;
      }
    }
    __syncthreads();
  }
}

} // namespace

extern "C" hipError_t ccmp_launch_clear_words(void *words, size_t n_u32, hipStream_t st); // ccmp_kernels_fd.hip

// One-lane kernel (+ hand-over of the samples past cap_iter iterations to the rows kernel when pool != NULL), or the
// rows kernel alone (lane_blocks == 0).  queue: kFastQueues words for the one-lane kernel, then one word for the rows
// kernel's tickets, one for the pool's fill count and one for the rows kernel's tickets of a split launch.
//
// Split launch (order != NULL): the batch is processed in `order` (longest predicted first); its first *split_ptr
// positions — the samples predicted longest — run on the six-lane kernel on stream `side` WHILE the one-lane kernel takes
// the rest on `st`, longest first; what the one-lane kernel still hands over past cap_iter (mispredictions) is finished
// by a second pass of the six-lane kernel behind both.  fork / join order the two streams (no host synchronisation).
extern "C" hipError_t ccmp_launch_project_fast(const ccmp_consts *K, int mode, const double *q_in, double *q_out,
                                               uint8_t *ok, uint16_t *iters, double *q_ambient, size_t B,
                                               unsigned long long *queue, unsigned long long seed,
                                               unsigned long long first, int lane_blocks, int rows_blocks, double *pool,
                                               int cap_iter, const unsigned int *order, const unsigned int *split_ptr,
                                               int front_blocks, hipStream_t side, hipEvent_t fork, hipEvent_t join, hipStream_t st)
{
  hipError_t e = ccmp_launch_clear_words(queue, (kFastQueues + 3) * 2, st); // a kernel, so that a stream capture replays it
  if (e != hipSuccess) return e;
  unsigned long long *rows_queue = queue + kFastQueues, *pool_count = queue + kFastQueues + 1, *front_queue = queue + kFastQueues + 2;
  const bool split = order != nullptr && split_ptr != nullptr && front_blocks > 0 && lane_blocks > 0;
  if (split) {
    if ((e = hipEventRecord(fork, st)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(side, fork, 0)) != hipSuccess) return e;
    if (mode == 0)
      hipLaunchKernelGGL((project_fast_rows_kernel<0>), dim3(front_blocks), dim3(64), 0, side, *K, q_in, q_out, ok, iters, q_ambient,
                         (unsigned long long)B, front_queue, seed, first, pool, pool_count, mode, order, split_ptr);
    else
      hipLaunchKernelGGL((project_fast_rows_kernel<1>), dim3(front_blocks), dim3(64), 0, side, *K, q_in, q_out, ok, iters, q_ambient,
                         (unsigned long long)B, front_queue, seed, first, pool, pool_count, mode, order, split_ptr);
    if ((e = hipEventRecord(join, side)) != hipSuccess) return e;
  }
  const unsigned int *lane_order = split ? order : nullptr, *lane_split = split ? split_ptr : nullptr;
  if (lane_blocks > 0) {
#define CCMP_LAUNCH_FAST(MODE, STOCK)                                                                                                \
  hipLaunchKernelGGL((project_fast_kernel<MODE, STOCK>), dim3(lane_blocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient, \
                     (unsigned long long)B, queue, seed, first, pool, pool_count, cap_iter, lane_order, lane_split)
    if (mode == 0) {
      if (K->stock) CCMP_LAUNCH_FAST(0, true);
      else CCMP_LAUNCH_FAST(0, false);
    } else {
      if (K->stock) CCMP_LAUNCH_FAST(1, true);
      else CCMP_LAUNCH_FAST(1, false);
    }
#undef CCMP_LAUNCH_FAST
    if (split && (e = hipStreamWaitEvent(st, join, 0)) != hipSuccess) return e;
    if (pool != nullptr && rows_blocks > 0) // the pool's fill count is read on the device: surplus waves exit at once
      hipLaunchKernelGGL((project_fast_rows_kernel<2>), dim3(rows_blocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient,
                         (unsigned long long)B, rows_queue, seed, first, pool, pool_count, mode, nullptr, nullptr);
  } else {
    if (mode == 0)
      hipLaunchKernelGGL((project_fast_rows_kernel<0>), dim3(rows_blocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient,
                         (unsigned long long)B, rows_queue, seed, first, pool, pool_count, mode, nullptr, nullptr);
    else
      hipLaunchKernelGGL((project_fast_rows_kernel<1>), dim3(rows_blocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient,
                         (unsigned long long)B, rows_queue, seed, first, pool, pool_count, mode, nullptr, nullptr);
  }
  return hipGetLastError();
}
