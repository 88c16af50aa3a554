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
                                                          unsigned long long first_index)
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

  for (;;) {
    unsigned long long need = __builtin_amdgcn_ballot_w64(!active && !drained);
    while (need != 0ull) {
      const unsigned long long lo = B * (unsigned long long)qk / kFastQueues, hi = B * (unsigned long long)(qk + 1) / kFastQueues;
      const int n = __builtin_popcountll(need);
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(queue + qk, (unsigned long long)n);
      base = __shfl(base, 0);  // ticket of the first free lane, relative to the slice
      const bool mine = (need >> lane) & 1ull;
      const unsigned long long t = lo + base + (unsigned long long)__builtin_popcountll(need & ((1ull << lane) - 1ull));
      if (mine && t < hi) {
        idx = t; active = true; iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
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

} // namespace

extern "C" hipError_t ccmp_launch_project_fast(const ccmp_consts *K, int mode, const double *q_in, double *q_out,
                                               uint8_t *ok, uint16_t *iters, double *q_ambient, size_t B,
                                               unsigned long long *queue, unsigned long long seed,
                                               unsigned long long first, int nblocks, hipStream_t st)
{
  hipError_t e = hipMemsetAsync(queue, 0, kFastQueues * sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
#define CCMP_LAUNCH_FAST(MODE, STOCK)                                                                                              \
  hipLaunchKernelGGL((project_fast_kernel<MODE, STOCK>), dim3(nblocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient, \
                     (unsigned long long)B, queue, seed, first)
  if (mode == 0) {
    if (K->stock) CCMP_LAUNCH_FAST(0, true);
    else CCMP_LAUNCH_FAST(0, false);
  } else {
    if (K->stock) CCMP_LAUNCH_FAST(1, true);
    else CCMP_LAUNCH_FAST(1, false);
  }
#undef CCMP_LAUNCH_FAST
  return hipGetLastError();
}
