// ccmp_kernels_wave.hip — the one-wavefront-per-sample projector in the canonical (bit-reproducible) rounding model:
// the north star's literal layout, selectable with the context option "flat_kernel" = 0 (the default latency kernel is
// ccmp_kernels_flat.hip).  Built -ffp-contract=off -DCCMP_USE_FMA like ccmp_kernels_fd.hip; kept in its own
// translation unit because the two benefit from different code-generation options (build.py).
#include "ccmp_fd_common.h"

using namespace ccmp;

namespace {

#ifndef CCMP_WAVE_WAVES_PER_SIMD
#define CCMP_WAVE_WAVES_PER_SIMD 2
#endif

// ------------------------------------------------------------------------------------------------
// project_fd_wave_kernel — the same arithmetic, ONE WAVEFRONT PER SAMPLE (latency-oriented).
// The 84 stencil evaluations of an iteration are spread over the 64 lanes (two rounds: 64 + 20, the
// second round holds the evaluations with the shortest chain suffix), the two arms' chains at x run
// on the two half-waves, 28 lanes combine the stencil into J.  Arm/joint indices differ per lane
// here, so the kinematic constants are read from an LDS copy of ccmp_consts (same source functions,
// same operation order, hence the same bits as the group kernel and the oracle).  ~2.5x the
// wave-instructions per sample-iteration of the group kernel, but ~5x lower latency per sample:
// used for the stragglers the group kernel hands over and for small batches.
// SRC 0: q_in, SRC 1: ambient sampler, SRC 2: straggler pool.
constexpr int wX = 0, wSC = 14, wPre = 42, wEE = 210, wJ = 234, wT = 262, wY = 430, wRec = 514;
constexpr int kConstsDoubles = (int)((sizeof(ccmp_consts) + 7) / 8);

// Newton iterations of the sample whose iterate sits in rec[wX..wX+13], all 64 lanes cooperating.
// Returns the reference's bool (without jointValid: see wave_joint_valid) and leaves x at the last
// iterate.  iter/updates/norm1/norm2 carry the loop state (non-zero when resuming a handed-over
// sample).  Every lane returns the same values (the control flow is wave-uniform).
__device__ __forceinline__ bool wave_newton(const ccmp_consts &K, const ccmp_consts &KL, double *rec, int lane, int &iter,
                                            int &updates, double &norm1, double &norm2)
{
  for (;;) {
    // ---- phase 1: function(x) -------------------------------------------------------------------
    if (lane < 14) {
      double s, c;
      ccmp_sincos(rec[wX + lane], &s, &c);
      rec[wSC + 2 * lane] = s;
      rec[wSC + 2 * lane + 1] = c;
    }
    __syncthreads();
    {
      const int arm = lane >> 5; // half-wave per arm
      double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0}, T[12];
      const bool wr = (lane & 31) == 0;
      for (int i = 0; i < 7; i++) {
        const int col = arm * 7 + i;
        double Rj[9], Rn[9];
        mulvec_acc(R, KL.offset[arm][i], o);
        if (wr) {
#pragma unroll
          for (int k = 0; k < 9; k++) rec[wPre + col * 12 + k] = R[k];
#pragma unroll
          for (int k = 0; k < 3; k++) rec[wPre + col * 12 + 9 + k] = o[k];
        }
        rot_sc(KL.axis[arm][i], KL.aprod[arm][i], rec[wSC + 2 * col], rec[wSC + 2 * col + 1], Rj);
        mul33(R, Rj, Rn);
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = Rn[k];
      }
      tool_pose(KL, arm, R, o, &T[0], &T[9]);
      if (wr) {
#pragma unroll
        for (int k = 0; k < 12; k++) rec[wEE + arm * 12 + k] = T[k];
      }
    }
    __syncthreads();
    double f0, f1;
    {
      double T0[12], T1[12], f[2];
#pragma unroll
      for (int k = 0; k < 12; k++) { T0[k] = rec[wEE + k]; T1[k] = rec[wEE + 12 + k]; }
      chain_residual(K, &T0[0], &T0[9], &T1[0], &T1[9], f, nullptr, nullptr);
      f0 = f[0]; f1 = f[1];
    }
    // ---- loop condition of ConstraintFunction.h:68 (wave-uniform here) ---------------------------
    bool cont = false;
    {
      const bool c1 = f0 > K.tol_pos;
      norm1 = c1 ? 1.0 : 0.0;
      bool resid = c1;
      if (!c1) { norm2 = f1; resid = f1 > K.tol_rot; }
      if (resid) { cont = iter < K.max_iter; iter++; }
    }
    if (!cont) return (norm1 < K.tol_pos) && (norm2 < K.tol_rot);

    // ---- phase 2: the 84 stencil evaluations, two rounds -------------------------------------------
    // evaluation e = 6*cs + point, columns sorted by chain-suffix length: cs -> (arm = cs&1, j = cs>>1)
#pragma unroll
    for (int round = 0; round < 2; round++) {
      const int e = lane + 64 * round;
      const bool valid = e < 84;
      const int ec = valid ? e : 83;
      const int cs = ec / 6, pt = ec - 6 * cs;
      const int arm = cs & 1, j = cs >> 1, col = arm * 7 + j;
      const bool plus = pt < 3;
      const int nstep = (plus ? pt : pt - 3) + 1;
      const double xj = rec[wX + col];
      const double axj = ccmp_abs(xj);
      const double h = 1.4901161193847656e-08 * (axj >= 1 ? axj : 1);
      const double hh = plus ? h : -h;
      double y = xj + hh;
      if (nstep >= 2) y = y + hh;
      if (nstep >= 3) y = y + hh;
      double R[9], o[3], s, c;
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = rec[wPre + col * 12 + k];
#pragma unroll
      for (int k = 0; k < 3; k++) o[k] = rec[wPre + col * 12 + 9 + k];
      ccmp_sincos(y, &s, &c);
      {
        double Rj[9], Rn[9];
        rot_sc(KL.axis[arm][j], KL.aprod[arm][j], s, c, Rj);
        mul33(R, Rj, Rn);
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = Rn[k];
      }
      // suffix j+1..6; round 1 only holds j >= 5, so its loop is a single step
      for (int i = (round == 0 ? 1 : 6); i < 7; i++) {
        if (i > j) joint_step(KL, arm, i, rec[wSC + 2 * (arm * 7 + i)], rec[wSC + 2 * (arm * 7 + i) + 1], R, o);
      }
      double Tw[12], To[12], tt[2];
      tool_pose(KL, arm, R, o, &Tw[0], &Tw[9]);
#pragma unroll
      for (int k = 0; k < 12; k++) To[k] = rec[wEE + (1 - arm) * 12 + k];
      {
        double A[12], Bq[12]; // (T1, T2) in role order: the perturbed arm's pose takes its own slot
#pragma unroll
        for (int k = 0; k < 12; k++) { A[k] = arm ? To[k] : Tw[k]; Bq[k] = arm ? Tw[k] : To[k]; }
        chain_residual(K, &A[0], &A[9], &Bq[0], &Bq[9], tt, nullptr, nullptr);
      }
      if (valid) {
        rec[wT + 2 * (6 * col + pt)] = tt[0];
        rec[wT + 2 * (6 * col + pt) + 1] = tt[1];
        rec[wY + 6 * col + pt] = y;
      }
    }
    __syncthreads();
    if (lane < 28) { // J[row][col] = 1.5 m1 - 0.6 m2 + 0.1 m3, m_s = (t1 - t2) / (y1[j] - y2[j])
      const int row = lane >= 14 ? 1 : 0, col = lane - 14 * row;
      double m[3];
#pragma unroll
      for (int sidx = 0; sidx < 3; sidx++) {
        const int e1 = 6 * col + sidx, e2 = 6 * col + 3 + sidx;
        m[sidx] = (rec[wT + 2 * e1 + row] - rec[wT + 2 * e2 + row]) / (rec[wY + e1] - rec[wY + e2]);
      }
      rec[wJ + lane] = CCMP_FMA(0.1, m[2], CCMP_FMA(-0.6, m[1], 1.5 * m[0]));
    }
    __syncthreads();
    {
      double Jr[28], dx[14];
#pragma unroll
      for (int k = 0; k < 28; k++) Jr[k] = rec[wJ + k];
      solve_minnorm(Jr, f0, f1, dx);
#pragma unroll
      for (int e = 0; e < 14; e++)
        if (e == lane) rec[wX + e] = CCMP_FMA(-K.step, dx[e], rec[wX + e]);
      updates++;
    }
    __syncthreads();
  }
}

// jointValid(x) of the iterate in rec[wX..] (ConstraintFunction.h:43-55), wave-uniform result
__device__ __forceinline__ bool wave_joint_valid(const ccmp_consts &KL, const double *rec, int lane)
{
  bool bad = false;
  if (lane < 14) {
    const double v = rec[wX + lane];
    const int jj = lane < 7 ? lane : lane - 7;
    if (v < KL.lbe[jj]) bad = true;
    if (v > KL.ube[jj]) bad = true;
  }
  return __builtin_amdgcn_ballot_w64(bad) == 0ull;
}

__device__ __forceinline__ void stage_consts(const ccmp_consts &K, double *ktab, int lane)
{
  const double *src = reinterpret_cast<const double *>(&K);
  for (int k = lane; k < kConstsDoubles; k += 64) ktab[k] = src[k];
}

template <int SRC>
__global__ __launch_bounds__(64, CCMP_WAVE_WAVES_PER_SIMD) void project_fd_wave_kernel(
    const ccmp_consts K, const double *__restrict__ q_in, double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
    uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient, unsigned long long B, unsigned long long *queue,
    unsigned long long seed, unsigned long long first_index, const double *__restrict__ pool,
    const unsigned long long *__restrict__ pool_count, int wrap_output)
{
  __shared__ double lds[wRec];
  __shared__ double ktab[kConstsDoubles + 1];
  const int lane = threadIdx.x;
  stage_consts(K, ktab, lane);
  __syncthreads();
  const ccmp_consts &KL = *reinterpret_cast<const ccmp_consts *>(ktab);
  double *rec = lds;
  const unsigned long long total = (SRC == 2) ? *pool_count : B;

  for (;;) {
    // ---- next sample of this wave --------------------------------------------------------------
    unsigned long long t = 0;
    if (lane == 0) t = atomicAdd(queue, 1ull);
    t = shfl_u64(t, 0);
    if (t >= total) break;
    unsigned long long idx;
    int iter = 0, updates = 0;
    double norm1 = 0.0, norm2 = 0.0;
    if (SRC == 2) {
      const double *ent = pool + t * kPoolEntry;
      idx = (unsigned long long)__double_as_longlong(ent[14]);
      iter = __double2hiint(ent[15]);
      updates = __double2loint(ent[15]);
      norm1 = ent[16];
      norm2 = ent[17];
      if (lane < 14) rec[wX + lane] = ent[lane];
    } else {
      idx = t;
      if (lane < 14) {
        double v;
        if (SRC == 0) v = q_in[idx * 14 + lane];
        else {
          v = ambient_uniform(KL, seed, first_index + idx, lane);
          if (q_ambient) q_ambient[idx * 14 + lane] = v;
        }
        rec[wX + lane] = v;
      }
    }
    __syncthreads();
    const bool conv = wave_newton(K, KL, rec, lane, iter, updates, norm1, norm2);
    const bool jv = wave_joint_valid(KL, rec, lane);
    if (lane < 14) {
      const double v = rec[wX + lane];
      q_out[idx * 14 + lane] = wrap_output ? wrap_pi(v) : v;
    }
    if (lane == 0) {
      ok_out[idx] = (uint8_t)(jv && conv);
      if (iters_out) iters_out[idx] = (uint16_t)updates;
    }
    __syncthreads();
  }
}

} // namespace

extern "C" {

// src 0: q_in, 1: ambient sampler, 2: straggler pool (count read from *pool_count on the device)
hipError_t ccmp_launch_project_wave(const ccmp_consts *K, int src, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue_head,
                                    unsigned long long seed, unsigned long long first, const double *pool,
                                    const unsigned long long *pool_count, int wrap_output, int nblocks, hipStream_t st)
{
  if (src == 0)
    hipLaunchKernelGGL(project_fd_wave_kernel<0>, dim3(nblocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient,
                       (unsigned long long)B, queue_head, seed, first, pool, pool_count, wrap_output);
  else if (src == 1)
    hipLaunchKernelGGL(project_fd_wave_kernel<1>, dim3(nblocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient,
                       (unsigned long long)B, queue_head, seed, first, pool, pool_count, wrap_output);
  else
    hipLaunchKernelGGL(project_fd_wave_kernel<2>, dim3(nblocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient,
                       (unsigned long long)B, queue_head, seed, first, pool, pool_count, wrap_output);
  return hipGetLastError();
}

} // extern "C"
