// ccmp_kernels_fd.hip — gfx950 kernels in the canonical (bit-reproducible) rounding model.
//
// Built with -ffp-contract=off -DCCMP_USE_FMA: the only fused operations are the CCMP_FMA calls in
// ccmp_detmath.h / ccmp_kin.h / this file, the same ones oracle/ccmp_oracle.c (det build) performs,
// so every kernel here is expected to agree with the CPU oracle BIT FOR BIT.
//
// Hot kernel: project_fd_kernel — KinematicChainConstraint::project with the reference's
// finite-difference Jacobian (include/closed_chain_motion_planner/base/constraints/
// ConstraintFunction.h:57-82 + OMPL's default Constraint::jacobian).
//
// Work decomposition: one wavefront = 10 independent samples x 6 lanes.
// The 6 lanes of a group are the 6 evaluation points of OMPL's 7-point central stencil
// (+h,+2h,+3h,-h,-2h,-3h) of ONE Jacobian column; the wave walks the 14 columns in lock-step, so
// arm/joint indices are wave-uniform and every kinematic constant is an SGPR operand.  A perturbed
// evaluation reuses the unperturbed prefix of the chain (bit-identical to recomputing it) from LDS
// and recomputes only the suffix.  Joint state, sines/cosines, prefix frames, tool poses and the
// 2x14 Jacobian of each sample are staged in LDS (one 265-double record per group; odd stride =>
// the 10 groups hit distinct banks, lanes of a group broadcast).  Groups pull samples from a
// global atomic queue, so a group whose sample converges early refills while its neighbours keep
// iterating (iteration counts spread 15..250).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccmp_kin.h"
#include "ccmp_solve.h"

using namespace ccmp;

namespace {

constexpr int kGroup = 6;             // lanes per sample = stencil evaluations per column
constexpr int kGroupsPerWave = 10;    // 60 of 64 lanes busy
// LDS record of one group (in doubles).  The prefix frames are kept for ONE arm at a time (the
// second arm's chain is re-run before its columns: +343 operations per iteration, -84 doubles of
// LDS per sample), and the second arm's Jacobian columns overwrite prefix slots that have already
// been consumed.  167 x 8 B x 10 groups = 13.4 KB per wave -> 12 waves per CU.
constexpr int kX = 0;                 // x[14]        current iterate
constexpr int kSC = 14;               // sc[14][2]    sin, cos of every joint of x
constexpr int kPre = 42;              // pre[7][12]   chain frame in front of joint j of the current arm: R(9), o(3)
                                      //              (slot j, doubles 0..1, is reused for J[:, 7+j] once consumed)
constexpr int kEE = 126;              // ee[2][12]    world tool pose of each arm at x: R(9), p(3)
constexpr int kJ0 = 150;              // J[:, 0..6]   interleaved (row0, row1) per column of arm 0
constexpr int kF = 164;               // f[2]         residual at x (parked here across the Jacobian phase: VGPR relief)
constexpr int kRec = 167;             // 166 used; odd stride keeps the 10 groups on distinct LDS banks
constexpr int kPoolEntry = 18;        // straggler hand-over record: x[14], idx, (iter,updates), norm1, norm2

#ifndef CCMP_FD_WAVES_PER_SIMD
#define CCMP_FD_WAVES_PER_SIMD 3
#endif
#ifndef CCMP_WAVE_WAVES_PER_SIMD
#define CCMP_WAVE_WAVES_PER_SIMD 2
#endif

__device__ __forceinline__ double shfl_f64(double v, int src_lane)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, lo);
  hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, hi);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src_lane)
{
  int lo = (int)(v & 0xffffffffull), hi = (int)(v >> 32);
  lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, lo);
  hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, hi);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// One arm's chain at x (sines/cosines from LDS).  With STORE the writer lane keeps the frame in
// front of every joint (R before the joint's rotation, o including the joint's offset) in LDS.
template <int ARM, bool STORE>
__device__ __forceinline__ void chain_at_x(const ccmp_consts &K, double *rec, bool writer, double *T)
{
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
  for (int i = 0; i < 7; i++) {
    double Rj[9], Rn[9];
    mulvec_acc(R, K.offset[ARM][i], o);
    if (STORE && writer) {
#pragma unroll
      for (int k = 0; k < 9; k++) rec[kPre + i * 12 + k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; k++) rec[kPre + i * 12 + 9 + k] = o[k];
    }
    rot_sc(K.axis[ARM][i], K.aprod[ARM][i], rec[kSC + 2 * (ARM * 7 + i)], rec[kSC + 2 * (ARM * 7 + i) + 1], Rj);
    mul33(R, Rj, Rn);
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
  }
  tool_pose(K, ARM, R, o, &T[0], &T[9]);
}

// OMPL's default Constraint::jacobian for the 7 columns of one arm: each lane evaluates its stencil
// point of column j from the cached prefix frame, the +/- lanes pair up through ds_bpermute.
template <int ARM>
__device__ __forceinline__ void jacobian_columns(const ccmp_consts &K, double *rec, bool writer, bool plus, int nstep,
                                                 int partner, int leader)
{
  for (int j = 0; j < 7; j++) {
    const double xj = rec[kX + ARM * 7 + j];
    const double axj = ccmp_abs(xj);
    const double h = 1.4901161193847656e-08 * (axj >= 1 ? axj : 1); // sqrt(eps)*max(1,|x_j|)
    const double hh = plus ? h : -h;
    double y = xj + hh;                 // y1[j] += h   /  y2[j] -= h
    if (nstep >= 2) y = y + hh;
    if (nstep >= 3) y = y + hh;
    double R[9], o[3], s, c;
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = rec[kPre + j * 12 + k];
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] = rec[kPre + j * 12 + 9 + k];
    ccmp_sincos(y, &s, &c);
    {
      double Rj[9], Rn[9];
      rot_sc(K.axis[ARM][j], K.aprod[ARM][j], s, c, Rj);
      mul33(R, Rj, Rn);
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = Rn[k];
    }
    for (int i = j + 1; i < 7; i++)
      joint_step(K, ARM, i, rec[kSC + 2 * (ARM * 7 + i)], rec[kSC + 2 * (ARM * 7 + i) + 1], R, o);
    double Tw[12], To[12], t[2];
    tool_pose(K, ARM, R, o, &Tw[0], &Tw[9]);
    {
      // the other arm's (unperturbed) tool pose, re-read from LDS every column: keeping it in
      // registers across the column loop costs 24 VGPRs and pushes the kernel into scratch spills
      const double *ee = rec + kEE + (1 - ARM) * 12;
      asm volatile("" : "+v"(ee));
#pragma unroll
      for (int k = 0; k < 12; k++) To[k] = ee[k];
    }
    if (ARM == 0) chain_residual(K, &Tw[0], &Tw[9], &To[0], &To[9], t, nullptr, nullptr);
    else chain_residual(K, &To[0], &To[9], &Tw[0], &Tw[9], t, nullptr, nullptr);
    // m_s = (t1 - t2) / (y1[j] - y2[j]) with the stored perturbed values
    const double tp0 = shfl_f64(t[0], partner), tp1 = shfl_f64(t[1], partner), yp = shfl_f64(y, partner);
    const double den = plus ? (y - yp) : (yp - y);
    const double m0 = (plus ? (t[0] - tp0) : (tp0 - t[0])) / den;
    const double m1 = (plus ? (t[1] - tp1) : (tp1 - t[1])) / den;
    const double m10 = shfl_f64(m0, leader), m20 = shfl_f64(m0, leader + 1), m30 = shfl_f64(m0, leader + 2);
    const double m11 = shfl_f64(m1, leader), m21 = shfl_f64(m1, leader + 1), m31 = shfl_f64(m1, leader + 2);
    if (writer) {
      // out.col(j) = 1.5*m1 - 0.6*m2 + 0.1*m3; arm 1's columns go into the prefix slot just consumed
      double *dst = (ARM == 0) ? (rec + kJ0 + 2 * j) : (rec + kPre + 12 * j);
      dst[0] = CCMP_FMA(0.1, m30, CCMP_FMA(-0.6, m20, 1.5 * m10));
      dst[1] = CCMP_FMA(0.1, m31, CCMP_FMA(-0.6, m21, 1.5 * m11));
    }
  }
}

// MODE 0: project q_in -> q_out.  MODE 1: sampleUniform = ambient sample -> project -> enforceBounds.
template <int MODE>
__global__ __launch_bounds__(64, CCMP_FD_WAVES_PER_SIMD) void project_fd_kernel(
    const ccmp_consts K, const double *__restrict__ q_in, double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
    uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient, unsigned long long B, unsigned long long *queue,
    unsigned long long seed, unsigned long long first_index, double *__restrict__ pool, unsigned long long *pool_count,
    int dump_threshold, const unsigned int *__restrict__ order)
{
  __shared__ double lds[kGroupsPerWave * kRec];
  const int lane = threadIdx.x;
  const int g = lane / kGroup;                // 0..10; lanes 60..63 form the idle "group 10"
  const int r = lane - kGroup * g;            // evaluation point 0..5 inside the group
  const bool live = g < kGroupsPerWave;
  const int leader = live ? kGroup * g : 0;
  double *rec = lds + (live ? g : 0) * kRec;  // idle lanes alias group 0 for reads, never write
  const bool writer = live && r == 0;
  const bool plus = r < 3;                    // y1 side of the stencil; r>=3 is the y2 side
  const int nstep = (plus ? r : r - 3) + 1;   // how many h-steps this lane's point is away
  const int partner = live ? (plus ? lane + 3 : lane - 3) : lane;

  unsigned long long idx = 0;
  int iter = 0, updates = 0;
  double norm1 = 0.0, norm2 = 0.0;
  bool active = false, drained = false;

  for (;;) {
    // ---- refill: groups without a sample pull the next index from the global queue ----------
    {
      const bool want = live && !active && !drained;
      unsigned long long t = 0;
      if (want && r == 0) t = atomicAdd(queue, 1ull);
      t = shfl_u64(t, leader);
      if (want) {
        if (t < B) {
          idx = order ? (unsigned long long)order[t] : t; // ticket -> sample: longest-predicted-first when scheduled
          active = true; iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
#pragma unroll
          for (int e = 0; e < 14; e++) {
            if (e % kGroup == r) {
              double v;
              if (MODE == 0) v = q_in[idx * 14 + e];
              else {
                v = ambient_uniform(K, seed, first_index + idx, e);
                if (q_ambient) q_ambient[idx * 14 + e] = v;
              }
              rec[kX + e] = v;
            }
          }
        } else drained = true;
      }
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
    __syncthreads();
    // ---- hand-over: once the queue is empty the samples still in flight go to the straggler pool
    // (x, index, counters — everything else is recomputed from x) and the wave retires; the
    // wave-per-sample kernel finishes them.  State is dumped at the loop top, where the next thing
    // that happens to a sample is function(x) + the loop test — exactly where the other kernel starts.
    if (pool != nullptr && __builtin_amdgcn_ballot_w64(drained) != 0ull &&
        __builtin_popcountll(__builtin_amdgcn_ballot_w64(active && r == 0)) <= dump_threshold) {
      unsigned long long slot = 0;
      if (active && r == 0) slot = atomicAdd(pool_count, 1ull);
      slot = shfl_u64(slot, leader);
      if (active) {
        double *ent = pool + slot * kPoolEntry;
        for (int e = r; e < 14; e += kGroup) ent[e] = rec[kX + e];
        if (r == 0) {
          ent[14] = __longlong_as_double((long long)idx);
          ent[15] = __hiloint2double(updates, iter);
          ent[16] = norm1;
          ent[17] = norm2;
        }
      }
      break;
    }

    // ---- phase 1: function(x) — sines/cosines, both chains, residual ---------------------------
    for (int e = r; e < 14; e += kGroup) {
      double s, c;
      ccmp_sincos(rec[kX + e], &s, &c);
      if (live) { rec[kSC + 2 * e] = s; rec[kSC + 2 * e + 1] = c; }
    }
    __syncthreads();
    bool cont = false;
    {
      double T0[12], T1[12], f[2];
      chain_at_x<1, false>(K, rec, writer, T1);
      if (writer) {
#pragma unroll
        for (int k = 0; k < 12; k++) rec[kEE + 12 + k] = T1[k];
      }
      chain_at_x<0, true>(K, rec, writer, T0); // arm 0's prefix frames stay in LDS for its columns
      if (writer) {
#pragma unroll
        for (int k = 0; k < 12; k++) rec[kEE + k] = T0[k];
      }
      chain_residual(K, &T0[0], &T0[9], &T1[0], &T1[9], f, nullptr, nullptr);
      if (writer) { rec[kF] = f[0]; rec[kF + 1] = f[1]; } // the solve reads them back after the Jacobian phase
      // ---- loop condition of ConstraintFunction.h:68, quirks included -------------------------
      // while (((norm1 = f[0] > tol1) || (norm2 = f[1]) > tol2) && iter++ < maxIterations)
      if (active) {
        const bool c1 = f[0] > K.tol_pos;
        norm1 = c1 ? 1.0 : 0.0;
        bool resid = c1;
        if (!c1) { norm2 = f[1]; resid = f[1] > K.tol_rot; }
        if (resid) { cont = iter < K.max_iter; iter++; }
      }
    }
    // ---- finished groups: jointValid, write-back --------------------------------------------
    {
      const bool fin = active && !cont;
      bool bad = false;
      if (fin) {
#pragma unroll
        for (int e = 0; e < 14; e++) {
          if (e % kGroup == r) { // e is a compile-time constant here: limits come from SGPRs
            const double v = rec[kX + e];
            if (v < K.lbe[e % 7]) bad = true;
            if (v > K.ube[e % 7]) bad = true;
            q_out[idx * 14 + e] = (MODE == 1) ? wrap_pi(v) : v;
          }
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
    if (__builtin_amdgcn_ballot_w64(cont) == 0ull) continue; // nobody iterates: straight to refill
    __syncthreads(); // prefix frames / tool poses written by the writer lane are visible to the group

    // ---- phase 2: OMPL's default Constraint::jacobian, one column per step --------------------
    jacobian_columns<0>(K, rec, writer, plus, nstep, partner, leader);
    __syncthreads();
    {
      double T1[12];
      chain_at_x<1, true>(K, rec, writer, T1); // re-run arm 1's chain to stage ITS prefix frames
    }
    __syncthreads();
    jacobian_columns<1>(K, rec, writer, plus, nstep, partner, leader);
    __syncthreads();

    // ---- Newton update: x -= 0.30 * J.jacobiSvd().solve(f) --------------------------------------
    {
      double Jr[28], dx[14];
#pragma unroll
      for (int j = 0; j < 7; j++) {
        Jr[j] = rec[kJ0 + 2 * j];
        Jr[14 + j] = rec[kJ0 + 2 * j + 1];
        Jr[7 + j] = rec[kPre + 12 * j];
        Jr[21 + j] = rec[kPre + 12 * j + 1];
      }
      solve_minnorm(Jr, rec[kF], rec[kF + 1], dx);
      if (cont) {
#pragma unroll
        for (int e = 0; e < 14; e++)
          if (e % kGroup == r) rec[kX + e] = CCMP_FMA(-K.step, dx[e], rec[kX + e]);
        updates++;
      }
    }
    __syncthreads();
  }
}


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

// RealVectorStateSpace::distance over the 14 joints (plain Euclidean, KinematicChainSpace does not
// override it), summed serially in the canonical order; every lane computes it from LDS.
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

// ------------------------------------------------------------------------------------------------
// geodesic_wave_kernel — jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:
// 32-96), one wavefront per edge (from -> to): interpolate a step of delta towards `to`
// (KinematicChainSpace::interpolate, KinematicChain.h:145-171), project it, apply the reference's
// four break tests, record the state.  The StateValidityChecker (MoveIt collision) stays on the host:
// the kernel runs as the reference does with interpolate == true and the host truncates the list at
// the first invalid state, which is what the reference's break would have produced.
constexpr int gPrev = wRec, gTo = wRec + 14, gRec = wRec + 28;

__global__ __launch_bounds__(64, CCMP_WAVE_WAVES_PER_SIMD) void geodesic_wave_kernel(
    const ccmp_consts K, const double delta, const double lambda, const double *__restrict__ from,
    const double *__restrict__ to, unsigned long long E, int max_states, double *__restrict__ states,
    int *__restrict__ n_states, uint8_t *__restrict__ ok_out, int *__restrict__ newton_iters, unsigned long long *queue)
{
  __shared__ double lds[gRec];
  __shared__ double ktab[kConstsDoubles + 1];
  const int lane = threadIdx.x;
  stage_consts(K, ktab, lane);
  __syncthreads();
  const ccmp_consts &KL = *reinterpret_cast<const ccmp_consts *>(ktab);
  double *rec = lds;
  const double pi = 3.14159265358979323846;

  for (;;) {
    unsigned long long t = 0;
    if (lane == 0) t = atomicAdd(queue, 1ull);
    t = shfl_u64(t, 0);
    if (t >= E) break;
    double *out = states + t * (unsigned long long)max_states * 14ull;
    if (lane < 14) {
      const double a = from[t * 14 + lane];
      rec[gPrev + lane] = a;
      rec[gTo + lane] = to[t * 14 + lane];
      if (max_states > 0) out[lane] = a; // geodesic->push_back(cloneState(from))
    }
    __syncthreads();
    int n = max_states > 0 ? 1 : 0, its = 0;
    double dist = lds_distance(rec + gPrev, rec + gTo), total = 0.0;
    if (dist > delta) {
      const double maxd = dist * lambda;
      for (int guard = 0; guard < 1000000; guard++) { // the reference loop ends by itself; guard bounds a non-finite input
        if (lane < 14) { // WrapperStateSpace::interpolate(previous, to, delta_ / dist, scratch)
          const double tt = delta / dist;
          const double fr = rec[gPrev + lane];
          double diff = rec[gTo + lane] - fr, v;
          if (ccmp_abs(diff) <= pi) v = CCMP_FMA(diff, tt, fr);
          else {
            if (diff > 0.0) diff = 2.0 * pi - diff;
            else diff = -2.0 * pi - diff;
            v = CCMP_FMA(-diff, tt, fr);
            if (v > pi) v -= 2.0 * pi;
            else if (v < -pi) v += 2.0 * pi;
          }
          rec[wX + lane] = v;
        }
        __syncthreads();
        int iter = 0, updates = 0;
        double norm1 = 0.0, norm2 = 0.0;
        const bool conv = wave_newton(K, KL, rec, lane, iter, updates, norm1, norm2);
        const bool jv = wave_joint_valid(KL, rec, lane);
        its += updates;
        if (!(conv && jv)) break;                        // not on manifold
        const double step = lds_distance(rec + gPrev, rec + wX);
        if (step > lambda * delta) break;                // deviated
        total += step;
        if (total > maxd) break;                         // wandered too far
        const double newDist = lds_distance(rec + wX, rec + gTo);
        if (newDist >= dist) break;                      // no closer than before
        dist = newDist;
        __syncthreads();
        if (lane < 14) {
          const double v = rec[wX + lane];
          rec[gPrev + lane] = v;
          if (n < max_states) out[(unsigned long long)n * 14ull + lane] = v;
        }
        if (n < max_states) n++;
        __syncthreads();
        if (!(dist >= delta)) break;
      }
    }
    if (lane == 0) {
      n_states[t] = n;
      ok_out[t] = (uint8_t)(dist <= delta);
      if (newton_iters) newton_iters[t] = its;
    }
    __syncthreads();
  }
}

// ---- simple per-lane kernels (one sample per lane; all bit-identical to the oracle) -------------
__global__ void function_kernel(const ccmp_consts K, const double *__restrict__ q, double *__restrict__ f, size_t B)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double x[14], out[2];
#pragma unroll
  for (int e = 0; e < 14; e++) x[e] = q[i * 14 + e];
  residual(K, x, out);
  f[2 * i] = out[0];
  f[2 * i + 1] = out[1];
}

// KinematicChainConstraint::isSatisfied (ConstraintFunction.h:114-120)
__global__ void is_satisfied_kernel(const ccmp_consts K, const double *__restrict__ q, uint8_t *__restrict__ ok, size_t B)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double x[14], f[2];
#pragma unroll
  for (int e = 0; e < 14; e++) x[e] = q[i * 14 + e];
  residual(K, x, f);
  const bool finite = (f[0] - f[0] == 0.0) && (f[1] - f[1] == 0.0);
  ok[i] = (uint8_t)(finite && f[0] <= K.tol_pos && f[1] <= K.tol_rot);
}

// KinematicChainConstraint::jointValid (ConstraintFunction.h:43-55)
__global__ void joint_valid_kernel(const ccmp_consts K, const double *__restrict__ q, uint8_t *__restrict__ ok, size_t B)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  bool good = true;
  for (int e = 0; e < 14; e++) {
    const double v = q[i * 14 + e];
    const int jj = e < 7 ? e : e - 7;
    if (v < K.lbe[jj]) good = false;
    if (v > K.ube[jj]) good = false;
  }
  ok[i] = (uint8_t)good;
}

__global__ void ambient_uniform_kernel(const ccmp_consts K, unsigned long long seed, unsigned long long first,
                                       double *__restrict__ q, size_t n /* = 14*B */)
{
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  q[t] = ambient_uniform(K, seed, first + t / 14, (int)(t % 14));
}

// sampleUniformNear / sampleGaussian ambient samples; `ref` holds one state per sample (stride 14) or a
// single state shared by all (stride 0)
__global__ void ambient_near_kernel(const ccmp_consts K, unsigned long long seed, unsigned long long first,
                                    const double *__restrict__ ref, int ref_stride, double dist, double *__restrict__ q, size_t n)
{
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t i = t / 14;
  const int j = (int)(t % 14);
  q[t] = ambient_near(K, seed, first + i, j, ref[i * (size_t)ref_stride + j], dist);
}
__global__ void ambient_gaussian_kernel(const ccmp_consts K, unsigned long long seed, unsigned long long first,
                                        const double *__restrict__ ref, int ref_stride, double stddev, double *__restrict__ q,
                                        size_t n)
{
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t i = t / 14;
  const int j = (int)(t % 14);
  q[t] = ambient_gaussian(K, seed, first + i, j, ref[i * (size_t)ref_stride + j], stddev);
}

// IKTask::compute_t_wo (src/base/constraints/ik_task.cpp:10-14): object pose from the left arm's joints,
// t_wb * FK(q) * t_o7.inverse(); out[i] = R (9, row-major) then p (3)
__global__ void t_wo_kernel(const ccmp_consts K, const double *__restrict__ q, int q_stride, double *__restrict__ out, size_t B)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double x[7], Rw[9], pw[3], R[9], p[3];
#pragma unroll
  for (int e = 0; e < 7; e++) x[e] = q[i * (size_t)q_stride + e];
  fk_arm(K, 0, x, Rw, pw);
  mul33(Rw, K.t_o7i_R, R);
  p[0] = pw[0]; p[1] = pw[1]; p[2] = pw[2];
  mulvec_acc(Rw, K.t_o7i_p, p);
#pragma unroll
  for (int k = 0; k < 9; k++) out[i * 12 + k] = R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) out[i * 12 + 9 + k] = p[k];
}

__global__ void enforce_bounds_kernel(double *__restrict__ q, size_t n)
{
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  q[t] = wrap_pi(q[t]);
}

__global__ void detmath_probe_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                     double *__restrict__ out, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  ccmp_sincos(x[i], &s, &c);
  out[5 * i + 0] = s;
  out[5 * i + 1] = c;
  out[5 * i + 2] = ccmp_atan2_nn(ccmp_abs(x[i]), ccmp_abs(y[i]));
  out[5 * i + 3] = ccmp_sqrt(ccmp_abs(x[i]));
  out[5 * i + 4] = x[i] / y[i];
}

// Stable stream compaction of valid rows: block-local scan + one atomic per block would reorder
// blocks, so this is the ordered two-pass form: (1) per-block counts, (2) single-block exclusive
// scan of the counts, (3) scatter.  B/256 counts fit one block's loop comfortably (1024 at 262144).
__global__ void compact_count_kernel(const uint8_t *__restrict__ ok, size_t B, unsigned int *__restrict__ block_counts)
{
  __shared__ unsigned int wsum[4];
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool v = i < B && ok[i] != 0;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned)__builtin_popcountll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ void compact_scan_kernel(unsigned int *__restrict__ block_counts, size_t nblocks,
                                    unsigned long long *__restrict__ total)
{
  // one 1024-thread block; each thread owns a contiguous chunk
  __shared__ unsigned long long part[1024];
  const size_t per = (nblocks + 1023) / 1024;
  const size_t lo = (size_t)threadIdx.x * per, hi = lo + per < nblocks ? lo + per : nblocks;
  unsigned long long s = 0;
  for (size_t k = lo; k < hi; k++) s += block_counts[k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int k = 0; k < 1024; k++) { unsigned long long v = part[k]; part[k] = run; run += v; }
    *total = run;
  }
  __syncthreads();
  unsigned long long run = part[threadIdx.x];
  for (size_t k = lo; k < hi; k++) { unsigned int v = block_counts[k]; block_counts[k] = (unsigned int)run; run += v; }
}
__global__ void compact_scatter_kernel(const double *__restrict__ q, const uint8_t *__restrict__ ok, size_t B,
                                       const unsigned int *__restrict__ block_offsets, double *__restrict__ out)
{
  __shared__ unsigned int wsum[4];
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool v = i < B && ok[i] != 0;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wsum[w] = (unsigned)__builtin_popcountll(m);
  __syncthreads();
  unsigned int base = block_offsets[blockIdx.x];
  for (int k = 0; k < w; k++) base += wsum[k];
  if (v) {
    const unsigned int pos = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
#pragma unroll
    for (int e = 0; e < 14; e++) out[(size_t)pos * 14 + e] = q[i * 14 + e];
  }
}

} // namespace

// ---- launchers (called from ccmp_api.cpp) --------------------------------------------------------
extern "C" {

hipError_t ccmp_launch_project_fd(const ccmp_consts *K, int mode, const double *q_in, double *q_out, uint8_t *ok,
                                  uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue,
                                  unsigned long long seed, unsigned long long first, int nblocks, double *pool,
                                  int nblocks_wave, int dump_threshold, const unsigned int *order, hipStream_t st)
{
  // queue[0]: sample queue of the group kernel; queue[1]: pool fill count; queue[2]: pool read head
  hipError_t e = hipMemsetAsync(queue, 0, 4 * sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  if (nblocks > 0) {
    double *pl = nblocks_wave > 0 ? pool : nullptr;
    if (mode == 0)
      hipLaunchKernelGGL(project_fd_kernel<0>, dim3(nblocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient,
                         (unsigned long long)B, queue, seed, first, pl, queue + 1, dump_threshold, order);
    else
      hipLaunchKernelGGL(project_fd_kernel<1>, dim3(nblocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient,
                         (unsigned long long)B, queue, seed, first, pl, queue + 1, dump_threshold, order);
    if (nblocks_wave > 0)
      hipLaunchKernelGGL(project_fd_wave_kernel<2>, dim3(nblocks_wave), dim3(64), 0, st, *K, q_in, q_out, ok, iters,
                         q_ambient, (unsigned long long)B, queue + 2, seed, first, pool, queue + 1, mode);
  } else { // small batch: wave-per-sample kernel on everything
    if (mode == 0)
      hipLaunchKernelGGL(project_fd_wave_kernel<0>, dim3(nblocks_wave), dim3(64), 0, st, *K, q_in, q_out, ok, iters,
                         q_ambient, (unsigned long long)B, queue + 2, seed, first, pool, queue + 1, 0);
    else
      hipLaunchKernelGGL(project_fd_wave_kernel<1>, dim3(nblocks_wave), dim3(64), 0, st, *K, q_in, q_out, ok, iters,
                         q_ambient, (unsigned long long)B, queue + 2, seed, first, pool, queue + 1, 1);
  }
  return hipGetLastError();
}

hipError_t ccmp_launch_function(const ccmp_consts *K, const double *q, double *f, size_t B, hipStream_t st)
{
  hipLaunchKernelGGL(function_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, *K, q, f, B);
  return hipGetLastError();
}
hipError_t ccmp_launch_is_satisfied(const ccmp_consts *K, const double *q, uint8_t *ok, size_t B, hipStream_t st)
{
  hipLaunchKernelGGL(is_satisfied_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, *K, q, ok, B);
  return hipGetLastError();
}
hipError_t ccmp_launch_joint_valid(const ccmp_consts *K, const double *q, uint8_t *ok, size_t B, hipStream_t st)
{
  hipLaunchKernelGGL(joint_valid_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, *K, q, ok, B);
  return hipGetLastError();
}
hipError_t ccmp_launch_ambient_uniform(const ccmp_consts *K, unsigned long long seed, unsigned long long first,
                                       double *q, size_t B, hipStream_t st)
{
  size_t n = B * 14;
  hipLaunchKernelGGL(ambient_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *K, seed, first, q, n);
  return hipGetLastError();
}
hipError_t ccmp_launch_ambient_ref(const ccmp_consts *K, int kind, unsigned long long seed, unsigned long long first,
                                   const double *ref, int ref_stride, double param, double *q, size_t B, hipStream_t st)
{
  size_t n = B * 14;
  if (kind == 0)
    hipLaunchKernelGGL(ambient_near_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *K, seed, first, ref,
                       ref_stride, param, q, n);
  else
    hipLaunchKernelGGL(ambient_gaussian_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *K, seed, first, ref,
                       ref_stride, param, q, n);
  return hipGetLastError();
}
hipError_t ccmp_launch_t_wo(const ccmp_consts *K, const double *q, int q_stride, double *out, size_t B, hipStream_t st)
{
  hipLaunchKernelGGL(t_wo_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, *K, q, q_stride, out, B);
  return hipGetLastError();
}
hipError_t ccmp_launch_geodesic(const ccmp_consts *K, double delta, double lambda, const double *from, const double *to,
                                size_t E, int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters,
                                unsigned long long *queue, int nblocks, hipStream_t st)
{
  hipError_t e = hipMemsetAsync(queue, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(geodesic_wave_kernel, dim3(nblocks), dim3(64), 0, st, *K, delta, lambda, from, to,
                     (unsigned long long)E, max_states, states, n_states, ok, newton_iters, queue);
  return hipGetLastError();
}
hipError_t ccmp_launch_enforce_bounds(double *q, size_t B, hipStream_t st)
{
  size_t n = B * 14;
  hipLaunchKernelGGL(enforce_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q, n);
  return hipGetLastError();
}
hipError_t ccmp_launch_detmath_probe(const double *x, const double *y, double *out, size_t n, hipStream_t st)
{
  hipLaunchKernelGGL(detmath_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, out, n);
  return hipGetLastError();
}
hipError_t ccmp_launch_compact(const double *q, const uint8_t *ok, size_t B, double *out, unsigned int *block_counts,
                               unsigned long long *total, hipStream_t st)
{
  const size_t nblocks = (B + 255) / 256;
  hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, ok, B, block_counts);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, block_counts, nblocks, total);
  hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, q, ok, B, block_counts, out);
  return hipGetLastError();
}

} // extern "C"
