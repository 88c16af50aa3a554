// ccmp_kernels_scout.hip — FP32 "scout" pass and longest-predicted-first ordering.
//
// The reference-arithmetic projector is fed from a work queue; Newton iteration counts spread from 15
// to 250, so a batch ends with a tail in which a few long samples run on an otherwise idle chip.  An
// oracle-perfect longest-first order removes that tail (tools/time_lpt.py).  This unit provides the
// predictor: the SAME Newton iteration (same residual, stopping rule, step 0.30, cap) with the exact
// (analytic) Jacobian, in single precision, one sample per lane — ~35x less arithmetic than the
// finite-difference Jacobian and cheap FP32 sin/cos/div/sqrt — whose only output is the number of
// updates each sample needed; then a 3-kernel counting sort (descending) produces the processing
// order handed to project_fd_kernel.  NOTHING computed here reaches q_out/ok/iters: the order in
// which samples are processed does not change a single bit of the results (the parity tests run with
// and without it).  Built with contraction and fast-math; no rounding-model obligations.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccmp_kin.h" // ccmp_consts, ambient_uniform (double, so that MODE 1 sees the same samples)
#include "ccmp_split.h"

namespace {

struct consts_f {
  float axis[2][7][3], offset[2][7][3], ee[2][3], R_tool[2][9], base_R[2][9], base_p[2][3];
  float init_p[3], init_q[4];
  float tol_pos, tol_rot, step;
  int max_iter; // scout cap: min(problem cap, kScoutCap) — samples still unconverged there are all "long"
  float lbe[7], ube[7]; // joint limits with the margin (extend-step scout: project() must return true for a state to count)
};
constexpr int kScoutCap = 96;

__device__ __forceinline__ void mul33f(const float *A, const float *B, float *C)
{
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

__device__ __forceinline__ void rotf(const float *a, float s, float c, float *R)
{
  const float t = 1.0f - c;
  R[0] = a[0] * a[0] * t + c;        R[1] = a[0] * a[1] * t - a[2] * s; R[2] = a[0] * a[2] * t + a[1] * s;
  R[3] = a[0] * a[1] * t + a[2] * s; R[4] = a[1] * a[1] * t + c;        R[5] = a[1] * a[2] * t - a[0] * s;
  R[6] = a[0] * a[2] * t - a[1] * s; R[7] = a[1] * a[2] * t + a[0] * s; R[8] = a[2] * a[2] * t + c;
}

// quaternion (x,y,z,w) of a rotation matrix (Shepperd's branches, as Eigen)
__device__ __forceinline__ void quatf(const float *m, float *q)
{
  const float tr = m[0] + m[4] + m[8];
  if (tr > 0.0f) {
    float t = sqrtf(tr + 1.0f);
    q[3] = 0.5f * t; t = 0.5f / t;
    q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
  } else if (m[0] >= m[4] && m[0] >= m[8]) {
    float t = sqrtf(m[0] - m[4] - m[8] + 1.0f);
    q[0] = 0.5f * t; t = 0.5f / t;
    q[3] = (m[7] - m[5]) * t; q[1] = (m[3] + m[1]) * t; q[2] = (m[6] + m[2]) * t;
  } else if (m[4] >= m[8]) {
    float t = sqrtf(m[4] - m[8] - m[0] + 1.0f);
    q[1] = 0.5f * t; t = 0.5f / t;
    q[3] = (m[2] - m[6]) * t; q[2] = (m[7] + m[5]) * t; q[0] = (m[1] + m[3]) * t;
  } else {
    float t = sqrtf(m[8] - m[0] - m[4] + 1.0f);
    q[2] = 0.5f * t; t = 0.5f / t;
    q[3] = (m[3] - m[1]) * t; q[0] = (m[2] + m[6]) * t; q[1] = (m[5] + m[7]) * t;
  }
}

// One arm's chain in float.  PASS 0: only the tool pose.  PASS 1: additionally the two Jacobian rows of the
// arm's 7 joints from the probe vectors (al, bl, pl) expressed in the arm's base frame; sin/cos are simply
// recomputed (FP32 sincos is two transcendental instructions) instead of keeping 84 floats of joint frames.
template <int PASS>
__device__ __forceinline__ void chain_f(const consts_f &K, const int arm, const float *q, float *Rw, float *pw,
                                        const float *al, const float *bl, const float *pl, float sgn, float *J0, float *J1)
{
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 7; i++) {
    float s, c;
    __sincosf(q[i], &s, &c);
    const float *off = K.offset[arm][i], *a = K.axis[arm][i];
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] += R[3 * k] * off[0] + R[3 * k + 1] * off[1] + R[3 * k + 2] * off[2];
    if (PASS == 1) {
      float z[3];
#pragma unroll
      for (int k = 0; k < 3; k++) z[k] = R[3 * k] * a[0] + R[3 * k + 1] * a[1] + R[3 * k + 2] * a[2];
      const float r0 = pl[0] - o[0], r1 = pl[1] - o[1], r2 = pl[2] - o[2];
      const float cx = z[1] * r2 - z[2] * r1, cy = z[2] * r0 - z[0] * r2, cz = z[0] * r1 - z[1] * r0;
      J0[i] = sgn * (al[0] * cx + al[1] * cy + al[2] * cz);
      J1[i] = sgn * (bl[0] * z[0] + bl[1] * z[1] + bl[2] * z[2]);
    }
    if (PASS == 0 || i < 6) {
      float Rj[9], Rn[9];
      rotf(a, s, c, Rj);
      mul33f(R, Rj, Rn);
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = Rn[k];
    }
  }
  if (PASS == 0) {
    float pf[3], Rf[9];
#pragma unroll
    for (int k = 0; k < 3; k++) pf[k] = o[k] + R[3 * k] * K.ee[arm][0] + R[3 * k + 1] * K.ee[arm][1] + R[3 * k + 2] * K.ee[arm][2];
    mul33f(R, K.R_tool[arm], Rf);
    mul33f(K.base_R[arm], Rf, Rw);
#pragma unroll
    for (int k = 0; k < 3; k++)
      pw[k] = K.base_p[arm][k] + K.base_R[arm][3 * k] * pf[0] + K.base_R[arm][3 * k + 1] * pf[1] + K.base_R[arm][3 * k + 2] * pf[2];
  }
}

// The same chain for the stock Panda structure, where every joint axis is a coordinate axis (to FP32: the 6e-17
// leftovers of cos(pi/2) in the reference's constants are far below single precision): joint rotations are planar,
// most offset components vanish, the tool rotation is block-diagonal.  A prediction needs no more than that; the
// launcher falls back to chain_f when the constants do not look like this (calibrated arms).
//                                joint:  1   2   3   4   5   6   7
constexpr int kAxIdx[7] = {2, 1, 2, 1, 2, 1, 2};          // rotation axis: 1 = y, 2 = z
constexpr float kAxSgn[7] = {1.f, 1.f, 1.f, -1.f, 1.f, -1.f, -1.f};
constexpr int kOffNz[7] = {4, 0, 4, 1, 5, 0, 1};          // non-zero offset components (bit k)

template <int PASS>
__device__ __forceinline__ void chain_stock_f(const consts_f &K, const int arm, const float *q, float *Rw, float *pw,
                                              const float *al, const float *bl, const float *pl, float sgn, float *J0, float *J1)
{
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 7; i++) {
    float s, c;
    __sincosf(q[i], &s, &c);
    s *= kAxSgn[i];
    const float *off = K.offset[arm][i];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (kOffNz[i] & 1) o[k] += R[3 * k] * off[0];
      if (kOffNz[i] & 4) o[k] += R[3 * k + 2] * off[2];
    }
    if (PASS == 1) {
      float z[3];
#pragma unroll
      for (int k = 0; k < 3; k++) z[k] = kAxSgn[i] * R[3 * k + kAxIdx[i]];
      const float r0 = pl[0] - o[0], r1 = pl[1] - o[1], r2 = pl[2] - o[2];
      const float cx = z[1] * r2 - z[2] * r1, cy = z[2] * r0 - z[0] * r2, cz = z[0] * r1 - z[1] * r0;
      J0[i] = sgn * (al[0] * cx + al[1] * cy + al[2] * cz);
      J1[i] = sgn * (bl[0] * z[0] + bl[1] * z[1] + bl[2] * z[2]);
    }
    if (PASS == 0 || i < 6) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (kAxIdx[i] == 2) { // about z: columns 0 and 1 turn
          const float a = R[3 * k], b = R[3 * k + 1];
          R[3 * k] = a * c + b * s;
          R[3 * k + 1] = b * c - a * s;
        } else {              // about y: columns 0 and 2 turn
          const float a = R[3 * k], b = R[3 * k + 2];
          R[3 * k] = a * c - b * s;
          R[3 * k + 2] = a * s + b * c;
        }
      }
    }
  }
  if (PASS == 0) {
    // hand offset along the flange's z only; tool rotation = [[t00, t01, 0], [t10, t11, 0], [0, 0, t22]]; base = diag
    const float *T = K.R_tool[arm], *Bm = K.base_R[arm];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float pf = o[k] + R[3 * k + 2] * K.ee[arm][2];
      pw[k] = K.base_p[arm][k] + Bm[4 * k] * pf;
      Rw[3 * k] = Bm[4 * k] * (R[3 * k] * T[0] + R[3 * k + 1] * T[3]);
      Rw[3 * k + 1] = Bm[4 * k] * (R[3 * k] * T[1] + R[3 * k + 1] * T[4]);
      Rw[3 * k + 2] = Bm[4 * k] * (R[3 * k + 2] * T[8]);
    }
  }
}

template <int PASS, bool STOCK>
__device__ __forceinline__ void chain_sel(const consts_f &K, const int arm, const float *q, float *Rw, float *pw, const float *al,
                                          const float *bl, const float *pl, float sgn, float *J0, float *J1)
{
  if (STOCK) chain_stock_f<PASS>(K, arm, q, Rw, pw, al, bl, pl, sgn, J0, J1);
  else chain_f<PASS>(K, arm, q, Rw, pw, al, bl, pl, sgn, J0, J1);
}

// One Newton round of the scout on the lane's iterate: residual, the reference's loop test (without its precedence quirk: a
// prediction does not need it), and — if any lane of the wave goes on — the analytic step.  Returns whether THIS lane
// updated x; `resid` = the residual is still above tolerance.
template <bool STOCK>
__device__ __forceinline__ bool scout_round(const consts_f &K, float *x, bool active, int iter, bool &resid)
{
    float Rw0[9], pw0[3], Rw1[9], pw1[3];
  chain_sel<0, STOCK>(K, 0, x, Rw0, pw0, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr);
  chain_sel<0, STOCK>(K, 1, x + 7, Rw1, pw1, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr);
  // chain = T2^-1 T1, residual against the initial chain
  float Rc[9], pc[3], qc[4];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) Rc[3 * i + j] = Rw1[i] * Rw0[j] + Rw1[3 + i] * Rw0[3 + j] + Rw1[6 + i] * Rw0[6 + j];
  {
    const float d0 = pw0[0] - pw1[0], d1 = pw0[1] - pw1[1], d2 = pw0[2] - pw1[2];
#pragma unroll
    for (int i = 0; i < 3; i++) pc[i] = Rw1[i] * d0 + Rw1[3 + i] * d1 + Rw1[6 + i] * d2;
  }
  quatf(Rc, qc);
  const float bx = -K.init_q[0], by = -K.init_q[1], bz = -K.init_q[2], bw = K.init_q[3];
  const float dw = qc[3] * bw - qc[0] * bx - qc[1] * by - qc[2] * bz;
  const float dx = qc[3] * bx + qc[0] * bw + qc[1] * bz - qc[2] * by;
  const float dy = qc[3] * by + qc[1] * bw + qc[2] * bx - qc[0] * bz;
  const float dz = qc[3] * bz + qc[2] * bw + qc[0] * by - qc[1] * bx;
  const float vn = sqrtf(dx * dx + dy * dy + dz * dz);
  const float f1 = 2.0f * atan2f(vn, fabsf(dw));
  const float e0 = pc[0] - K.init_p[0], e1 = pc[1] - K.init_p[1], e2 = pc[2] - K.init_p[2];
  const float f0 = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);

  resid = (f0 > K.tol_pos) || (f1 > K.tol_rot);
  const bool cont = active && resid && iter < K.max_iter;
  if (__builtin_amdgcn_ballot_w64(cont) == 0ull) return false;

    // analytic 2x14 Jacobian and the minimum-norm step through the 2x2 Gram matrix
  float u[3] = {0, 0, 0}, n[3] = {0, 0, 0};
  if (f0 > 0.0f) { const float inv = 1.0f / f0; u[0] = e0 * inv; u[1] = e1 * inv; u[2] = e2 * inv; }
  if (vn > 0.0f) { const float sg = (dw < 0.0f ? -1.0f : 1.0f) / vn; n[0] = dx * sg; n[1] = dy * sg; n[2] = dz * sg; }
  float aw[3], bwv[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    aw[k] = Rw1[3 * k] * u[0] + Rw1[3 * k + 1] * u[1] + Rw1[3 * k + 2] * u[2];
    bwv[k] = Rw1[3 * k] * n[0] + Rw1[3 * k + 1] * n[1] + Rw1[3 * k + 2] * n[2];
  }
  float J0[14], J1[14];
#pragma unroll
  for (int arm = 0; arm < 2; arm++) {
    float al[3], bl[3], pl[3];
    const float *Bm = K.base_R[arm];
    const float q0 = pw0[0] - K.base_p[arm][0], q1 = pw0[1] - K.base_p[arm][1], q2 = pw0[2] - K.base_p[arm][2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      al[k] = Bm[k] * aw[0] + Bm[3 + k] * aw[1] + Bm[6 + k] * aw[2];
      bl[k] = Bm[k] * bwv[0] + Bm[3 + k] * bwv[1] + Bm[6 + k] * bwv[2];
      pl[k] = Bm[k] * q0 + Bm[3 + k] * q1 + Bm[6 + k] * q2;
    }
    chain_sel<1, STOCK>(K, arm, x + 7 * arm, nullptr, nullptr, al, bl, pl, arm == 0 ? 1.0f : -1.0f, J0 + 7 * arm, J1 + 7 * arm);
  }
  float ga = 0, gd = 0, gb = 0;
#pragma unroll
  for (int e = 0; e < 14; e++) { ga += J0[e] * J0[e]; gd += J1[e] * J1[e]; gb += J0[e] * J1[e]; }
  const float det = ga * gd - gb * gb;
  float y0 = 0.0f, y1 = 0.0f;
  if (det > 1e-30f) { const float inv = 1.0f / det; y0 = (gd * f0 - gb * f1) * inv; y1 = (ga * f1 - gb * f0) * inv; }
  if (cont) {
#pragma unroll
    for (int e = 0; e < 14; e++) x[e] -= K.step * (J0[e] * y0 + J1[e] * y1);
  }
  return cont;
}

template <int MODE, bool STOCK>
__global__ __launch_bounds__(256) void scout_kernel(const consts_f K, const ccmp_consts KD, const double *__restrict__ q_in,
                                                    uint16_t *__restrict__ pred, unsigned long long B,
                                                    unsigned long long *queue, unsigned long long seed,
                                                    unsigned long long first_index)
{
  float x[14];
  unsigned long long idx = 0, next = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  int iter = 0;
  bool active = false, drained = false;
  (void)queue;
  for (;;) {
    if (!active && !drained) {
#ifdef CCMP_SCOUT_ATOMIC_QUEUE
      const unsigned long long t = atomicAdd(queue, 1ull);
#else
      // static striding instead of a shared queue head: 262144 single-lane dequeues on one word cost more than
      // the whole scout (one word saturates at ~88 dequeues/us); the imbalance of a few samples per lane is small
      const unsigned long long t = next;
      next += (unsigned long long)gridDim.x * blockDim.x;
#endif
      if (t < B) {
        idx = t; active = true; iter = 0;
#pragma unroll
        for (int e = 0; e < 14; e++)
          x[e] = (float)(MODE == 0 ? q_in[idx * 14 + e] : ccmp::ambient_uniform(KD, seed, first_index + idx, e));
      } else drained = true;
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
    bool resid;
    const bool cont = scout_round<STOCK>(K, x, active, iter, resid);
    if (active && !cont) {
      pred[idx] = (uint16_t)iter;
      active = false;
    }
    if (cont) iter++;
  }
}

// ---- extend-step scout: jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:32-96) in single
// precision, one edge per lane, with the scout's Newton round as project().  Its only output is the number of Newton
// rounds (evaluations of the Jacobian-and-residual round of the latency kernel: updates + 1 per projected state) each
// edge needs, capped at `round_cap` — every edge at the cap is simply "long".  The real kernel then takes the edges
// longest-predicted-first, so that a launch of many more edges than resident blocks ends on short edges instead of on
// a 200-round edge that happened to start last (the number of rounds is not predictable from the endpoints' distance:
// it is the conditioning of the projections along the way that makes an edge long).
__device__ __forceinline__ float distf(const float *a, const float *b)
{
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < 14; i++) { const float v = a[i] - b[i]; d += v * v; }
  return sqrtf(d);
}
__device__ __forceinline__ void interpolatef(const float *from, const float *to, float t, float *out)
{
  const float pi = 3.14159265358979f;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    float diff = to[i] - from[i], v;
    if (fabsf(diff) <= pi) v = from[i] + diff * t;
    else {
      diff = diff > 0.f ? 2.f * pi - diff : -2.f * pi - diff;
      v = from[i] - diff * t;
      if (v > pi) v -= 2.f * pi;
      else if (v < -pi) v += 2.f * pi;
    }
    out[i] = v;
  }
}

// ---- two lanes per sample (round 4) --------------------------------------------------------------------------------------
// The scout's run time is its longest lane's: cap x (instructions per round).  Where lanes are plentiful — small and mid-size
// batches, the extend step — a LANE PAIR takes a sample, the even lane arm 0 and the odd lane arm 1: one chain per pass and
// lane instead of two, seven joints of x, J and the update instead of fourteen; the two tool poses, the Gram sums and (extend
// step) the partial distances cross with a quad_perm swap.  Stock structure with twin arms only (both arms read arm 0's chain
// constants; the base frame is selected per lane); everything else keeps the one-lane scout.  Predictions may differ from the
// one-lane scout's in the last bit of a sum — an order, never a result.
__device__ __forceinline__ float pair_swap(float v) // the partner lane's value (lanes 2k <-> 2k + 1)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true));
}
__device__ __forceinline__ bool pair_and(bool v)
{
  return v && __builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true) != 0;
}

struct pair_base { // the lane's arm: base frame (diagonal, to single precision) and its place in the chain T2^-1 T1
  float d[3], p[3], sgn;
  bool second;
};
__device__ __forceinline__ pair_base pair_base_of(const consts_f &K, bool second)
{
  pair_base b;
#pragma unroll
  for (int k = 0; k < 3; k++) { b.d[k] = second ? K.base_R[1][4 * k] : K.base_R[0][4 * k]; b.p[k] = second ? K.base_p[1][k] : K.base_p[0][k]; }
  b.sgn = second ? -1.0f : 1.0f;
  b.second = second;
  return b;
}

// chain_stock_f for the lane's own arm with arm 0's chain constants (twin arms) and the lane's base frame
template <int PASS>
__device__ __forceinline__ void chain_pair_f(const consts_f &K, const pair_base &Bs, const float *q, float *Rw, float *pw, const float *al,
                                             const float *bl, const float *pl, float *J0, float *J1)
{
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 7; i++) {
    float s, c;
    __sincosf(q[i], &s, &c);
    s *= kAxSgn[i];
    const float *off = K.offset[0][i];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (kOffNz[i] & 1) o[k] += R[3 * k] * off[0];
      if (kOffNz[i] & 4) o[k] += R[3 * k + 2] * off[2];
    }
    if (PASS == 1) {
      float z[3];
#pragma unroll
      for (int k = 0; k < 3; k++) z[k] = kAxSgn[i] * R[3 * k + kAxIdx[i]];
      const float r0 = pl[0] - o[0], r1 = pl[1] - o[1], r2 = pl[2] - o[2];
      const float cx = z[1] * r2 - z[2] * r1, cy = z[2] * r0 - z[0] * r2, cz = z[0] * r1 - z[1] * r0;
      J0[i] = Bs.sgn * (al[0] * cx + al[1] * cy + al[2] * cz);
      J1[i] = Bs.sgn * (bl[0] * z[0] + bl[1] * z[1] + bl[2] * z[2]);
    }
    if (PASS == 0 || i < 6) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (kAxIdx[i] == 2) {
          const float a = R[3 * k], b = R[3 * k + 1];
          R[3 * k] = a * c + b * s;
          R[3 * k + 1] = b * c - a * s;
        } else {
          const float a = R[3 * k], b = R[3 * k + 2];
          R[3 * k] = a * c - b * s;
          R[3 * k + 2] = a * s + b * c;
        }
      }
    }
  }
  if (PASS == 0) {
    const float *T = K.R_tool[0];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float pf = o[k] + R[3 * k + 2] * K.ee[0][2];
      pw[k] = Bs.p[k] + Bs.d[k] * pf;
      Rw[3 * k] = Bs.d[k] * (R[3 * k] * T[0] + R[3 * k + 1] * T[3]);
      Rw[3 * k + 1] = Bs.d[k] * (R[3 * k] * T[1] + R[3 * k + 1] * T[4]);
      Rw[3 * k + 2] = Bs.d[k] * (R[3 * k + 2] * T[8]);
    }
  }
}

// scout_round for a lane pair: xa = the seven joints of the lane's arm.  Both lanes of a pair see the same residual (the same
// operations on the same numbers) and therefore take the same decisions.
__device__ __forceinline__ bool scout_round_pair(const consts_f &K, const pair_base &Bs, float *xa, bool active, int iter, bool &resid)
{
  float Ro[9], po[3], Rp[9], pp[3];
  chain_pair_f<0>(K, Bs, xa, Ro, po, nullptr, nullptr, nullptr, nullptr, nullptr);
#pragma unroll
  for (int k = 0; k < 9; k++) Rp[k] = pair_swap(Ro[k]);
#pragma unroll
  for (int k = 0; k < 3; k++) pp[k] = pair_swap(po[k]);
  float Rw0[9], pw0[3], Rw1[9], pw1[3];
#pragma unroll
  for (int k = 0; k < 9; k++) { Rw0[k] = Bs.second ? Rp[k] : Ro[k]; Rw1[k] = Bs.second ? Ro[k] : Rp[k]; }
#pragma unroll
  for (int k = 0; k < 3; k++) { pw0[k] = Bs.second ? pp[k] : po[k]; pw1[k] = Bs.second ? po[k] : pp[k]; }
  float Rc[9], pc[3], qc[4];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) Rc[3 * i + j] = Rw1[i] * Rw0[j] + Rw1[3 + i] * Rw0[3 + j] + Rw1[6 + i] * Rw0[6 + j];
  {
    const float d0 = pw0[0] - pw1[0], d1 = pw0[1] - pw1[1], d2 = pw0[2] - pw1[2];
#pragma unroll
    for (int i = 0; i < 3; i++) pc[i] = Rw1[i] * d0 + Rw1[3 + i] * d1 + Rw1[6 + i] * d2;
  }
  quatf(Rc, qc);
  const float bx = -K.init_q[0], by = -K.init_q[1], bz = -K.init_q[2], bw = K.init_q[3];
  const float dw = qc[3] * bw - qc[0] * bx - qc[1] * by - qc[2] * bz;
  const float dx = qc[3] * bx + qc[0] * bw + qc[1] * bz - qc[2] * by;
  const float dy = qc[3] * by + qc[1] * bw + qc[2] * bx - qc[0] * bz;
  const float dz = qc[3] * bz + qc[2] * bw + qc[0] * by - qc[1] * bx;
  const float vn = sqrtf(dx * dx + dy * dy + dz * dz);
  const float f1 = 2.0f * atan2f(vn, fabsf(dw));
  const float e0 = pc[0] - K.init_p[0], e1 = pc[1] - K.init_p[1], e2 = pc[2] - K.init_p[2];
  const float f0 = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  resid = (f0 > K.tol_pos) || (f1 > K.tol_rot);
  const bool cont = active && resid && iter < K.max_iter;
  if (__builtin_amdgcn_ballot_w64(cont) == 0ull) return false;

  float u[3] = {0, 0, 0}, n[3] = {0, 0, 0};
  if (f0 > 0.0f) { const float inv = 1.0f / f0; u[0] = e0 * inv; u[1] = e1 * inv; u[2] = e2 * inv; }
  if (vn > 0.0f) { const float sg = (dw < 0.0f ? -1.0f : 1.0f) / vn; n[0] = dx * sg; n[1] = dy * sg; n[2] = dz * sg; }
  float al[3], bl[3], pl[3]; // the probes in the lane's own base frame (diagonal base: B^T v = d * v)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float aw = Rw1[3 * k] * u[0] + Rw1[3 * k + 1] * u[1] + Rw1[3 * k + 2] * u[2];
    const float bv = Rw1[3 * k] * n[0] + Rw1[3 * k + 1] * n[1] + Rw1[3 * k + 2] * n[2];
    al[k] = Bs.d[k] * aw;
    bl[k] = Bs.d[k] * bv;
    pl[k] = Bs.d[k] * (pw0[k] - Bs.p[k]);
  }
  float J0[7], J1[7];
  chain_pair_f<1>(K, Bs, xa, nullptr, nullptr, al, bl, pl, J0, J1);
  float ga = 0, gd = 0, gb = 0;
#pragma unroll
  for (int e = 0; e < 7; e++) { ga += J0[e] * J0[e]; gd += J1[e] * J1[e]; gb += J0[e] * J1[e]; }
  // the same three numbers in both lanes: own + partner's, summed in arm order
  const float pa = pair_swap(ga), pd = pair_swap(gd), pb = pair_swap(gb);
  ga = Bs.second ? pa + ga : ga + pa;
  gd = Bs.second ? pd + gd : gd + pd;
  gb = Bs.second ? pb + gb : gb + pb;
  const float det = ga * gd - gb * gb;
  float y0 = 0.0f, y1 = 0.0f;
  if (det > 1e-30f) { const float inv = 1.0f / det; y0 = (gd * f0 - gb * f1) * inv; y1 = (ga * f1 - gb * f0) * inv; }
  if (cont) {
#pragma unroll
    for (int e = 0; e < 7; e++) xa[e] -= K.step * (J0[e] * y0 + J1[e] * y1);
  }
  return cont;
}

template <int MODE>
__global__ __launch_bounds__(256) void scout_pair_kernel(const consts_f K, const ccmp_consts KD, const double *__restrict__ q_in,
                                                         uint16_t *__restrict__ pred, unsigned long long B, unsigned long long seed,
                                                         unsigned long long first_index)
{
  const unsigned long long gtid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = (gtid & 1ull) != 0ull;
  const pair_base Bs = pair_base_of(K, second);
  float xa[7];
  unsigned long long idx = 0, next = gtid >> 1;
  int iter = 0;
  bool active = false, drained = false;
  for (;;) {
    if (!active && !drained) {
      const unsigned long long t = next;
      next += ((unsigned long long)gridDim.x * blockDim.x) >> 1;
      if (t < B) {
        idx = t; active = true; iter = 0;
#pragma unroll
        for (int e = 0; e < 7; e++) {
          const int j = e + (second ? 7 : 0);
          xa[e] = (float)(MODE == 0 ? q_in[idx * 14 + j] : ccmp::ambient_uniform(KD, seed, first_index + idx, j));
        }
      } else drained = true;
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
    bool resid;
    const bool cont = scout_round_pair(K, Bs, xa, active, iter, resid);
    if (active && !cont) {
      if (!second) pred[idx] = (uint16_t)iter;
      active = false;
    }
    if (cont) iter++;
  }
}

// the extend-step scout on lane pairs: each lane keeps its arm's half of x, previous and target; the distances are sums of two
// partial sums (arm order)
__device__ __forceinline__ float pair_dist(const pair_base &Bs, const float *a, const float *b)
{
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < 7; i++) { const float v = a[i] - b[i]; d += v * v; }
  const float o = pair_swap(d);
  return sqrtf(Bs.second ? o + d : d + o);
}
__device__ __forceinline__ void interpolate7f(const float *from, const float *to, float t, float *out)
{
  const float pi = 3.14159265358979f;
#pragma unroll
  for (int i = 0; i < 7; i++) {
    float diff = to[i] - from[i], v;
    if (fabsf(diff) <= pi) v = from[i] + diff * t;
    else {
      diff = diff > 0.f ? 2.f * pi - diff : -2.f * pi - diff;
      v = from[i] - diff * t;
      if (v > pi) v -= 2.f * pi;
      else if (v < -pi) v += 2.f * pi;
    }
    out[i] = v;
  }
}

__global__ __launch_bounds__(64) void scout_geodesic_pair_kernel(const consts_f K, const double *__restrict__ from, const double *__restrict__ to,
                                                                 unsigned long long E, float delta, float lambda, int max_states, int round_cap,
                                                                 uint16_t *__restrict__ pred)
{
  const unsigned long long gtid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = (gtid & 1ull) != 0ull;
  const pair_base Bs = pair_base_of(K, second);
  const int h = second ? 7 : 0;
  float x[7], prev[7], tgt[7];
  unsigned long long idx = 0, next = gtid >> 1;
  int iter = 0, rounds = 0, n = 1;
  float dist = 0.f, total = 0.f, maxd = 0.f;
  bool active = false, drained = false;
  for (;;) {
    if (!active && !drained) {
      const unsigned long long t = next;
      next += ((unsigned long long)gridDim.x * blockDim.x) >> 1;
      if (t < E) {
        idx = t;
#pragma unroll
        for (int e = 0; e < 7; e++) { prev[e] = (float)from[idx * 14 + h + e]; tgt[e] = (float)to[idx * 14 + h + e]; }
        dist = pair_dist(Bs, prev, tgt);
        if (dist > delta) {
          active = true; iter = 0; rounds = 0; n = 1; total = 0.f; maxd = dist * lambda;
          interpolate7f(prev, tgt, delta / dist, x);
        } else if (!second) {
          pred[idx] = 0; // already there (or not a number): nothing to traverse
        }
      } else drained = true;
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) {
      if (__builtin_amdgcn_ballot_w64(!drained) == 0ull) break;
      continue;
    }
    bool resid;
    const bool cont = scout_round_pair(K, Bs, x, active, iter, resid);
    // the distances below cross lanes: every lane of the wave goes through them, inactive pairs on stale values
    bool inside = true;
#pragma unroll
    for (int e = 0; e < 7; e++) inside = inside && x[e] >= K.lbe[e] && x[e] <= K.ube[e];
    inside = pair_and(inside);
    const float step = pair_dist(Bs, prev, x), nd = pair_dist(Bs, x, tgt);
    if (!active) continue;
    rounds++;
    if (cont) iter++;
    bool done = rounds >= round_cap;
    if (!cont && !done) { // project() of this state has returned: the reference's break tests, then the next state
      done = resid || !inside;
      if (!done) {
        total += step;
        done = step > lambda * delta || total > maxd || nd >= dist || ++n > max_states || !(nd >= delta);
        if (!done) {
          dist = nd;
#pragma unroll
          for (int e = 0; e < 7; e++) prev[e] = x[e];
          interpolate7f(prev, tgt, delta / dist, x);
          iter = 0;
        }
      }
    }
    if (done) {
      if (!second) pred[idx] = (uint16_t)rounds;
      active = false;
    }
  }
}

template <bool STOCK>
__global__ __launch_bounds__(64) void scout_geodesic_kernel(const consts_f K, const double *__restrict__ from, const double *__restrict__ to,
                                                            unsigned long long E, float delta, float lambda, int max_states, int round_cap,
                                                            uint16_t *__restrict__ pred)
{
  float x[14], prev[14], tgt[14];
  unsigned long long idx = 0, next = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  int iter = 0, rounds = 0, n = 1;
  float dist = 0.f, total = 0.f, maxd = 0.f;
  bool active = false, drained = false;
  for (;;) {
    if (!active && !drained) {
      const unsigned long long t = next;
      next += (unsigned long long)gridDim.x * blockDim.x;
      if (t < E) {
        idx = t;
#pragma unroll
        for (int e = 0; e < 14; e++) { prev[e] = (float)from[idx * 14 + e]; tgt[e] = (float)to[idx * 14 + e]; }
        dist = distf(prev, tgt);
        if (dist > delta) {
          active = true; iter = 0; rounds = 0; n = 1; total = 0.f; maxd = dist * lambda;
          interpolatef(prev, tgt, delta / dist, x);
        } else {
          pred[idx] = 0; // already there (or not a number): nothing to traverse
        }
      } else drained = true;
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) {
      if (__builtin_amdgcn_ballot_w64(!drained) == 0ull) break;
      continue;
    }
    bool resid;
    const bool cont = scout_round<STOCK>(K, x, active, iter, resid);
    if (!active) continue;
    rounds++;
    if (cont) iter++;
    bool done = rounds >= round_cap;
    if (!cont && !done) { // project() of this state has returned: the reference's break tests, then the next state
      bool ok = !resid;
#pragma unroll
      for (int e = 0; e < 14; e++) ok = ok && x[e] >= K.lbe[e % 7] && x[e] <= K.ube[e % 7];
      done = !ok;
      if (!done) {
        const float step = distf(prev, x);
        total += step;
        const float nd = distf(x, tgt);
        done = step > lambda * delta || total > maxd || nd >= dist || ++n > max_states || !(nd >= delta);
        if (!done) {
          dist = nd;
#pragma unroll
          for (int e = 0; e < 14; e++) prev[e] = x[e];
          interpolatef(prev, tgt, delta / dist, x);
          iter = 0;
        }
      }
    }
    if (done) {
      pred[idx] = (uint16_t)rounds;
      active = false;
    }
  }
}

constexpr int kBins = 1024;
#ifndef CCMP_SORT_FUSED_MAX
#define CCMP_SORT_FUSED_MAX 16384
#endif
constexpr int kSortFusedMax = CCMP_SORT_FUSED_MAX; // up to this many keys the counting sort is one launch (sort_fused_kernel)

__global__ void hist_kernel(const uint16_t *__restrict__ pred, unsigned long long B, unsigned int *__restrict__ hist)
{
  __shared__ unsigned int h[kBins];
  for (int k = threadIdx.x; k < kBins; k += blockDim.x) h[k] = 0;
  __syncthreads();
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < B;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    unsigned int key = pred[i];
    atomicAdd(&h[key < kBins ? key : kBins - 1], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kBins; k += blockDim.x)
    if (h[k]) atomicAdd(&hist[k], h[k]);
}

// The cut of the order for a split launch (ccmp_split.h), by ONE wavefront of the sort's own kernel: ge[k] = number of keys >= k for
// every k < kBins (LDS).  Kind 1 is fd_split_kernel's rule (below: it remains for the option sets that fix the
// cut themselves), kind 2 the extend step's default (described at geo_split_kernel).
__device__ __forceinline__ void apply_split(const unsigned int *ge, const ccmp_split_req &req, int lane)
{
  unsigned long long front = 0;
  if (req.kind == 1) {
    const unsigned int c = ge[req.p_low < kBins ? req.p_low : kBins - 1];
    front = c < req.limit ? c : req.limit;
  } else {
    const int p_high = req.p_high > kBins - 1 ? kBins - 1 : req.p_high;
    unsigned long long all = 0, above = 0;
    for (int j = lane; j < kBins; j += 64) {
      const unsigned long long v = j >= 1 ? ge[j] : 0u;
      all += v;
      if (j > p_high) above += v;
    }
    for (int off = 32; off > 0; off >>= 1) {
      all += __shfl_down(all, off);
      above += __shfl_down(above, off);
    }
    all = __shfl(all, 0);
    above = __shfl(above, 0);
    const bool heavy = ((unsigned long long)p_high * ge[p_high] + above) * 1000ull >= all * (unsigned long long)req.permille;
    front = ge[heavy ? p_high : req.p_low];
  }
  const int n = req.clear > 5 ? req.clear : 5;
  if (lane < n && (lane < req.clear || lane == 0 || lane == 3 || lane == 4))
    req.queue[lane] = (lane == 0 || lane == 3 || lane == 4) ? front : 0ull;
}

// exclusive scan in DESCENDING key order: base[k] = number of samples predicted longer than k.  One wavefront: lane l
// owns the l-th run of ceil(nbins / 64) bins from the top, sums it, the 64 sums are scanned with shuffles, every lane
// writes its run's bases.  (A single thread walking 1024 bins took 17 us as a compile-time loop and 98 us once the bin
// count became a run-time argument: dependent global round trips.)
__global__ __launch_bounds__(64) void scan_desc_kernel(unsigned int *__restrict__ hist /* in: counts, out: running cursor = base */, int nbins,
                                                       const ccmp_split_req split)
{
  __shared__ unsigned int ge[kBins]; // keys >= k (what hist holds once the scatter has run), for the split
  const int lane = threadIdx.x;
  if (split.queue)
    for (int k = lane; k < kBins; k += 64) ge[k] = 0;
  __syncthreads();
  const int per = (nbins + 63) / 64; // <= 16 (nbins <= kBins)
  unsigned int c[16], sum = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int k = nbins - 1 - (lane * per + i);
    c[i] = (i < per && k >= 0) ? hist[k] : 0u;
    sum += c[i];
  }
  unsigned int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int v = (unsigned int)__shfl_up((int)incl, off);
    if (lane >= off) incl += v;
  }
  unsigned int run = incl - sum;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int k = nbins - 1 - (lane * per + i);
    if (i < per && k >= 0) { hist[k] = run; run += c[i]; if (split.queue) ge[k] = run; }
  }
  if (split.queue) {
    __syncthreads();
    apply_split(ge, split, lane);
  }
}

// Counting-sort scatter with block-local ranking: LDS histogram of the block's keys, ONE global atomic per
// (block, non-empty bin) to reserve a range, LDS atomics for the rank inside the range.
constexpr int kScatterPerThread = 4;
__global__ __launch_bounds__(256) void scatter_kernel(const uint16_t *__restrict__ pred, unsigned long long B,
                                                      unsigned int *__restrict__ cursor, unsigned int *__restrict__ order)
{
  __shared__ unsigned int cnt[kBins], base[kBins];
  for (int k = threadIdx.x; k < kBins; k += 256) cnt[k] = 0;
  __syncthreads();
  const unsigned long long i0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * kScatterPerThread;
  unsigned int key[kScatterPerThread];
#pragma unroll
  for (int k = 0; k < kScatterPerThread; k++) {
    const unsigned long long i = i0 + k;
    key[k] = 0xffffffffu;
    if (i < B) {
      unsigned int v = pred[i];
      key[k] = v < kBins ? v : kBins - 1;
      atomicAdd(&cnt[key[k]], 1u);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kBins; k += 256) {
    const unsigned int c = cnt[k];
    base[k] = c ? atomicAdd(&cursor[k], c) : 0u;
    cnt[k] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kScatterPerThread; k++)
    if (key[k] != 0xffffffffu) order[base[key[k]] + atomicAdd(&cnt[key[k]], 1u)] = (unsigned int)(i0 + k);
}

// The whole descending counting sort in ONE launch for small batches (round 4): histogram, scan and scatter of up to a few ten
// thousand keys are microseconds of work each, and three launches (plus the one that clears the histogram) cost more than
// that in dispatch alone — a tenth of a 4 096-sample call was launch gaps.  One 1024-thread block: LDS histogram, thread t
// owns bin kBins - 1 - t for the descending exclusive scan (wave scan + sixteen wave sums), LDS-atomic ranks for the scatter.
// Leaves in hist what the three-kernel form leaves: at k the number of keys >= k (the split kernels read it).
__global__ __launch_bounds__(1024) void sort_fused_kernel(const uint16_t *__restrict__ pred, unsigned int B, unsigned int *__restrict__ hist,
                                                          unsigned int *__restrict__ order, const ccmp_split_req split)
{
  __shared__ unsigned int cnt[kBins], base[kBins], ge[kBins], wsum[16];
  const int t = threadIdx.x, lane = t & 63;
  cnt[t] = 0;
  __syncthreads();
  for (unsigned int i = t; i < B; i += 1024) {
    const unsigned int v = pred[i];
    atomicAdd(&cnt[v < kBins ? v : kBins - 1], 1u);
  }
  __syncthreads();
  const int k = kBins - 1 - t;
  const unsigned int c = cnt[k];
  unsigned int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int v = (unsigned int)__shfl_up((int)incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wsum[t >> 6] = incl;
  __syncthreads();
  unsigned int above = 0;
  for (int w = 0; w < (t >> 6); w++) above += wsum[w];
  const unsigned int excl = above + incl - c; // keys in the bins above k
  base[k] = excl;
  hist[k] = excl + c;
  ge[k] = excl + c;
  cnt[k] = 0;
  __syncthreads();
  if (split.queue && t < 64) apply_split(ge, split, t); // the first wavefront decides the cut, then scatters with the others
  for (unsigned int i = t; i < B; i += 1024) {
    const unsigned int v = pred[i];
    const unsigned int key = v < kBins ? v : kBins - 1;
    order[base[key] + atomicAdd(&cnt[key], 1u)] = i;
  }
}
static_assert(kBins == 1024, "sort_fused_kernel: one thread per bin");

// after scatter_kernel the cursor array holds, at k, the number of samples predicted >= k iterations: the front of
// the order that a split launch gives to the latency kernel = those predicted >= pred_min, at most `limit`
__global__ void split_kernel(const unsigned int *__restrict__ hist, int pred_min, unsigned int limit, unsigned int *__restrict__ out)
{
  if (threadIdx.x == 0) {
    const unsigned int n = hist[pred_min < kBins ? pred_min : kBins - 1];
    *out = n < limit ? n : limit;
  }
}

// The split launch of the reference-arithmetic path (mid-size batches, ccmp_api.cpp): the front of the descending order — the
// samples predicted >= pred_min iterations, at most `limit` — goes to latency blocks BESIDE the throughput kernel.  queue[4]:
// the front's length (the latency blocks' ticket limit); queue[0]: the throughput kernel's queue starts behind the front;
// queue[3]: its count of finished samples starts there too (its occupancy rule counts B minus that as "in flight").
__global__ void fd_split_kernel(const unsigned int *__restrict__ hist, int pred_min, unsigned int limit, unsigned long long *__restrict__ queue)
{
  if (threadIdx.x == 0) {
    const unsigned int c = hist[pred_min < kBins ? pred_min : kBins - 1];
    const unsigned long long n = c < limit ? c : limit;
    queue[4] = n;
    queue[0] = n;
    queue[3] = n;
  }
}

// (The default cut of a bulk extend call — one of TWO, by what the batch looks like: where the edges the scout's cap cut off,
// predicted >= p_high rounds, already carry permille / 1000 of the predicted work (stefan, dumbbell: 17 %) they are the front;
// where they do not (Wine_Bottle: 5 %) the front starts at p_low rounds, or its blocks would idle while the group kernel's longest
// edges run at a twelfth of their pace — is apply_split's kind 2 above: hist[k] = edges predicted >= k rounds, the predicted work
// is sum_{j >= 1} hist[j], that of the edges predicted >= P is P * hist[P] + sum_{j > P} hist[j].)

// Bulk extend calls (ccmp_api.cpp: geodesic_common): where to cut the scout's descending order between the latency blocks (front)
// and the throughput layout (rest).  hist[k] = number of edges predicted >= k rounds (hist[0] = all), so the predicted work of
// the edges predicted >= P is P * hist[P] + sum_{j > P} hist[j] and all of it is sum_{j >= 1} hist[j].  The cut is the LARGEST
// P <= p_max whose front carries at least permille / 1000 of the predicted work — the front's share follows the batch's own
// distribution (stefan's edges need more rounds than Wine_Bottle's), p_max bounds the longest edge the slow-per-round layout
// is given.  queue[4] = front length, queue[0] = the group kernel's first ticket, as fd_split_kernel leaves them.
__global__ __launch_bounds__(64) void geo_split_kernel(const unsigned int *__restrict__ hist, int p_min, int p_max, int permille,
                                                       unsigned long long *__restrict__ queue)
{
  __shared__ unsigned int n[kBins];
  for (int k = threadIdx.x; k < kBins; k += 64) n[k] = hist[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long all = 0;
    for (int j = 1; j < kBins; j++) all += n[j];
    if (p_max > kBins - 1) p_max = kBins - 1;
    unsigned long long above = 0; // sum_{j > P} hist[j]
    for (int j = kBins - 1; j > p_max; j--) above += n[j];
    int P = p_max;
    for (; P > p_min; P--) {
      if (((unsigned long long)P * n[P] + above) * 1000ull >= all * (unsigned long long)permille) break;
      above += n[P];
    }
    const unsigned long long front = n[P];
    queue[4] = front;
    queue[0] = front;
    queue[3] = front; // the group kernel's count of finished edges starts there (its hand-over rule)
  }
}

} // namespace

extern "C" hipError_t ccmp_launch_clear_words(void *words, size_t n_u32, hipStream_t st); // ccmp_kernels_fd.hip

extern "C" hipError_t ccmp_launch_geo_split(const unsigned int *hist, int p_min, int p_max, int permille, unsigned long long *queue, hipStream_t st)
{
  hipLaunchKernelGGL(geo_split_kernel, dim3(1), dim3(64), 0, st, hist, p_min, p_max, permille, queue);
  return hipGetLastError();
}

extern "C" hipError_t ccmp_launch_fd_split(const unsigned int *hist, int pred_min, unsigned int limit, unsigned long long *queue, hipStream_t st)
{
  hipLaunchKernelGGL(fd_split_kernel, dim3(1), dim3(64), 0, st, hist, pred_min, limit, queue);
  return hipGetLastError();
}

extern "C" hipError_t ccmp_launch_split_count(const unsigned int *hist, int pred_min, unsigned int limit, unsigned int *out, hipStream_t st)
{
  hipLaunchKernelGGL(split_kernel, dim3(1), dim3(64), 0, st, hist, pred_min, limit, out);
  return hipGetLastError();
}

// single-precision copy of the kernel constants; *stock = the model looks like the stock Panda to single precision (axes
// on coordinate axes, the expected zero offsets, block-diagonal tool rotation, diagonal base rotation) — only the
// prediction depends on it
static void make_consts_f(const ccmp_consts *K, consts_f &F, bool *stock_out)
{
  for (int a = 0; a < 2; a++) {
    for (int i = 0; i < 7; i++)
      for (int k = 0; k < 3; k++) { F.axis[a][i][k] = (float)K->axis[a][i][k]; F.offset[a][i][k] = (float)K->offset[a][i][k]; }
    for (int k = 0; k < 3; k++) { F.ee[a][k] = (float)K->ee[a][k]; F.base_p[a][k] = (float)K->base_p[a][k]; }
    for (int k = 0; k < 9; k++) { F.R_tool[a][k] = (float)K->R_tool[a][k]; F.base_R[a][k] = (float)K->base_R[a][k]; }
  }
  for (int k = 0; k < 3; k++) F.init_p[k] = (float)K->init_p[k];
  for (int k = 0; k < 4; k++) F.init_q[k] = (float)K->init_q[k];
  F.tol_pos = (float)K->tol_pos; F.tol_rot = (float)K->tol_rot; F.step = (float)K->step;
  F.max_iter = K->max_iter < kScoutCap ? K->max_iter : kScoutCap;
  for (int i = 0; i < 7; i++) { F.lbe[i] = (float)K->lbe[i]; F.ube[i] = (float)K->ube[i]; }
  // does the model look like the stock Panda to single precision?  (axes on coordinate axes, the expected zero offsets,
  // block-diagonal tool rotation, diagonal base rotation)  Only the prediction depends on it.
  bool stock = true;
  const float tol = 1e-6f;
  for (int a = 0; a < 2 && stock; a++) {
    for (int i = 0; i < 7; i++) {
      for (int k = 0; k < 3; k++) {
        const float want = k == kAxIdx[i] ? kAxSgn[i] : 0.0f;
        if (!(fabsf(F.axis[a][i][k] - want) < tol)) stock = false;
        if (!((kOffNz[i] >> k) & 1) && !(fabsf(F.offset[a][i][k]) < tol)) stock = false;
      }
    }
    if (!(fabsf(F.ee[a][0]) < tol && fabsf(F.ee[a][1]) < tol)) stock = false;
    const int zero_t[4] = {2, 5, 6, 7};
    for (int k = 0; k < 4; k++)
      if (!(fabsf(F.R_tool[a][zero_t[k]]) < tol)) stock = false;
    for (int k = 0; k < 9; k++)
      if (k % 4 != 0 && !(fabsf(F.base_R[a][k]) < tol)) stock = false;
  }
  *stock_out = stock;
}

extern "C" hipError_t ccmp_launch_scout_order(const ccmp_consts *K, int mode, const double *q_in, size_t B, uint16_t *pred,
                                              unsigned int *hist, unsigned int *order, unsigned long long *queue,
                                              unsigned long long seed, unsigned long long first, int nblocks, int pair_max_blocks,
                                              const ccmp_split_req *split, hipStream_t st)
{
  const ccmp_split_req sr = split ? *split : ccmp_split_req{nullptr, 0, 0, 0, 0, 0u, 0};
  consts_f F;
  bool stock;
  make_consts_f(K, F, &stock);
  const bool fused = B <= (size_t)kSortFusedMax; // the fused sort writes every bin itself: nothing to clear
  hipError_t e = hipSuccess;
#ifdef CCMP_SCOUT_ATOMIC_QUEUE
  e = ccmp_launch_clear_words(queue, 2, st); // kernels, so that a stream capture replays them
  if (e != hipSuccess) return e;
#endif
  if (!fused) {
    e = ccmp_launch_clear_words(hist, kBins, st);
    if (e != hipSuccess) return e;
  }
#define CCMP_LAUNCH_SCOUT(MODE, STOCK) \
  hipLaunchKernelGGL((scout_kernel<MODE, STOCK>), dim3(nblocks), dim3(256), 0, st, F, *K, q_in, pred, (unsigned long long)B, queue, seed, first)
  // two lanes per sample while every sample still gets its pair at once (pair_max_blocks blocks of 128 pairs); larger batches
  // keep one lane per sample, several samples per lane
  const size_t pair_blocks = (2 * B + 255) / 256;
  if (stock && K->twin_arms && pair_max_blocks > 0 && pair_blocks <= (size_t)pair_max_blocks) {
    if (mode == 0) hipLaunchKernelGGL(scout_pair_kernel<0>, dim3((unsigned)pair_blocks), dim3(256), 0, st, F, *K, q_in, pred, (unsigned long long)B, seed, first);
    else hipLaunchKernelGGL(scout_pair_kernel<1>, dim3((unsigned)pair_blocks), dim3(256), 0, st, F, *K, q_in, pred, (unsigned long long)B, seed, first);
  } else
  if (mode == 0) {
    if (stock) CCMP_LAUNCH_SCOUT(0, true);
    else CCMP_LAUNCH_SCOUT(0, false);
  } else {
    if (stock) CCMP_LAUNCH_SCOUT(1, true);
    else CCMP_LAUNCH_SCOUT(1, false);
  }
#undef CCMP_LAUNCH_SCOUT
  if (fused) {
    hipLaunchKernelGGL(sort_fused_kernel, dim3(1), dim3(1024), 0, st, pred, (unsigned int)B, hist, order, sr);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(hist_kernel, dim3(256), dim3(256), 0, st, pred, (unsigned long long)B, hist);
  hipLaunchKernelGGL(scan_desc_kernel, dim3(1), dim3(64), 0, st, hist, kBins, sr);
  hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((B + 256 * kScatterPerThread - 1) / (256 * kScatterPerThread))), dim3(256), 0, st, pred,
                     (unsigned long long)B, hist, order);
  return hipGetLastError();
}

// extend-step order: FP32 scout of every edge (rounds capped at round_cap < 1024), then the descending counting sort
extern "C" hipError_t ccmp_launch_geodesic_scout_order(const ccmp_consts *K, const double *from, const double *to, size_t E, double delta,
                                                        double lambda, int max_states, int round_cap, uint16_t *pred, unsigned int *hist,
                                                        unsigned int *order, int pairs, const ccmp_split_req *split, hipStream_t st)
{
  const ccmp_split_req sr = split ? *split : ccmp_split_req{nullptr, 0, 0, 0, 0, 0u, 0};
  consts_f F;
  bool stock;
  make_consts_f(K, F, &stock);
  if (round_cap > kBins - 1) round_cap = kBins - 1;
  F.max_iter = K->max_iter < round_cap ? K->max_iter : round_cap;
  const int nbins = round_cap + 1;
  const bool fused = E <= (size_t)kSortFusedMax;
  hipError_t e = hipSuccess;
  if (!fused) {
    // every bin, not only the nbins this sort fills: the split rules sum the histogram over all kBins, and the bins above were
    // whatever the context's last sort left there (a projector batch's counts up to its cap of 96: a cut decided on them)
    e = ccmp_launch_clear_words(hist, (size_t)kBins, st);
    if (e != hipSuccess) return e;
  }
  const unsigned blocks = (unsigned)((E + 63) / 64);
  if (stock && K->twin_arms && pairs)
    hipLaunchKernelGGL(scout_geodesic_pair_kernel, dim3((unsigned)((2 * E + 63) / 64)), dim3(64), 0, st, F, from, to, (unsigned long long)E,
                       (float)delta, (float)lambda, max_states, round_cap, pred);
  else if (stock)
    hipLaunchKernelGGL(scout_geodesic_kernel<true>, dim3(blocks), dim3(64), 0, st, F, from, to, (unsigned long long)E, (float)delta,
                       (float)lambda, max_states, round_cap, pred);
  else
    hipLaunchKernelGGL(scout_geodesic_kernel<false>, dim3(blocks), dim3(64), 0, st, F, from, to, (unsigned long long)E, (float)delta,
                       (float)lambda, max_states, round_cap, pred);
  if (fused) {
    hipLaunchKernelGGL(sort_fused_kernel, dim3(1), dim3(1024), 0, st, pred, (unsigned int)E, hist, order, sr);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(hist_kernel, dim3(64), dim3(256), 0, st, pred, (unsigned long long)E, hist);
  hipLaunchKernelGGL(scan_desc_kernel, dim3(1), dim3(64), 0, st, hist, nbins, sr);
  hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((E + 256 * kScatterPerThread - 1) / (256 * kScatterPerThread))), dim3(256), 0, st, pred,
                     (unsigned long long)E, hist, order);
  return hipGetLastError();
}
