// ccmp_kernels_scene.hip — proxy-geometry clearance (include/ccmp.h: ccmp_clearance_batch): the pre-filter a planner puts
// in front of KinematicChainValidityChecker::isValid (src/kinematics/KinematicChain.cpp:94-123, MoveIt on the host).
// Canonical rounding model (-ffp-contract=off -DCCMP_USE_FMA): bit-identical to oracle/ccmp_oracle.c:orc_clearance.
//
// One 256-thread block per tile of 64 states; lane l of every wavefront works on state l of the tile.
//   phase 1  wavefront w walks the chain of arm (w & 1) — the projector's own FK (ccmp_kin.h: joint_step, the general
//            form; the STOCK specialisations produce the same bits) — and, the moment a body frame exists, places the
//            spheres attached to it whose stored index has parity (w >> 1): world centre = t_wb (o + R c).  Spheres are
//            stored sorted by frame, so "the spheres of this frame" is a scalar index range and every load of proxy
//            data is a scalar load.  Centres go to LDS as [sphere][component][lane]: conflict-free.
//   phase 2  wavefront w takes the tested pairs p = w, w + 4, … of the scene's list (scalar indices, LDS reads at
//            lane-consecutive addresses), keeps the smallest signed distance and the first pair attaining it.
//   phase 3  the four partial minima meet in LDS; wavefront 0 writes clearance / pair / flag.
// HBM traffic: 112 B read + 13 B written per state; the work is the pair loop (one IEEE square root per pair).
#include "ccmp_fd_common.h"
#include "ccmp_scene.h"

using namespace ccmp;

namespace {

constexpr int kTile = 64;

// spheres [b, e) with stored-index parity `half`, attached to the frame (R, o) of arm `arm` (in the arm's base)
__device__ __forceinline__ void place_spheres(const ccmp_consts &K, const scene_dev *__restrict__ S, int arm, int b, int e, int half,
                                              const double *R, const double *o, double *cen, int lane)
{
  for (int s = b; s < e; s++) {
    if ((s & 1) != half) continue;
    const double c[3] = {S->c[s][0], S->c[s][1], S->c[s][2]};
    double v[3] = {o[0], o[1], o[2]};
    mulvec_acc(R, c, v);
    double wv[3] = {K.base_p[arm][0], K.base_p[arm][1], K.base_p[arm][2]};
    mulvec_acc(K.base_R[arm], v, wv);
#pragma unroll
    for (int k = 0; k < 3; k++) cen[(s * 3 + k) * kTile + lane] = wv[k];
  }
}

__global__ __launch_bounds__(256) void clearance_kernel(const ccmp_consts K, const scene_dev *__restrict__ S, const double *__restrict__ q,
                                                        const uint8_t *__restrict__ ok_in, unsigned long long B, double margin,
                                                        double *__restrict__ clearance, int32_t *__restrict__ pair,
                                                        uint8_t *__restrict__ free_out)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int ns = S->n_spheres, np = S->n_pairs;
  double *cen = lds;                               // [ns][3][64]
  double *red = lds + (size_t)ns * 3 * kTile;      // [4][64] partial minima
  int *redp = reinterpret_cast<int *>(red + 4 * kTile); // [4][64] their positions in the pair list
  int *bad = redp + 4 * kTile;                     // [4][64] non-finite joint value seen
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int arm = w & 1, half = w >> 1;
  const unsigned long long tiles = (B + kTile - 1) / kTile;

  for (unsigned long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const unsigned long long idx = tile * kTile + lane;
    const bool live = idx < B;
    const unsigned long long src = live ? idx : B - 1; // idle lanes repeat the last state and write nothing
    // ---- phase 1: this wavefront's arm ---------------------------------------------------------------------------
    {
      double x[7];
      bool nonfinite = false;
#pragma unroll
      for (int i = 0; i < 7; i++) {
        x[i] = q[src * 14 + arm * 7 + i];
        if (!(x[i] - x[i] == 0.0)) nonfinite = true;
      }
      bad[w * kTile + lane] = nonfinite ? 1 : 0;
      double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
#pragma unroll
      for (int i = 0; i < 7; i++) {
        double s, c;
        ccmp_sincos(x[i], &s, &c);
        joint_step(K, arm, i, s, c, R, o);
        const int slot = arm * 9 + i;
        place_spheres(K, S, arm, S->slot_begin[slot], S->slot_begin[slot + 1], half, R, o, cen, lane);
      }
      { // hand frame: PandaModel::getTranslation / getRotation (ccmp_kin.h: tool_pose_t, before t_wb)
        double oh[3] = {o[0], o[1], o[2]}, Rh[9];
        mulvec_acc(R, K.ee[arm], oh);
        mul33(R, K.R_tool[arm], Rh);
        const int slot = arm * 9 + 7;
        place_spheres(K, S, arm, S->slot_begin[slot], S->slot_begin[slot + 1], half, Rh, oh, cen, lane);
      }
      { // the arm's base: only t_wb applies
        const int slot = arm * 9 + 8;
        for (int s = S->slot_begin[slot]; s < S->slot_begin[slot + 1]; s++) {
          if ((s & 1) != half) continue;
          const double c[3] = {S->c[s][0], S->c[s][1], S->c[s][2]};
          double wv[3] = {K.base_p[arm][0], K.base_p[arm][1], K.base_p[arm][2]};
          mulvec_acc(K.base_R[arm], c, wv);
#pragma unroll
          for (int k = 0; k < 3; k++) cen[(s * 3 + k) * kTile + lane] = wv[k];
        }
      }
      // world-frame spheres, a quarter per wavefront
      for (int s = S->slot_begin[kSceneSlots - 1]; s < S->slot_begin[kSceneSlots]; s++) {
        if ((s & 3) != w) continue;
#pragma unroll
        for (int k = 0; k < 3; k++) cen[(s * 3 + k) * kTile + lane] = S->c[s][k];
      }
    }
    __syncthreads();
    // ---- phase 2: this wavefront's quarter of the pair list ------------------------------------------------------------
    double best = __builtin_inf();
    int bestp = 0x7fffffff;
    for (int p = w; p < np; p += 4) {
      const unsigned ij = S->pair_ij[p];
      const int i = (int)(ij & 0xffu), j = (int)(ij >> 8);
      const double a0 = cen[(i * 3 + 0) * kTile + lane], a1 = cen[(i * 3 + 1) * kTile + lane], a2 = cen[(i * 3 + 2) * kTile + lane];
      double clr;
      if (j < CCMP_MAX_SPHERES) {
        const double d0 = a0 - cen[(j * 3 + 0) * kTile + lane], d1 = a1 - cen[(j * 3 + 1) * kTile + lane],
                     d2 = a2 - cen[(j * 3 + 2) * kTile + lane];
        clr = ccmp_sqrt(dot3(d0, d0, d1, d1, d2, d2)) - (S->r[i] + S->r[j]);
      } else {
        const int b = j - CCMP_MAX_SPHERES;
        const double d[3] = {a0 - S->box_c[b][0], a1 - S->box_c[b][1], a2 - S->box_c[b][2]};
        double l[3], e[3];
        mulTvec(S->box_R[b], d, l); // into the box's axes
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double a = ccmp_abs(l[k]) - S->box_half[b][k];
          e[k] = a > 0.0 ? a : 0.0;
        }
        clr = ccmp_sqrt(dot3(e[0], e[0], e[1], e[1], e[2], e[2])) - S->r[i];
      }
      if (clr < best) { best = clr; bestp = p; }
    }
    red[w * kTile + lane] = best;
    redp[w * kTile + lane] = bestp;
    __syncthreads();
    // ---- phase 3 -------------------------------------------------------------------------------------------------
    if (w == 0) {
      int any_bad = bad[lane] | bad[kTile + lane]; // wavefronts 0 and 1 saw the two arms (2 and 3 saw them again)
#pragma unroll
      for (int k = 1; k < 4; k++) {
        const double v = red[k * kTile + lane];
        const int vp = redp[k * kTile + lane];
        if (v < best || (v == best && vp < bestp)) { best = v; bestp = vp; }
      }
      if (live) {
        const double out = any_bad ? __builtin_nan("") : best;
        clearance[idx] = out;
        if (pair) pair[idx] = (any_bad || bestp == 0x7fffffff) ? -1 : S->pair_code[bestp];
        if (free_out) free_out[idx] = (uint8_t)((ok_in == nullptr || ok_in[idx] != 0) && out > margin);
      }
    }
    __syncthreads(); // the next tile overwrites the centres
  }
}

} // namespace

extern "C" {

size_t ccmp_clearance_lds_bytes(int n_spheres)
{
  return (size_t)n_spheres * 3 * kTile * sizeof(double) + 4 * kTile * sizeof(double) + 2 * 4 * kTile * sizeof(int);
}

hipError_t ccmp_launch_clearance(const ccmp_consts *K, const scene_dev *scene_dev_ptr, int n_spheres, const double *q, const uint8_t *ok_in,
                                 size_t B, double margin, double *clearance, int32_t *pair, uint8_t *free_out, int nblocks, hipStream_t st)
{
  const size_t lds = ccmp_clearance_lds_bytes(n_spheres);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(clearance_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(clearance_kernel, dim3(nblocks), dim3(256), lds, st, *K, scene_dev_ptr, q, ok_in, (unsigned long long)B, margin,
                     clearance, pair, free_out);
  return hipGetLastError();
}

} // extern "C"
