// ccmp_kernels_scene.hip — proxy-geometry clearance (include/ccmp.h: ccmp_clearance_batch): the pre-filter a planner puts
// in front of KinematicChainValidityChecker::isValid (src/kinematics/KinematicChain.cpp:94-123, MoveIt on the host).
// Canonical rounding model (-ffp-contract=off -DCCMP_USE_FMA): bit-identical to oracle/ccmp_oracle.c:orc_clearance.
//
// Large batches: one 256-thread block per tile of 64 states; lane l of every wavefront works on state l of the tile.
//   phase 1  wavefront w walks the chain of arm (w & 1) — the projector's own FK (ccmp_kin.h: joint_step, the general
//            form; the STOCK specialisations produce the same bits) — and, the moment a body frame exists, places the
//            spheres attached to it whose stored index has parity (w >> 1): world centre = t_wb (o + R c).  Spheres are
//            stored sorted by frame, so "the spheres of this frame" is a scalar index range and every load of proxy
//            data is a scalar load.  Centres go to LDS as [sphere][component][lane]: conflict-free.
//   phase 2  wavefront w takes the tested pairs p = w, w + 4, … of the scene's list — its share of the table sits in
//            registers, spread over the lanes, and is fetched with v_readlane — four pairs at a time, keeps the smallest
//            signed distance and the first pair attaining it.
//   phase 3  the four partial minima meet in LDS; wavefront 0 writes clearance / pair / flag.
// HBM traffic: 112 B read + 13 B written per state; the work is the pair loop (one IEEE square root per pair).
// Small batches: one block per state (clearance_state_kernel below).
#include "ccmp_fd_common.h"
#include "ccmp_scene.h"

using namespace ccmp;

namespace {

constexpr int kTile = 64;
constexpr int kTileWaves = 4, kTileThreads = 64 * kTileWaves; // wavefronts that share one tile's centres (8: each chain walked by four
                                                               // wavefronts, twice the occupancy — measured 13 % slower at 262144 states)
constexpr int kSmallShare = 1024 / (64 * kTileWaves); // register sets that hold a wavefront's share of up to 1024 tested pairs
constexpr int kMaxShare = (kSceneMaxPairs / kTileWaves + 63) / 64; // register sets for the largest pair table

// spheres [b, e) whose stored index falls to this wavefront (index mod kTileWaves / 2 == part), attached to the frame (R, o) of arm `arm` (in the arm's base)
__device__ __forceinline__ void place_spheres(const ccmp_consts &K, const scene_dev *__restrict__ S, int arm, int b, int e, int part,
                                              const double *R, const double *o, double *cen, int lane)
{
  for (int s = b; s < e; s++) {
    if ((s & (kTileWaves / 2 - 1)) != part) continue;
    const double c[3] = {S->c[s][0], S->c[s][1], S->c[s][2]};
    double v[3] = {o[0], o[1], o[2]};
    mulvec_acc(R, c, v);
    double wv[3] = {K.base_p[arm][0], K.base_p[arm][1], K.base_p[arm][2]};
    mulvec_acc(K.base_R[arm], v, wv);
#pragma unroll
    for (int k = 0; k < 3; k++) cen[(s * 3 + k) * kTile + lane] = wv[k];
  }
}

// squared distance between the centres of one sphere-sphere pair of the tile (stored indices in ij)
__device__ __forceinline__ double sphere_pair_d2(const double *cen, int lane, unsigned ij)
{
  const int i = (int)(ij & 0xffu), j = (int)(ij >> 8);
  const double d0 = cen[(i * 3 + 0) * kTile + lane] - cen[(j * 3 + 0) * kTile + lane];
  const double d1 = cen[(i * 3 + 1) * kTile + lane] - cen[(j * 3 + 1) * kTile + lane];
  const double d2 = cen[(i * 3 + 2) * kTile + lane] - cen[(j * 3 + 2) * kTile + lane];
  return dot3(d0, d0, d1, d1, d2, d2);
}
// One pair against the running minimum.  The signed distance sqrt(d2) - rsum can only become the minimum (ties keep
// the earlier pair) if sqrt(d2) < best + rsum; a pair that is further than that by a nanometre is dropped on its
// squared distance — the margin is 1e6 times the rounding of the four operations involved — and the square root, ten
// dependent operations, runs only when some lane of the wavefront still needs it: after the first few pairs of a state
// almost never.  Pairs that do run are computed exactly as before, so the result is unchanged bit for bit.
__device__ __forceinline__ void consider_pair(double d2, double rsum, int p, double &best, int &bestp)
{
  const double t = (best + rsum) + 1e-9;
  const bool need = !(t <= 0.0 || d2 > t * t); // also true for a NaN (never the case for finite states)
  if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
    const double clr = ccmp_sqrt(d2) - rsum;
    if (clr < best) { best = clr; bestp = p; }
  }
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// NK = register sets of 64 entries that hold a wavefront's quarter of the pair table (NK = 4: up to 1024 tested pairs)
template <int NK>
__global__ __launch_bounds__(kTileThreads) void clearance_kernel(const ccmp_consts K, const scene_dev *__restrict__ S, const double *__restrict__ q,
                                                        const uint8_t *__restrict__ ok_in, unsigned long long B, double margin,
                                                        double *__restrict__ clearance, int32_t *__restrict__ pair,
                                                        uint8_t *__restrict__ free_out)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int ns = S->n_spheres, np = S->n_pairs;
  double *cen = lds;                               // [ns][3][64]
  double *red = lds + (size_t)ns * 3 * kTile;      // [kTileWaves][64] partial minima
  int *redp = reinterpret_cast<int *>(red + kTileWaves * kTile); // [kTileWaves][64] their positions in the pair list
  int *bad = redp + kTileWaves * kTile;            // [kTileWaves][64] non-finite joint value seen
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int arm = w & 1, part = w >> 1; // this wavefront's arm and its share of that arm's spheres
  const unsigned long long tiles = (B + kTile - 1) / kTile;
  // This wavefront's quarter of the pair table (pairs w, w + 4, …), spread over its lanes once per launch: entry
  // n = 64 k + l sits in lane l of register set k and is fetched with v_readlane inside the pair loop.  (Scalar loads
  // there put two dependent scalar-memory round trips — pair, then radii — in front of every pair: 60 % of the run time.)
  unsigned my_ij[NK];
  double my_rs[NK];
#pragma unroll
  for (int k = 0; k < NK; k++) {
    const int p = w + kTileWaves * (64 * k + lane);
    my_ij[k] = p < np ? S->pair_ij[p] : 0u;
    my_rs[k] = p < np ? S->pair_rsum[p] : 0.0;
  }
  const int nss = S->n_pairs_ss;
  const int nss_w = nss > w ? (nss - w + kTileWaves - 1) / kTileWaves : 0; // entries of this wavefront's share that are sphere-sphere pairs
  const int np_w = np > w ? (np - w + kTileWaves - 1) / kTileWaves : 0;    // … and all of them

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
        place_spheres(K, S, arm, S->slot_begin[slot], S->slot_begin[slot + 1], part, R, o, cen, lane);
      }
      { // hand frame: PandaModel::getTranslation / getRotation (ccmp_kin.h: tool_pose_t, before t_wb)
        double oh[3] = {o[0], o[1], o[2]}, Rh[9];
        mulvec_acc(R, K.ee[arm], oh);
        mul33(R, K.R_tool[arm], Rh);
        const int slot = arm * 9 + 7;
        place_spheres(K, S, arm, S->slot_begin[slot], S->slot_begin[slot + 1], part, Rh, oh, cen, lane);
      }
      { // the arm's base: only t_wb applies
        const int slot = arm * 9 + 8;
        for (int s = S->slot_begin[slot]; s < S->slot_begin[slot + 1]; s++) {
          if ((s & (kTileWaves / 2 - 1)) != part) continue;
          const double c[3] = {S->c[s][0], S->c[s][1], S->c[s][2]};
          double wv[3] = {K.base_p[arm][0], K.base_p[arm][1], K.base_p[arm][2]};
          mulvec_acc(K.base_R[arm], c, wv);
#pragma unroll
          for (int k = 0; k < 3; k++) cen[(s * 3 + k) * kTile + lane] = wv[k];
        }
      }
      // world-frame spheres, shared out over the wavefronts
      for (int s = S->slot_begin[kSceneSlots - 1]; s < S->slot_begin[kSceneSlots]; s++) {
        if ((s & (kTileWaves - 1)) != w) continue;
#pragma unroll
        for (int k = 0; k < 3; k++) cen[(s * 3 + k) * kTile + lane] = S->c[s][k];
      }
    }
    __syncthreads();
    // ---- phase 2: this wavefront's quarter of the pair list ------------------------------------------------------------
    // Sphere-sphere pairs four at a time (their LDS reads and squared distances in flight together: with the two or three
    // wavefronts per SIMD that the LDS footprint allows nothing else would fill the pipeline); the minimum is taken in
    // list order, so the result does not depend on the grouping.
    double best = __builtin_inf();
    int bestp = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NK; k++) {
      const int base = 64 * k;
      const int ss_end = nss_w - base < 0 ? 0 : (nss_w - base > 64 ? 64 : nss_w - base);
      const int end = np_w - base > 64 ? 64 : np_w - base; // <= 0 past the table: nothing runs
      int l = 0;
      for (; l + 3 < ss_end; l += 4) {
        double d2[4], rs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          d2[u] = sphere_pair_d2(cen, lane, (unsigned)__builtin_amdgcn_readlane((int)my_ij[k], l + u));
          rs[u] = readlane_f64(my_rs[k], l + u);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) consider_pair(d2[u], rs[u], w + kTileWaves * (base + l + u), best, bestp);
      }
      for (; l < ss_end; l++)
        consider_pair(sphere_pair_d2(cen, lane, (unsigned)__builtin_amdgcn_readlane((int)my_ij[k], l)), readlane_f64(my_rs[k], l),
                      w + kTileWaves * (base + l), best, bestp);
      for (; l < end; l++) { // sphere-box pairs
        const unsigned ij = (unsigned)__builtin_amdgcn_readlane((int)my_ij[k], l);
        const int i = (int)(ij & 0xffu), b = (int)(ij >> 8) - CCMP_MAX_SPHERES;
        const double d[3] = {cen[(i * 3 + 0) * kTile + lane] - S->box_c[b][0], cen[(i * 3 + 1) * kTile + lane] - S->box_c[b][1],
                             cen[(i * 3 + 2) * kTile + lane] - S->box_c[b][2]};
        double lb[3], e[3];
        mulTvec(S->box_R[b], d, lb); // into the box's axes
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const double a = ccmp_abs(lb[c]) - S->box_half[b][c];
          e[c] = a > 0.0 ? a : 0.0;
        }
        consider_pair(dot3(e[0], e[0], e[1], e[1], e[2], e[2]), readlane_f64(my_rs[k], l), w + kTileWaves * (base + l), best, bestp);
      }
    }
    red[w * kTile + lane] = best;
    redp[w * kTile + lane] = bestp;
    __syncthreads();
    // ---- phase 3 -------------------------------------------------------------------------------------------------
    if (w == 0) {
      int any_bad = bad[lane] | bad[kTile + lane]; // wavefronts 0 and 1 saw the two arms (the others saw them again)
#pragma unroll
      for (int k = 1; k < kTileWaves; k++) {
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

// ---- small batches: one 256-thread block per STATE ---------------------------------------------------------------------
// The tile kernel above gives a lone state one lane of each wavefront (≈ 20 us: 223 pairs in sequence behind the chain).
// Here the parallelism of one state is used instead: threads 0..13 compute the fourteen sines / cosines, thread s walks
// the chain of sphere s's arm up to its frame and places it (≤ 7 joint steps), then thread t takes the pairs t, t + 256, …
// and the minimum — ties to the earlier pair — is reduced across the block.  The same operations on the same operands
// as the tile kernel, hence the same bits.  What a StateValidityChecker wrapper calls with one state, and the quicker
// kernel up to a few thousand states (more blocks than the tile kernel has tiles).
constexpr int kConstsDoubles = (int)((sizeof(ccmp_consts) + 7) / 8);

__global__ __launch_bounds__(256) void clearance_state_kernel(const ccmp_consts K, const scene_dev *__restrict__ S, const double *__restrict__ q,
                                                              const uint8_t *__restrict__ ok_in, unsigned long long B, double margin,
                                                              double *__restrict__ clearance, int32_t *__restrict__ pair,
                                                              uint8_t *__restrict__ free_out, unsigned int *done_flag, unsigned int done_seq)
{
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ double sc[28];                      // (sin, cos) of the 14 joints
  __shared__ double cen[CCMP_MAX_SPHERES * 3];
  __shared__ double red[4];
  __shared__ int redp[4];
  __shared__ int bad_state;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  {
    const double *src = reinterpret_cast<const double *>(&K);
    for (int k = tid; k < kConstsDoubles; k += 256) ktab[k] = src[k];
  }
  const ccmp_consts &KL = *reinterpret_cast<const ccmp_consts *>(ktab);
  const int ns = S->n_spheres, np = S->n_pairs;
  for (unsigned long long st = blockIdx.x; st < B; st += gridDim.x) {
    bool nonfinite = false;
    if (tid < 14) {
      const double x = q[st * 14 + tid];
      nonfinite = !(x - x == 0.0);
      double s, c;
      ccmp_sincos(x, &s, &c);
      sc[2 * tid] = s;
      sc[2 * tid + 1] = c;
    }
    if (w == 0) { // threads 0..13 sit in wavefront 0
      const bool any = __builtin_amdgcn_ballot_w64(nonfinite) != 0ull;
      if (tid == 0) bad_state = any ? 1 : 0;
    }
    __syncthreads();
    if (tid < ns) {
      const int slot = S->slot[tid];
      double wv[3];
      if (slot == kSceneSlots - 1) {
        wv[0] = S->c[tid][0]; wv[1] = S->c[tid][1]; wv[2] = S->c[tid][2];
      } else {
        const int arm = slot >= 9 ? 1 : 0, k = slot - 9 * arm;
        const double c[3] = {S->c[tid][0], S->c[tid][1], S->c[tid][2]};
        double v[3];
        if (k == 8) {
          v[0] = c[0]; v[1] = c[1]; v[2] = c[2];
        } else {
          double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
          const int last = k < 7 ? k : 6;
          for (int i = 0; i <= last; i++) joint_step(KL, arm, i, sc[2 * (arm * 7 + i)], sc[2 * (arm * 7 + i) + 1], R, o);
          if (k == 7) {
            double Rh[9];
            mulvec_acc(R, KL.ee[arm], o);
            mul33(R, KL.R_tool[arm], Rh);
#pragma unroll
            for (int e = 0; e < 9; e++) R[e] = Rh[e];
          }
          v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
          mulvec_acc(R, c, v);
        }
        wv[0] = KL.base_p[arm][0]; wv[1] = KL.base_p[arm][1]; wv[2] = KL.base_p[arm][2];
        mulvec_acc(KL.base_R[arm], v, wv);
      }
      cen[3 * tid] = wv[0]; cen[3 * tid + 1] = wv[1]; cen[3 * tid + 2] = wv[2];
    }
    __syncthreads();
    double best = __builtin_inf();
    int bestp = 0x7fffffff;
    for (int p = tid; p < np; p += 256) {
      const unsigned ij = S->pair_ij[p];
      const int i = (int)(ij & 0xffu), j = (int)(ij >> 8);
      const double a0 = cen[3 * i], a1 = cen[3 * i + 1], a2 = cen[3 * i + 2];
      double clr;
      if (j < CCMP_MAX_SPHERES) {
        const double d0 = a0 - cen[3 * j], d1 = a1 - cen[3 * j + 1], d2 = a2 - cen[3 * j + 2];
        clr = ccmp_sqrt(dot3(d0, d0, d1, d1, d2, d2)) - (S->r[i] + S->r[j]);
      } else {
        const int b = j - CCMP_MAX_SPHERES;
        const double d[3] = {a0 - S->box_c[b][0], a1 - S->box_c[b][1], a2 - S->box_c[b][2]};
        double l[3], e[3];
        mulTvec(S->box_R[b], d, l);
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double a = ccmp_abs(l[k]) - S->box_half[b][k];
          e[k] = a > 0.0 ? a : 0.0;
        }
        clr = ccmp_sqrt(dot3(e[0], e[0], e[1], e[1], e[2], e[2])) - S->r[i];
      }
      if (clr < best) { best = clr; bestp = p; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { // minimum of the wavefront, ties to the earlier pair
      const double v = shfl_f64(best, lane ^ m);
      const int vp = __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, bestp);
      if (v < best || (v == best && vp < bestp)) { best = v; bestp = vp; }
    }
    if (lane == 0) { red[w] = best; redp[w] = bestp; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
      for (int k = 1; k < 4; k++)
        if (red[k] < best || (red[k] == best && redp[k] < bestp)) { best = red[k]; bestp = redp[k]; }
      const double out = bad_state ? __builtin_nan("") : best;
      clearance[st] = out;
      if (pair) pair[st] = (bad_state || bestp == 0x7fffffff) ? -1 : S->pair_code[bestp];
      if (free_out) free_out[st] = (uint8_t)((ok_in == nullptr || ok_in[st] != 0) && out > margin);
    }
    __syncthreads();
  }
  // single-state calls through the host entry point: results sit in pinned host memory, the host polls this word
  if (done_flag != nullptr && tid == 0) {
    __threadfence_system();
    __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

} // namespace

extern "C" {

size_t ccmp_clearance_lds_bytes(int n_spheres)
{
  return (size_t)n_spheres * 3 * kTile * sizeof(double) + kTileWaves * kTile * sizeof(double) + 2 * kTileWaves * kTile * sizeof(int);
}

// per_state != 0: one block per state (small batches, nblocks <= B); done_flag (nullable) is honoured for B == 1 only
hipError_t ccmp_launch_clearance(const ccmp_consts *K, const scene_dev *scene_dev_ptr, int n_spheres, int n_pairs, const double *q, const uint8_t *ok_in,
                                 size_t B, double margin, double *clearance, int32_t *pair, uint8_t *free_out, int nblocks, int per_state,
                                 unsigned int *done_flag, unsigned int done_seq, hipStream_t st)
{
  if (per_state) {
    if (B != 1) done_flag = nullptr;
    hipLaunchKernelGGL(clearance_state_kernel, dim3(nblocks), dim3(256), 0, st, *K, scene_dev_ptr, q, ok_in, (unsigned long long)B, margin,
                       clearance, pair, free_out, done_flag, done_seq);
    return hipGetLastError();
  }
  const size_t lds = ccmp_clearance_lds_bytes(n_spheres);
  if (lds > 64 * 1024) { // per device and function; cheap, but not free: asked once per size class and thread
    static thread_local size_t granted = 0;
    static thread_local int granted_dev = -1;
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (dev != granted_dev || lds > granted) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(clearance_kernel<kSmallShare>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(clearance_kernel<kMaxShare>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      granted = lds;
      granted_dev = dev;
    }
  }
  if (n_pairs <= kSmallShare * 64 * kTileWaves)
    hipLaunchKernelGGL(clearance_kernel<kSmallShare>, dim3(nblocks), dim3(kTileThreads), lds, st, *K, scene_dev_ptr, q, ok_in, (unsigned long long)B, margin,
                       clearance, pair, free_out);
  else
    hipLaunchKernelGGL(clearance_kernel<kMaxShare>, dim3(nblocks), dim3(kTileThreads), lds, st, *K, scene_dev_ptr, q, ok_in, (unsigned long long)B,
                       margin, clearance, pair, free_out);
  return hipGetLastError();
}

} // extern "C"
