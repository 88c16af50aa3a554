// ccmp_scene.cpp — proxy scenes and the clearance entry points of the C ABI (include/ccmp.h): validation, the pair
// list, the device copy, launches.  The arithmetic lives in ccmp_kernels_scene.hip.
#include <cmath>
#include <cstring>
#include <new>

#include "ccmp_ctx.h"
#include "ccmp_host.h"
#include "ccmp_resident.h"
#include "ccmp_scene.h"

using namespace ccmp_host;
using ccmp::kSceneSlots;
using ccmp::scene_dev;

extern "C" {
hipError_t ccmp_launch_clearance(const ccmp_consts *K, const scene_dev *scene_dev_ptr, int n_spheres, int n_pairs, const double *q, const uint8_t *ok_in,
                                 size_t B, double margin, double *clearance, int32_t *pair, uint8_t *free_out, int nblocks, int per_state,
                                 unsigned int *done_flag, unsigned int done_seq, hipStream_t st);
size_t ccmp_clearance_lds_bytes(int n_spheres);
}

namespace {

bool finite3(const double *v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }
int slot_of(int frame) { return frame < 0 ? kSceneSlots - 1 : frame; }
bool static_frame(int frame) { return frame < 0 || frame % 9 == 8; }
bool pair_allowed(const uint32_t *allowed, int g, int h)
{
  return allowed && ((((allowed[g] >> h) & 1u) != 0) || (((allowed[h] >> g) & 1u) != 0));
}

}  // namespace

extern "C" {

int ccmp_scene_create(ccmp_ctx *ctx, const ccmp_sphere *spheres, int n_spheres, const ccmp_box *boxes, int n_boxes,
                      const uint32_t allowed[32], ccmp_scene **out)
{
  if (!out) return CCMP_EINVAL;
  *out = nullptr;
  if (!ctx || n_spheres < 0 || n_spheres > CCMP_MAX_SPHERES || n_boxes < 0 || n_boxes > CCMP_MAX_BOXES) return CCMP_EINVAL;
  if ((n_spheres > 0 && !spheres) || (n_boxes > 0 && !boxes)) return CCMP_EINVAL;
  for (int i = 0; i < n_spheres; i++) {
    const ccmp_sphere &s = spheres[i];
    if (s.frame < CCMP_FRAME_WORLD || s.frame >= 18 || s.group < 0 || s.group > 31) return CCMP_EINVAL;
    if (!finite3(s.c) || !std::isfinite(s.r) || s.r < 0.0) return CCMP_EINVAL;
  }
  for (int b = 0; b < n_boxes; b++) {
    const ccmp_box &x = boxes[b];
    if (x.group < 0 || x.group > 31 || !finite3(x.c) || !finite3(x.half)) return CCMP_EINVAL;
    if (x.half[0] < 0.0 || x.half[1] < 0.0 || x.half[2] < 0.0) return CCMP_EINVAL;
    for (int k = 0; k < 9; k++)
      if (!std::isfinite(x.R[k])) return CCMP_EINVAL;
  }
  ccmp_scene *sc = new (std::nothrow) ccmp_scene();
  if (!sc) return CCMP_ENOMEM;
  sc->device = ctx->device;
  scene_dev &H = sc->host;
  memset(&H, 0, sizeof H);
  H.n_spheres = n_spheres;
  H.n_boxes = n_boxes;
  // stable counting sort of the spheres by frame slot
  int count[kSceneSlots + 1] = {0};
  for (int i = 0; i < n_spheres; i++) count[slot_of(spheres[i].frame) + 1]++;
  for (int s = 0; s < kSceneSlots; s++) count[s + 1] += count[s];
  for (int s = 0; s <= kSceneSlots; s++) H.slot_begin[s] = count[s];
  int stored_of[CCMP_MAX_SPHERES];
  {
    int next[kSceneSlots];
    for (int s = 0; s < kSceneSlots; s++) next[s] = count[s];
    for (int i = 0; i < n_spheres; i++) {
      const int k = next[slot_of(spheres[i].frame)]++;
      stored_of[i] = k;
      H.user[k] = i;
      H.slot[k] = slot_of(spheres[i].frame);
      memcpy(H.c[k], spheres[i].c, sizeof H.c[k]);
      H.r[k] = spheres[i].r;
    }
  }
  for (int b = 0; b < n_boxes; b++) {
    memcpy(H.box_c[b], boxes[b].c, sizeof H.box_c[b]);
    memcpy(H.box_R[b], boxes[b].R, sizeof H.box_R[b]);
    memcpy(H.box_half[b], boxes[b].half, sizeof H.box_half[b]);
  }
  // the tested pairs, in the public numbering
  int np = 0;
  for (int i = 0; i < n_spheres; i++)
    for (int j = i + 1; j < n_spheres; j++) {
      if (spheres[i].frame == spheres[j].frame) continue;
      if (static_frame(spheres[i].frame) && static_frame(spheres[j].frame)) continue;
      if (pair_allowed(allowed, spheres[i].group, spheres[j].group)) continue;
      H.pair_ij[np] = (uint32_t)stored_of[i] | ((uint32_t)stored_of[j] << 8);
      H.pair_code[np] = i | (j << 8);
      H.pair_rsum[np] = spheres[i].r + spheres[j].r;
      np++;
    }
  H.n_pairs_ss = np;
  for (int i = 0; i < n_spheres; i++)
    for (int b = 0; b < n_boxes; b++) {
      if (static_frame(spheres[i].frame)) continue;
      if (pair_allowed(allowed, spheres[i].group, boxes[b].group)) continue;
      H.pair_ij[np] = (uint32_t)stored_of[i] | ((uint32_t)(CCMP_MAX_SPHERES + b) << 8);
      H.pair_code[np] = i | ((CCMP_MAX_SPHERES + b) << 8);
      H.pair_rsum[np] = spheres[i].r;
      np++;
    }
  H.n_pairs = np;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) { delete sc; return CCMP_ENODEV; }
  ccmp_host::quiesce(ctx);
  hipError_t e = hipMalloc((void **)&sc->dev, sizeof(scene_dev));
  if (e == hipSuccess) e = hipMemcpy(sc->dev, &H, sizeof(scene_dev), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (sc->dev) (void)hipFree(sc->dev);
    delete sc;
    return hip_fail(e, "ccmp_scene_create");
  }
  sc->ctx = ctx;
  *out = sc;
  return CCMP_OK;
}

void ccmp_scene_destroy(ccmp_scene *scene)
{
  if (!scene) return;
  {
    DeviceGuard guard(scene->device);
    if (scene->ctx && ccmp_host::context_alive(scene->ctx)) ccmp_host::quiesce(scene->ctx);
    if (scene->dev) (void)hipFree(scene->dev);
  }
  delete scene;
}

int ccmp_scene_num_pairs(const ccmp_scene *scene) { return scene ? scene->host.n_pairs : 0; }

int ccmp_clearance_batch(ccmp_ctx *ctx, const ccmp_problem *p, const ccmp_scene *scene, const double *q, const uint8_t *ok_in,
                         size_t B, double margin, double *clearance, int32_t *pair, uint8_t *free_out, void *hip_stream)
{
  if (!ctx || !p || !scene) return CCMP_EINVAL;
  if (scene->device != ctx->device) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q || !clearance || std::isnan(margin)) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  ccmp_consts K;
  make_consts(*p, K);
  if (B <= ctx->clearance_per_state_max) {
    // small batches: one block per state (the parallelism of a single state; what a validity-checker wrapper calls)
    size_t blocks = B;
    const size_t cap = (size_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    unsigned int *flag = nullptr;
    if (ctx->want_done && B == 1 && ctx->pin_dev) {
      ctx->done_seq++;
      ctx->done_armed = true;
      flag = (unsigned int *)((char *)ctx->pin_dev + ccmp_host::kPinData);
    }
    HIP_TRY(ccmp_launch_clearance(&K, scene->dev, scene->host.n_spheres, scene->host.n_pairs, q, ok_in, B, margin, clearance, pair, free_out, (int)blocks, 1,
                                  flag, ctx->done_seq, (hipStream_t)hip_stream));
    return CCMP_OK;
  }
  // one 256-thread block per tile of 64 states; the centres of a tile take 1.5 KB of LDS per sphere, so a CU holds
  // 160 KB / that many blocks (and at most eight: 2048 threads)
  const size_t lds = ccmp_clearance_lds_bytes(scene->host.n_spheres);
  size_t per_cu = (160 * 1024) / lds;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  size_t blocks = (B + 63) / 64;
  const size_t cap = (size_t)ctx->num_cus * per_cu;
  if (blocks > cap) blocks = cap;
  HIP_TRY(ccmp_launch_clearance(&K, scene->dev, scene->host.n_spheres, scene->host.n_pairs, q, ok_in, B, margin, clearance, pair, free_out, (int)blocks, 0,
                                nullptr, 0, (hipStream_t)hip_stream));
  return CCMP_OK;
}

}  // extern "C"
