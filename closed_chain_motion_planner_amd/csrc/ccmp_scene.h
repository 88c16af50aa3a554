/* ccmp_scene.h — device-side description of a proxy scene (ccmp_scene.cpp builds it, ccmp_kernels_scene.hip reads it);
 * not part of the public ABI. */
#ifndef CCMP_SCENE_H
#define CCMP_SCENE_H
#include <stdint.h>

#include "../../include/ccmp.h"

namespace ccmp {

constexpr int kSceneSlots = 19;                                                   // 2 arms x (7 bodies, hand, base) + world
constexpr int kSceneMaxPairs = CCMP_MAX_SPHERES * (CCMP_MAX_SPHERES - 1) / 2 + CCMP_MAX_SPHERES * CCMP_MAX_BOXES;

/* Spheres are stored sorted by frame slot (slot = frame code, world last), so that a wavefront walking one arm's chain
 * places the spheres of a frame the moment that frame exists; `user` maps a stored index back to the caller's. */
struct scene_dev {
  int32_t n_spheres, n_boxes, n_pairs, n_pairs_ss; /* the first n_pairs_ss pairs are sphere-sphere, the rest sphere-box */
  int32_t slot_begin[kSceneSlots + 1];
  int32_t user[CCMP_MAX_SPHERES];
  int32_t slot[CCMP_MAX_SPHERES]; /* frame slot of each stored sphere (the per-state kernel walks to it) */
  double c[CCMP_MAX_SPHERES][3];
  double r[CCMP_MAX_SPHERES];
  double box_c[CCMP_MAX_BOXES][3];
  double box_R[CCMP_MAX_BOXES][9];
  double box_half[CCMP_MAX_BOXES][3];
  /* tested pairs in the public numbering (include/ccmp.h): stored index of the first sphere | (stored index of the
   * second, or 64 + box) << 8 — one dword, so that the kernel fetches it with a scalar load; code = the value reported
   * to the caller (the caller's own indices) */
  uint32_t pair_ij[kSceneMaxPairs];
  int32_t pair_code[kSceneMaxPairs];
  double pair_rsum[kSceneMaxPairs]; /* r_i + r_j (sphere-sphere) or r_i (sphere-box): what is subtracted from the distance */
};

}  // namespace ccmp

struct ccmp_scene {
  int device = 0;
  ccmp_ctx *ctx = nullptr; // the context the scene was created on: its resident service kernel is stopped before the scene's hipFree (ccmp_resident.h rule 2); the context must outlive the scene or be destroyed first with the service off
  ccmp::scene_dev host;
  ccmp::scene_dev *dev = nullptr;
};

#endif /* CCMP_SCENE_H */
