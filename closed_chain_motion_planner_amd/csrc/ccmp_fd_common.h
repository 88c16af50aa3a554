/* ccmp_fd_common.h — pieces shared by the reference-arithmetic translation units
 * (ccmp_kernels_fd.hip: 10-samples-per-wave throughput kernel; ccmp_kernels_wave.hip: one-wave-per-sample
 * latency kernels).  Both units are built -ffp-contract=off -DCCMP_USE_FMA. */
#ifndef CCMP_FD_COMMON_H
#define CCMP_FD_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccmp_kin.h"
#include "ccmp_solve.h"

/* Bulk extend calls with the LIVE hand-over (round 5; ccmp_api.cpp: geodesic_common): geodesic_group_kernel's wavefronts publish the
 * edges they give up WHILE the kernel runs, and the front's geodesic_flat_kernel blocks — once the front of the order is taken —
 * pick them up from the pool instead of leaving.  words = the 64-bit words of the bulk form (sixteen are reserved):
 *   [0] group kernel's ticket   [1] group wavefronts that have left   [2] blocks waiting for entries   [3] edges finished or given up
 *   [4] front length            [5] front's ticket                    [6] pool entries reserved        [7] pool entries claimed
 *   [8] entries published and not yet taken (a counting semaphore)
 * A producer reserves its entries ([6] += n), writes them, fences, flags them and then adds n to [8]; a taker subtracts one from [8]
 * (and puts it back if there was none), takes the next ticket of [7] — which is then below [6] — waits for THAT entry's flag (its
 * producer is a running wavefront between its reservation and its flag: a few hundred cycles) and clears it again — every call
 * leaves the flags as it found them, all zero, which is what makes a captured call replayable. */
struct ccmp_geo_live {
  unsigned long long *words; /* NULL: not a live call */
  unsigned int *flags;       /* one per pool entry */
  int group_waves;           /* producers: the grid of geodesic_group_kernel; 0 = none are running (the launch behind it): drain and leave */
  int max_pollers;           /* blocks that may wait for entries at one time (the others leave: the launch behind the group kernel drains what is left) */
  int poll_limit;            /* polls a block spends waiting once the group kernel is seen running (a backstop, not a schedule) */
};

namespace ccmp {

constexpr int kGeoPoolEntry = 40; // (= kGeoPoolDoubles, ccmp_ctx.h) extend step, hand-over of an edge in the middle of a projection: x[14], previous[14], dist, total, maxd,
                                  // edge, (n, its), (rounds, iter), updates, norm1, norm2 (geodesic_group_kernel -> geodesic_flat_kernel)
constexpr int kPoolEntry = 18; // straggler hand-over record: x[14], idx, (iter,updates), norm1, norm2

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


} /* namespace ccmp */
#endif
