/* ccmp_fd_common.h — pieces shared by the reference-arithmetic translation units
 * (ccmp_kernels_fd.hip: 10-samples-per-wave throughput kernel; ccmp_kernels_wave.hip: one-wave-per-sample
 * latency kernels).  Both units are built -ffp-contract=off -DCCMP_USE_FMA. */
#ifndef CCMP_FD_COMMON_H
#define CCMP_FD_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccmp_kin.h"
#include "ccmp_solve.h"

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
