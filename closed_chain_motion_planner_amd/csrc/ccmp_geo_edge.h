/* ccmp_geo_edge.h — what the extend step's per-edge body (ccmp_geo_edge_body.inc) needs around it: the LDS record's extra fields and
 * the distance.  Included behind ccmp_flat_newton.h by ccmp_kernels_geo.hip and ccmp_kernels_resident.hip; everything sits in the
 * including unit's anonymous namespace. */
#ifndef CCMP_GEO_EDGE_H
#define CCMP_GEO_EDGE_H
namespace {
// previous accepted state and target behind the Newton routine's record
constexpr int gPrev = fRec, gTo = fRec + 14, gRec = fRec + 28;


// RealVectorStateSpace::distance over the 14 joints (plain Euclidean, KinematicChainSpace does not override it),
// summed serially in the canonical order; every thread computes it from LDS.
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

} // namespace
#endif /* CCMP_GEO_EDGE_H */
