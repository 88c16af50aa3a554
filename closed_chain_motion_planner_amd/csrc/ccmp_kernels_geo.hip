// ccmp_kernels_geo.hip — the extend step (jy_ProjectedStateSpace::discreteGeodesic) on the latency kernels' Newton
// routine (ccmp_flat_newton.h).  One source, two objects (build.py):
//  * throughput flavour (ccmp_launch_geodesic) — the projector's latency kernel's flags: 128 registers, eight blocks per
//    CU.  For calls that bound the Newton rounds per edge (ccmp_geodesic_batch_ex, round_budget > 0): such a launch is
//    bound by how many rounds the chip turns over, and occupancy is what buys that.
//  * latency flavour (-DCCMP_GEO_LATENCY: ccmp_launch_geodesic_lat) — machine LICM on and a 256-register budget, four
//    blocks per CU: the ~60 FP64 literals of a Newton round live in registers instead of being re-materialised every round
//    (two moves each; 8 % fewer instructions per round).  For calls that end on ONE edge's serial chain — no round
//    budget, or no more edges than blocks.  In-process A/B, 16 384 near-neighbour edges, lists of 16: without a budget
//    2.55 -> 2.13 ms on this flavour; with 128 rounds per edge 1.51 ms on the throughput flavour and 1.85 ms on this one.
// Same source, same arithmetic (-ffp-contract=off): the two produce the same bits.
#include "ccmp_flat_newton.h"
#include "ccmp_geo_edge.h"
#ifdef CCMP_GEO_LATENCY
#define geodesic_flat_kernel geodesic_flat_kernel_lat // its own name in kernel traces
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// geodesic_flat_kernel — jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:32-96), one
// 128-thread block per edge (from -> to): interpolate a step of delta towards `to` (KinematicChainSpace::interpolate,
// KinematicChain.h:145-171), project it (flat_newton), apply the reference's four break tests, record the state.
// The StateValidityChecker (MoveIt collision) stays on the host: the kernel runs as the reference does with
// interpolate == true and the host truncates the list at the first invalid state, which is what the reference's
// break would have produced.
//
// Round 3.  Blocks are persistent and take edges from an atomic ticket (queue != NULL), optionally through a processing
// order (long edges first, geodesic_order_kernel below): the launch ends on short edges instead of on whichever long
// one the dispatcher happened to start last.  An edge that fills its list stops (n_states = max_states + 1) and leaves
// what a continuation needs in carry_out — the running length `total` BEFORE the step that did not fit and the bound
// lambda * dist(from, to); a later call with carry_in resumes it from its last stored state (passed as `from`): `dist`
// is then the distance of that state to the target, recomputed from the same operands as the value the first call held,
// so first call + continuation produce the states, flags and counts of one uninterrupted traversal bit for bit.
template <bool STOCK>
// (the general instantiation — calibrated arms, tilted bases — gets the register budget of three wavefronts per SIMD: under the
// stock one's 128 registers it spilled 33 of them, 96 B per lane)
__global__ __launch_bounds__(128, (STOCK || CCMP_FLAT_MIN_WAVES < 4) ? CCMP_FLAT_MIN_WAVES : 3) void geodesic_flat_kernel(
    const ccmp_consts K, const double delta, const double lambda, const double *__restrict__ from,
    const double *__restrict__ to, unsigned long long E, int max_states, double *__restrict__ states,
    int *__restrict__ n_states, uint8_t *__restrict__ ok_out, int *__restrict__ newton_iters, int check_target,
    unsigned long long *queue, const unsigned int *__restrict__ order, const double *__restrict__ carry_in,
    double *__restrict__ carry_out, int round_budget, const unsigned long long *__restrict__ total_ptr,
    const double *__restrict__ pool, const unsigned long long *__restrict__ pool_count, int static_first)
{
  __shared__ __attribute__((aligned(16))) double lds[gRec];
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ __attribute__((aligned(16))) double steptab[kStepTab];
  __shared__ unsigned long long ticket;
  const int tid = threadIdx.x;
  {
    const double *src = reinterpret_cast<const double *>(&K);
    for (int k = tid; k < kConstsDoubles; k += 128) ktab[k] = src[k];
  }
  stage_step_table(K, steptab, tid);
  __syncthreads();
  const ccmp_consts &KL = *reinterpret_cast<const ccmp_consts *>(ktab);
  double *rec = lds;
  const double pi = 3.14159265358979323846;

  // static_first (a launch that has the chip to itself: not the front or the hand-over of a bulk call, whose blocks are not all
  // resident at once — ccmp_kernels_flat.hip says why): the first ticket of every block is its own index, the queue word hands out
  // those behind the grid's — a fetch-add on one word costs 12 ns chip-wide, 2 048 blocks stood in line for up to 25 us
  unsigned long long tk = blockIdx.x;
  bool first = static_first != 0;
  for (;;) {
    if (queue && !first) {
      if (tid == 0) ticket = (static_first ? (unsigned long long)gridDim.x : 0ull) + atomicAdd(queue, 1ull);
      __syncthreads();
      tk = ticket;
    }
    first = false;
    // total_ptr (bulk calls, ccmp_api.cpp: geodesic_common): this launch takes the first *total_ptr tickets of the order — the
    // edges the scout predicts longest — beside geodesic_group_kernel, which takes the rest
    if (tk >= (pool ? *pool_count : (total_ptr ? *total_ptr : E))) break;
    // pool (bulk calls): the edges geodesic_group_kernel handed over in the middle of a projection — iterate, previous state,
    // running distances and counters — go on here exactly where they stood
    // (the ticket is block-uniform: a scalar address keeps the entry out of the vector registers this kernel has none to spare of)
    const unsigned long long tks = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(tk >> 32)) << 32) |
                                   (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
    const double *ent = pool ? pool + tks * (unsigned long long)kGeoPoolEntry : nullptr;
    const unsigned long long t = ent ? (unsigned long long)__double_as_longlong(ent[31]) : (order ? (unsigned long long)order[tk] : tk);
#ifdef CCMP_GEO_TRACE
    if (tid == 0 && t < 65536) { g_geo_trace[3 * t] = wall_clock64(); g_geo_trace[3 * t + 2] = ((unsigned long long)blockIdx.x << 32) | tk; }
#endif
#include "ccmp_geo_edge_body.inc"
    if (!queue) tk += gridDim.x;
  }
}

#ifndef CCMP_GEO_LATENCY // one copy: the throughput flavour's object holds the ordering pass
// Processing order for a large batch of edges: those longer than `long_dist` first (they need the most states, and the
// edges that creep — joint values on either side of the +-pi wrap — are among them), the rest behind, so that the launch
// ends on short edges.  One thread per edge; the long edges fill the order from the front, the others from the back
// (two atomic counters); the order inside a class is whatever the atomics give and changes no result.  The distance is a
// scheduling hint only: plain arithmetic, no claim on its bits.
__global__ void geodesic_order_kernel(const double *__restrict__ from, const double *__restrict__ to, unsigned long long E,
                                      double long_dist2, unsigned int *__restrict__ counters, unsigned int *__restrict__ order)
{
  const unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  double d2 = 0.0;
#pragma unroll
  for (int k = 0; k < 14; k++) {
    const double d = to[e * 14 + k] - from[e * 14 + k];
    d2 += d * d;
  }
  // ballot-aggregated atomics: one per wave and class
  const bool is_long = !(d2 <= long_dist2); // NaN counts as long
  const unsigned long long mL = __builtin_amdgcn_ballot_w64(is_long), mS = __builtin_amdgcn_ballot_w64(!is_long);
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned int baseL = 0, baseS = 0;
  if (lane == 0) {
    if (mL) baseL = atomicAdd(&counters[0], (unsigned int)__builtin_popcountll(mL));
    if (mS) baseS = atomicAdd(&counters[1], (unsigned int)__builtin_popcountll(mS));
  }
  baseL = __builtin_amdgcn_readfirstlane(baseL);
  baseS = __builtin_amdgcn_readfirstlane(baseS);
  if (is_long) order[baseL + (unsigned int)__builtin_popcountll(mL & below)] = (unsigned int)e;
  else order[(unsigned int)(E - 1) - (baseS + (unsigned int)__builtin_popcountll(mS & below))] = (unsigned int)e;
}

#endif

} // namespace

extern "C" {

#ifdef CCMP_GEO_LATENCY
#define CCMP_LAUNCH_GEODESIC ccmp_launch_geodesic_lat
#else
#define CCMP_LAUNCH_GEODESIC ccmp_launch_geodesic
#endif
hipError_t CCMP_LAUNCH_GEODESIC(const ccmp_consts *K, double delta, double lambda, const double *from, const double *to,
                                size_t E, int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters,
                                int check_target, int nblocks, unsigned long long *queue, const unsigned int *order,
                                const double *carry_in, double *carry_out, int round_budget, const unsigned long long *total_ptr,
                                const double *pool, const unsigned long long *pool_count, hipStream_t st)
{
  const int static_first = (pool == nullptr && total_ptr == nullptr) ? 1 : 0; // alone on the chip: neither front nor hand-over of a bulk call
  if (K->stock)
    hipLaunchKernelGGL(geodesic_flat_kernel<true>, dim3(nblocks), dim3(128), 0, st, *K, delta, lambda, from, to, (unsigned long long)E,
                       max_states, states, n_states, ok, newton_iters, check_target, queue, order, carry_in, carry_out, round_budget, total_ptr, pool, pool_count, static_first);
  else
    hipLaunchKernelGGL(geodesic_flat_kernel<false>, dim3(nblocks), dim3(128), 0, st, *K, delta, lambda, from, to, (unsigned long long)E,
                       max_states, states, n_states, ok, newton_iters, check_target, queue, order, carry_in, carry_out, round_budget, total_ptr, pool, pool_count, static_first);
  return hipGetLastError();
}

#ifndef CCMP_GEO_LATENCY
// counters: two zeroed words; order: E words
hipError_t ccmp_launch_geodesic_order(const double *from, const double *to, size_t E, double long_dist, unsigned int *counters,
                                      unsigned int *order, hipStream_t st)
{
  hipLaunchKernelGGL(geodesic_order_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, from, to, (unsigned long long)E,
                     long_dist * long_dist, counters, order);
  return hipGetLastError();
}

#endif

#ifdef CCMP_GEO_TRACE
#ifdef CCMP_GEO_LATENCY
#define ccmp_debug_geo_trace ccmp_debug_geo_trace_lat
#endif
hipError_t ccmp_debug_geo_trace(unsigned long long *out, size_t n_edges)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_geo_trace), 3 * n_edges * sizeof(unsigned long long));
}
#endif

} // extern "C"
