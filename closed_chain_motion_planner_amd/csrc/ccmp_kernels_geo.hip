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

template <bool STOCK>
__global__ __launch_bounds__(128, CCMP_FLAT_MIN_WAVES) void geodesic_flat_kernel(
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
    double *out = states + t * (unsigned long long)max_states * 14ull;
    if (tid < 14) {
      if (ent) {
        rec[fX + tid] = ent[tid];
        rec[gPrev + tid] = ent[14 + tid];
      } else {
        const double a = from[t * 14 + tid];
        rec[gPrev + tid] = a;
        if (max_states > 0) out[tid] = a; // geodesic->push_back(cloneState(from))
      }
      rec[gTo + tid] = to[t * 14 + tid];
    }
    __syncthreads();
    int n = 1, its = 0, rounds = 0;
    bool resume = ent != nullptr; // the first pass through the loop below skips the interpolation and takes the projection's counters from the entry
    bool suspended = false; // the edge used up the call's budget of Newton rounds: it stops between two states (ok = 2)
    bool fits = true; // false: an accepted state found the list full — the edge stops there and reports max_states + 1
    bool target_ok = true;
    if (check_target) {
      // ConstrainedMotionValidator::checkMotion (src/planner/stefanBiPRM.cpp:397-398): isSatisfied(s2) first —
      // function(to) through one evaluation pass of the Newton routine (iteration cap 0: no update), then
      // KinematicChainConstraint::isSatisfied's test (finite, f0 <= tol1, f1 <= tol2; ConstraintFunction.h:114-120)
      if (tid < 14) rec[fX + tid] = rec[gTo + tid];
      __syncthreads();
      int iter0 = 0, upd0 = 0;
      double n1 = 0.0, n2 = 0.0;
      (void)flat_newton<STOCK>(K, KL, steptab, rec, tid, iter0, upd0, n1, n2, 0);
      const double f0 = rec[fF], f1 = rec[fF + 1];
      target_ok = (f0 - f0 == 0.0) && (f1 - f1 == 0.0) && f0 <= K.tol_pos && f1 <= K.tol_rot;
      __syncthreads();
    }
    double dist = lds_distance(rec + gPrev, rec + gTo), total = 0.0, total_before = 0.0;
    double maxd = dist * lambda;
    if (ent) { // (dist is the value the group kernel held: the distance of `previous` to the target, from the same operands)
      dist = ent[28]; total = ent[29]; maxd = ent[30];
      n = __double2hiint(ent[32]); its = __double2loint(ent[32]);
      rounds = __double2hiint(ent[33]);
    }
    // a continuation is in the middle of the reference's do-while: it re-enters on the loop's own condition
    // (dist >= delta) with the running length and the bound of the first call
    bool enter = dist > delta;
    if (carry_in) {
      total = carry_in[2 * t];
      maxd = carry_in[2 * t + 1];
      enter = dist >= delta;
    }
    if (ent) enter = true; // in the middle of the reference's loop
    if (target_ok && enter) {
      // Between two projections every thread does the reference's bookkeeping for itself, in ONE pass over the 14 joints
      // and without a barrier (round 3; before: jointValid through a ballot and two barriers, then the distances one
      // after the other — per state about as long as a Newton round): jointValid(x), step = |previous - x| and
      // newDist = |x - to| are accumulated side by side (two independent serial sums, the canonical order each), the
      // tests then run in the reference's order.  The joints' owners (tid < 14) keep x, previous and to in registers and
      // write the next interpolated state themselves: two block barriers per state instead of five.
      double x_own = 0.0, to_own = 0.0; // this thread's joint of the accepted state / of the target (tid < 14)
      if (tid < 14) { x_own = rec[gPrev + tid]; to_own = rec[gTo + tid]; }
      for (int guard = 0; guard < 1000000; guard++) { // the reference loop ends by itself; guard bounds a non-finite input
        if (tid < 14 && !resume) { // WrapperStateSpace::interpolate(previous, to, delta_ / dist, scratch)
          const double tt = delta / dist;
          const double fr = x_own;
          double diff = to_own - fr, v;
          if (ccmp_abs(diff) <= pi) v = CCMP_FMA(diff, tt, fr);
          else {
            if (diff > 0.0) diff = 2.0 * pi - diff;
            else diff = -2.0 * pi - diff;
            v = CCMP_FMA(-diff, tt, fr);
            if (v > pi) v -= 2.0 * pi;
            else if (v < -pi) v += 2.0 * pi;
          }
          rec[fX + tid] = v;
          rec[gPrev + tid] = fr; // previous := the accepted state (unchanged on the first pass)
        }
        __syncthreads();
        int iter = 0, updates = 0;
        double norm1 = 0.0, norm2 = 0.0;
        if (resume) { // the handed-over projection's own counters, read where they are needed
          iter = __double2loint(ent[33]); updates = __double2loint(ent[34]);
          norm1 = ent[35]; norm2 = ent[36];
          resume = false;
        }
        const bool conv = flat_newton<STOCK>(K, KL, steptab, rec, tid, iter, updates, norm1, norm2, K.max_iter);
        its += updates;
        rounds += updates + 1;
        // flat_newton leaves through a block barrier behind which nobody writes x any more: every thread reads the final
        // iterate, previous and the target straight from LDS
        bool jv = true;
        double s_acc = 0.0, d_acc = 0.0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
          const double xi = rec[fX + i];
          const int jj = i < 7 ? i : i - 7;
          if (xi < K.lbe[jj]) jv = false; // KinematicChainConstraint::jointValid (ConstraintFunction.h:43-55)
          if (xi > K.ube[jj]) jv = false;
          const double ds = rec[gPrev + i] - xi, dd = xi - rec[gTo + i];
          s_acc = CCMP_FMA(ds, ds, s_acc); // distance(previous, scratch)
          d_acc = CCMP_FMA(dd, dd, d_acc); // distance(scratch, to)
        }
        if (!(conv && jv)) break;                        // not on manifold
        const double step = ccmp_sqrt(s_acc), newDist = ccmp_sqrt(d_acc);
        if (step > lambda * delta) break;                // deviated
        total_before = total;
        total += step;
        if (total > maxd) break;                         // wandered too far
        if (newDist >= dist) break;                      // no closer than before
        // an edge that creeps (hundreds of accepted states, each a hair closer: seen at 1 in 16384 near-neighbour edges,
        // 952 states) must not hold the whole launch: when the list is full the edge stops and says so
        // (its running length and Newton count go back to what they were before this state: a continuation projects it again)
        if (n >= max_states) { fits = false; n = max_states + 1; total = total_before; its -= updates; break; }
        dist = newDist;
        if (tid < 14) {
          x_own = rec[fX + tid];
          out[(unsigned long long)n * 14ull + tid] = x_own;
        }
        n++;
        if (!(dist >= delta)) break;
        // A call bounds the serial work it spends on one edge: past round_budget Newton rounds the edge stops HERE — between
        // two states, where the reference's do-while has just found dist >= delta — and reports ok = 2; a continuation
        // from its last stored state with carry_out goes on exactly where this one stops (nothing is projected twice).
        // 16 384 near-neighbour edges, lists of 16: everything but one edge is through after 1.46 ms, that one creeping
        // edge needs 545 rounds for its 15 states and held the launch until 2.1 ms (profiles/r03_extend_timeline.log).
        if (round_budget > 0 && rounds >= round_budget) { suspended = true; break; }
        __syncthreads(); // everybody has read x and previous: their owners may overwrite them (top of the loop)
      }
    }
    if (tid == 0) {
      n_states[t] = n;
      ok_out[t] = suspended ? (uint8_t)2 : (uint8_t)(target_ok && fits && dist <= delta);
      if (newton_iters) newton_iters[t] = its;
      if (carry_out) { carry_out[2 * t] = total; carry_out[2 * t + 1] = maxd; }
#ifdef CCMP_GEO_TRACE
      if (t < 65536) g_geo_trace[3 * t + 1] = wall_clock64();
#endif
    }
    __syncthreads();
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
