// ccmp_kernels_flat.hip — the latency kernel of the reference-arithmetic path: one sample per 128-thread block,
// every residual evaluation of a Newton iteration in ONE round.  Canonical (bit-reproducible) rounding model:
// built -ffp-contract=off -DCCMP_USE_FMA like ccmp_kernels_fd.hip, and like it with -mllvm -disable-machine-licm
// (124 VGPRs and no scratch: four waves per SIMD, eight blocks per CU).
#include "ccmp_fd_common.h"

using namespace ccmp;

namespace {

constexpr int kConstsDoubles = (int)((sizeof(ccmp_consts) + 7) / 8);

// ------------------------------------------------------------------------------------------------
// "Flat" Newton: one sample per 128-thread block, every evaluation of an iteration in ONE round.
// Wave w of the block owns arm w.  Its lanes 0..41 own the 42 stencil points of that arm's 7 Jacobian columns
// (column lane / 6, point lane % 6) and each runs the WHOLE 7-joint chain of the arm with its joint perturbed — the
// same operations in the same order as re-entering at a cached prefix frame, hence the same bits; lane 42 runs the
// arm's chain at x; lanes 43..49 compute the sines/cosines of the arm's 7 joints.  function(x) and the 84
// evaluations of jacobian(x) are therefore computed side by side instead of one after the other (the Jacobian of
// the final iterate is computed and dropped): ~1.5 k dependent instructions per Newton iteration instead of ~3.3 k
// in the one-wavefront-per-sample kernel.  The arm is wave-uniform (no role selects); the per-joint constants come from
// an LDS step table read one joint ahead (scalar loads from the kernarg segment were 15 % slower here: one or two
// waves per SIMD cannot hide the scalar-cache round trips).  The stencil combination and the solve run on wave 0
// only; wave 1 waits at the barrier and leaves its issue slots to other blocks.
#ifdef CCMP_FLAT_TIMING
// phase timing of thread 0 of block 0 (tools/time_phases.py; never defined in the product build)
__device__ unsigned long long g_flat_timing[8];
__device__ __forceinline__ void flat_tick(int k, unsigned long long &prev)
{
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned long long now = __builtin_readcyclecounter();
    g_flat_timing[k] += now - prev;
    prev = now;
  }
}
#define FLAT_TICK(k) flat_tick(k, tprev)
#else
#define FLAT_TICK(k) do { } while (0)
#endif

constexpr int fX = 0, fSC = 14, fEE = 42, fF = 66, fJ = 68, fT = 96, fY = 264, fV = 348, fRec = 350;

// Per-joint constants of both arms in chain order, 12 doubles per joint (offset 3, axis 3, axis products 6), so that a
// joint's constants are six 16-byte LDS reads that can be issued one joint ahead of their use.
constexpr int kStepDoubles = 12, kStepTab = 2 * 7 * kStepDoubles;

__device__ __forceinline__ void stage_step_table(const ccmp_consts &K, double *tab, int tid)
{
  for (int k = tid; k < kStepTab; k += 128) {
    const int a = k / (7 * kStepDoubles), r = k - a * 7 * kStepDoubles, i = r / kStepDoubles, c = r - i * kStepDoubles;
    tab[k] = c < 3 ? K.offset[a][i][c] : (c < 6 ? K.axis[a][i][c - 3] : K.aprod[a][i][c - 6]);
  }
}

// joints I..6 of the chain of arm W with compile-time joint indices (STOCK: the exact-zero structure of the uncalibrated
// Panda is known to the compiler, ccmp_kin.h); the constants of joint I+1 are read while joint I is computed
template <int W, bool STOCK, int I>
__device__ __forceinline__ void flat_chain_from(const double2 *tab, const double2 *sct, int j, double s, double c, const double2 *cur,
                                                double2 sc_cur, double *R, double *o)
{
  if constexpr (I < 7) {
    double2 nxt[6], sc_nxt = sc_cur;
    if constexpr (I < 6) {
#pragma unroll
      for (int k = 0; k < 6; k++) nxt[k] = tab[6 * (I + 1) + k];
      sc_nxt = sct[I + 1];
    }
    double Rn[9];
    const double si = (I == j) ? s : sc_cur.x;
    const double ci = (I == j) ? c : sc_cur.y;
    const double off[3] = {cur[0].x, cur[0].y, cur[1].x};
    const double ax[3] = {cur[1].y, cur[2].x, cur[2].y};
    const double ap[6] = {cur[3].x, cur[3].y, cur[4].x, cur[4].y, cur[5].x, cur[5].y};
    chain_step<I, STOCK>(off, ax, ap, si, ci, R, Rn, o);
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
    flat_chain_from<W, STOCK, I + 1>(tab, sct, j, s, c, nxt, sc_nxt, R, o);
  }
}

template <int W, bool STOCK>
__device__ __forceinline__ void flat_chain_and_residual(const ccmp_consts &K, const ccmp_consts &KC, const double *steptab,
                                                        double *rec, int lane, int j, double s, double c, double y,
                                                        unsigned long long &tprev)
{
  (void)tprev; // only the -DCCMP_FLAT_TIMING build ticks it
  // ---- B: the chain of arm W, joint j at (s, c), the others at x --------------------------------------------------
  double Tw[12];
  {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
    const double2 *tab = reinterpret_cast<const double2 *>(steptab + W * 7 * kStepDoubles);
    const double2 *sct = reinterpret_cast<const double2 *>(rec + fSC + 2 * W * 7);
    double2 cur[6];
#pragma unroll
    for (int k = 0; k < 6; k++) cur[k] = tab[k];
    flat_chain_from<W, STOCK, 0>(tab, sct, j, s, c, cur, sct[0], R, o);
    tool_pose_t<STOCK>(KC, W, R, o, &Tw[0], &Tw[9]);
    if (lane == 42) {
#pragma unroll
      for (int k = 0; k < 12; k++) rec[fEE + W * 12 + k] = Tw[k];
    }
  }
  __syncthreads();
  FLAT_TICK(1);
  // ---- C: residuals against the partner arm's pose at x; lane 42 of wave 0 yields f(x) --------------------------
  {
    double To[12], tt[2];
#pragma unroll
    for (int k = 0; k < 12; k++) To[k] = rec[fEE + (1 - W) * 12 + k];
    if (W == 0) chain_residual(K, &Tw[0], &Tw[9], &To[0], &To[9], tt, nullptr, nullptr);
    else chain_residual(K, &To[0], &To[9], &Tw[0], &Tw[9], tt, nullptr, nullptr);
    if (lane < 42) {
      const int e = 6 * (W * 7 + j) + (lane - 6 * j);
      rec[fT + 2 * e] = tt[0];
      rec[fT + 2 * e + 1] = tt[1];
      rec[fY + e] = y;
    } else if (W == 0 && lane == 42) {
      rec[fF] = tt[0];
      rec[fF + 1] = tt[1];
    }
  }
}

template <bool STOCK>
__device__ __forceinline__ bool flat_newton(const ccmp_consts &K, const ccmp_consts &KC, const double *steptab, double *rec, int tid,
                                            int &iter, int &updates, double &norm1, double &norm2, int max_iter)
{
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform arm
  const int lane = tid & 63;
  const bool ev = lane < 42;                   // stencil evaluation
  const bool sc_lane = lane >= 43 && lane < 50; // sincos of joint lane - 43 of arm w
  const int j = ev ? lane / 6 : -1, pt = ev ? lane - 6 * j : 0;
  const bool plus = pt < 3;
  const int nstep = (plus ? pt : pt - 3) + 1;
  unsigned long long tprev = __builtin_readcyclecounter();
  for (;;) {
    // ---- A: angles.  y1[j] += h / y2[j] -= h by sequential adds as OMPL does; one sincos per lane -----------
    double y = 0.0, s, c;
    if (ev) {
      const double xj = rec[fX + w * 7 + j];
      const double axj = ccmp_abs(xj);
      const double h = 1.4901161193847656e-08 * (axj >= 1 ? axj : 1);
      const double hh = plus ? h : -h;
      y = xj + hh;
      if (nstep >= 2) y = y + hh;
      if (nstep >= 3) y = y + hh;
    } else if (sc_lane) {
      y = rec[fX + w * 7 + lane - 43];
    }
    ccmp_sincos(y, &s, &c);
    if (sc_lane) {
      rec[fSC + 2 * (w * 7 + lane - 43)] = s;
      rec[fSC + 2 * (w * 7 + lane - 43) + 1] = c;
    }
    __syncthreads();
    FLAT_TICK(0);
    if (w == 0) flat_chain_and_residual<0, STOCK>(K, KC, steptab, rec, lane, j, s, c, y, tprev);
    else flat_chain_and_residual<1, STOCK>(K, KC, steptab, rec, lane, j, s, c, y, tprev);
    __syncthreads();
    FLAT_TICK(2);
    const double f0 = rec[fF], f1 = rec[fF + 1];
    // ---- loop condition of ConstraintFunction.h:68 (block-uniform) ------------------------------------------
    bool cont = false;
    {
      const bool c1 = f0 > K.tol_pos;
      norm1 = c1 ? 1.0 : 0.0;
      bool resid = c1;
      if (!c1) { norm2 = f1; resid = f1 > K.tol_rot; }
      if (resid) { cont = iter < max_iter; iter++; }
    }
    if (!cont) return (norm1 < K.tol_pos) && (norm2 < K.tol_rot);
    // ---- D + E on wave 0 alone: the stencil combination needs 28 lanes, the min-norm solve is a serial computation
    // that every lane replicates — a second wave repeating it would only take issue slots from other blocks.  Wave 1
    // waits at the barrier below.  Inside one wave LDS operations execute in order: a fence orders the combination's
    // writes before the solve's reads, no s_barrier (wave 1 would never arrive at it).
    if (w == 0) {
      // J[row][col] = 1.5 m1 - 0.6 m2 + 0.1 m3, m_s = (t1 - t2) / (y1[j] - y2[j])
      if (lane < 28) {
        const int row = lane >= 14 ? 1 : 0, cc = lane - 14 * row;
        double m[3];
#pragma unroll
        for (int sidx = 0; sidx < 3; sidx++) {
          const int e1 = 6 * cc + sidx, e2 = 6 * cc + 3 + sidx;
          m[sidx] = (rec[fT + 2 * e1 + row] - rec[fT + 2 * e2 + row]) / (rec[fY + e1] - rec[fY + e2]);
        }
        rec[fJ + lane] = CCMP_FMA(0.1, m[2], CCMP_FMA(-0.6, m[1], 1.5 * m[0]));
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
      FLAT_TICK(3);
      // minimum-norm step
      double Jr[28], dx[14];
#pragma unroll
      for (int k = 0; k < 28; k++) Jr[k] = rec[fJ + k];
      solve_minnorm(Jr, f0, f1, dx);
      // lane e < 14 moves x[e]: its dx is picked by a select chain in registers and x is touched once (fourteen
      // predicated LDS read-modify-writes in a row cost ~1400 cycles of LDS round trips)
      double mine = dx[0];
#pragma unroll
      for (int e = 1; e < 14; e++) mine = (lane == e) ? dx[e] : mine;
      if (lane < 14) rec[fX + lane] = CCMP_FMA(-K.step, mine, rec[fX + lane]);
    }
    updates++;
    __syncthreads();
    FLAT_TICK(4);
  }
}

// jointValid(x) of the iterate in rec[fX..] (ConstraintFunction.h:43-55); every thread gets the result through LDS
__device__ __forceinline__ bool flat_joint_valid(const ccmp_consts &KL, double *rec, int tid)
{
  bool bad = false;
  if (tid < 14) {
    const double v = rec[fX + tid];
    const int jj = tid < 7 ? tid : tid - 7;
    if (v < KL.lbe[jj]) bad = true;
    if (v > KL.ube[jj]) bad = true;
  }
  const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull; // threads 0..13 sit in wave 0
  if (tid == 0) rec[fV] = any_bad ? 1.0 : 0.0; // its own slot: a slower wave may still be reading f(x) from fF
  __syncthreads();
  const bool jv = rec[fV] == 0.0;
  __syncthreads();
  return jv;
}

// SRC 0: q_in, SRC 1: ambient sampler, SRC 2: straggler pool.  queue == nullptr: static striding (one block per
// sample launches need no queue reset).
template <int SRC, bool STOCK>
__global__ __launch_bounds__(128, 2) void project_fd_flat_kernel(
    const ccmp_consts K, const double *__restrict__ q_in, double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
    uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient, unsigned long long B, unsigned long long *queue,
    unsigned long long seed, unsigned long long first_index, const double *__restrict__ pool,
    const unsigned long long *__restrict__ pool_count, int wrap_output, unsigned int *done_flag, unsigned int done_seq)
{
  __shared__ __attribute__((aligned(16))) double lds[fRec];
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
  const unsigned long long total = (SRC == 2) ? *pool_count : B;
  unsigned long long t = blockIdx.x;

  for (;;) {
    if (queue) {
      if (tid == 0) ticket = atomicAdd(queue, 1ull);
      __syncthreads();
      t = ticket;
    }
    if (t >= total) break;
    unsigned long long idx;
    int iter = 0, updates = 0;
    double norm1 = 0.0, norm2 = 0.0;
    if (SRC == 2) {
      const double *ent = pool + t * kPoolEntry;
      idx = (unsigned long long)__double_as_longlong(ent[14]);
      iter = __double2hiint(ent[15]);
      updates = __double2loint(ent[15]);
      norm1 = ent[16];
      norm2 = ent[17];
      if (tid < 14) rec[fX + tid] = ent[tid];
    } else {
      idx = t;
      if (tid < 14) {
        double v;
        if (SRC == 0) v = q_in[idx * 14 + tid];
        else {
          v = ambient_uniform(KL, seed, first_index + idx, tid);
          if (q_ambient) q_ambient[idx * 14 + tid] = v;
        }
        rec[fX + tid] = v;
      }
    }
    __syncthreads();
    const bool conv = flat_newton<STOCK>(K, KL, steptab, rec, tid, iter, updates, norm1, norm2, K.max_iter);
    const bool jv = flat_joint_valid(KL, rec, tid);
    if (tid < 14) {
      const double v = rec[fX + tid];
      q_out[idx * 14 + tid] = wrap_output ? wrap_pi(v) : v;
    }
    if (tid == 0) {
      ok_out[idx] = (uint8_t)(jv && conv);
      if (iters_out) iters_out[idx] = (uint16_t)updates;
    }
    __syncthreads();
    if (!queue) t += gridDim.x;
  }
  // single-state calls through the host entry point (one block): the results sit in pinned host memory; publish a
  // sequence number behind them so that the host can poll a word instead of waiting for the stream's completion signal
  if (done_flag != nullptr && tid == 0) {
    __threadfence_system();
    __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------------------------------------
// geodesic_flat_kernel — jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:32-96), one
// 128-thread block per edge (from -> to): interpolate a step of delta towards `to` (KinematicChainSpace::interpolate,
// KinematicChain.h:145-171), project it (flat_newton), apply the reference's four break tests, record the state.
// The StateValidityChecker (MoveIt collision) stays on the host: the kernel runs as the reference does with
// interpolate == true and the host truncates the list at the first invalid state, which is what the reference's
// break would have produced.
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
__global__ __launch_bounds__(128, 2) void geodesic_flat_kernel(
    const ccmp_consts K, const double delta, const double lambda, const double *__restrict__ from,
    const double *__restrict__ to, unsigned long long E, int max_states, double *__restrict__ states,
    int *__restrict__ n_states, uint8_t *__restrict__ ok_out, int *__restrict__ newton_iters, int check_target)
{
  __shared__ __attribute__((aligned(16))) double lds[gRec];
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ __attribute__((aligned(16))) double steptab[kStepTab];
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

  for (unsigned long long t = blockIdx.x; t < E; t += gridDim.x) {
    double *out = states + t * (unsigned long long)max_states * 14ull;
    if (tid < 14) {
      const double a = from[t * 14 + tid];
      rec[gPrev + tid] = a;
      rec[gTo + tid] = to[t * 14 + tid];
      if (max_states > 0) out[tid] = a; // geodesic->push_back(cloneState(from))
    }
    __syncthreads();
    int n = 1, its = 0;
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
    double dist = lds_distance(rec + gPrev, rec + gTo), total = 0.0;
    if (target_ok && dist > delta) {
      const double maxd = dist * lambda;
      for (int guard = 0; guard < 1000000; guard++) { // the reference loop ends by itself; guard bounds a non-finite input
        if (tid < 14) { // WrapperStateSpace::interpolate(previous, to, delta_ / dist, scratch)
          const double tt = delta / dist;
          const double fr = rec[gPrev + tid];
          double diff = rec[gTo + tid] - fr, v;
          if (ccmp_abs(diff) <= pi) v = CCMP_FMA(diff, tt, fr);
          else {
            if (diff > 0.0) diff = 2.0 * pi - diff;
            else diff = -2.0 * pi - diff;
            v = CCMP_FMA(-diff, tt, fr);
            if (v > pi) v -= 2.0 * pi;
            else if (v < -pi) v += 2.0 * pi;
          }
          rec[fX + tid] = v;
        }
        __syncthreads();
        int iter = 0, updates = 0;
        double norm1 = 0.0, norm2 = 0.0;
        const bool conv = flat_newton<STOCK>(K, KL, steptab, rec, tid, iter, updates, norm1, norm2, K.max_iter);
        const bool jv = flat_joint_valid(KL, rec, tid);
        its += updates;
        if (!(conv && jv)) break;                        // not on manifold
        const double step = lds_distance(rec + gPrev, rec + fX);
        if (step > lambda * delta) break;                // deviated
        total += step;
        if (total > maxd) break;                         // wandered too far
        const double newDist = lds_distance(rec + fX, rec + gTo);
        if (newDist >= dist) break;                      // no closer than before
        // an edge that creeps (hundreds of accepted states, each a hair closer: seen at 1 in 16384 near-neighbour edges,
        // 952 states) must not hold the whole launch: when the list is full the edge stops and says so
        if (n >= max_states) { fits = false; n = max_states + 1; break; }
        dist = newDist;
        __syncthreads();
        if (tid < 14) {
          const double v = rec[fX + tid];
          rec[gPrev + tid] = v;
          out[(unsigned long long)n * 14ull + tid] = v;
        }
        n++;
        __syncthreads();
        if (!(dist >= delta)) break;
      }
    }
    if (tid == 0) {
      n_states[t] = n;
      ok_out[t] = (uint8_t)(target_ok && fits && dist <= delta);
      if (newton_iters) newton_iters[t] = its;
    }
    __syncthreads();
  }
}

} // namespace

extern "C" {

// one sample per 128-thread block, all evaluations of an iteration in one round; queue_head may be NULL (static striding)
hipError_t ccmp_launch_project_flat(const ccmp_consts *K, int src, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue_head,
                                    unsigned long long seed, unsigned long long first, const double *pool,
                                    const unsigned long long *pool_count, int wrap_output, int nblocks, unsigned int *done_flag,
                                    unsigned int done_seq, hipStream_t st)
{
  if (nblocks != 1) done_flag = nullptr; // the completion word is written by the one block of a single-state call
#define CCMP_LAUNCH_FLAT(SRC, STOCK)                                                                                             \
  hipLaunchKernelGGL((project_fd_flat_kernel<SRC, STOCK>), dim3(nblocks), dim3(128), 0, st, *K, q_in, q_out, ok, iters, q_ambient, \
                     (unsigned long long)B, queue_head, seed, first, pool, pool_count, wrap_output, done_flag, done_seq)
  if (src == 0) {
    if (K->stock) CCMP_LAUNCH_FLAT(0, true);
    else CCMP_LAUNCH_FLAT(0, false);
  } else if (src == 1) {
    if (K->stock) CCMP_LAUNCH_FLAT(1, true);
    else CCMP_LAUNCH_FLAT(1, false);
  } else {
    if (K->stock) CCMP_LAUNCH_FLAT(2, true);
    else CCMP_LAUNCH_FLAT(2, false);
  }
#undef CCMP_LAUNCH_FLAT
  return hipGetLastError();
}

hipError_t ccmp_launch_geodesic(const ccmp_consts *K, double delta, double lambda, const double *from, const double *to,
                                size_t E, int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters,
                                int check_target, int nblocks, hipStream_t st)
{
  if (K->stock)
    hipLaunchKernelGGL(geodesic_flat_kernel<true>, dim3(nblocks), dim3(128), 0, st, *K, delta, lambda, from, to, (unsigned long long)E,
                       max_states, states, n_states, ok, newton_iters, check_target);
  else
    hipLaunchKernelGGL(geodesic_flat_kernel<false>, dim3(nblocks), dim3(128), 0, st, *K, delta, lambda, from, to, (unsigned long long)E,
                       max_states, states, n_states, ok, newton_iters, check_target);
  return hipGetLastError();
}

#ifdef CCMP_FLAT_TIMING
hipError_t ccmp_debug_flat_timing(unsigned long long *out8, int reset)
{
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_flat_timing), 8 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_flat_timing), z, sizeof z);
  }
  return e;
}
#endif

} // extern "C"
