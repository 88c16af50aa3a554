// ccmp_kernels_dense.hip — dense latency blocks (round 4): S samples (or edges) per block on 2 S wavefronts, the 2 x 14
// min-norm solves of all of them side by side in ONE wavefront, a slot-per-sample state machine.
//
// Why.  The latency kernels (ccmp_kernels_flat.hip, ccmp_kernels_geo.hip) give a sample a 128-thread block: wave w runs arm
// w's 42 stencil evaluations, and between two rounds wave 0 runs the serial min-norm solve (two Jacobi sweeps, eight
// divisions, six square roots: ~530 of the ~1 620 instructions on its path) on 14 of its 64 lanes while wave 1 waits.
// Loaded, the kernel is issue-bound (DESIGN_experiments.md 5.6: +23 % instructions on the waiting wave cost +21 % at 4 096
// samples), and the solve is the one part of a round whose instruction count does not depend on how many samples it serves:
// lane c of a 16-lane ROW owns column c, so a wavefront has room for four systems.  Here a block holds S slots (S = 2: 256
// threads, 4 blocks per CU; S = 4: 512 threads, 2 per CU), slot s = waves 2 s and 2 s + 1 exactly as a flat block's two
// waves, and wave 1 solves every running slot's system in its row s: 530 instructions per round and BLOCK instead of per
// sample.  The slots run their rounds in lock-step (three block barriers per round, as before); they do NOT run their samples
// in lock-step: a slot whose sample has converged finalises it and takes the next ticket while the others go on
// (slot master = the slot's arm-0 wavefront: loop test, jointValid, outputs, refill — all wave-local).
//
// Same arithmetic, same order (ccmp_flat_newton.h's round; the solve is flat_newton's part E with the wave-wide readlane
// replaced by a row broadcast): bit-identical to the flat kernels and to the det oracle.
// Built like ccmp_kernels_flat.hip (-ffp-contract=off -DCCMP_USE_FMA, machine LICM off, max-ilp).
#include "ccmp_flat_newton.h"

namespace {

// ---- a slot's record: the flat record (fRec doubles) + the extend step's previous / target states ------------------------
constexpr int dPrev = fRec, dTo = fRec + 14, fSlot = (fRec + 28 + 1) & ~1;
// control words of a slot (LDS ints): run (the slot holds a sample in its Newton loop), iter (ConstraintFunction.h:68's
// counter BEFORE this round's test), cap (its iteration cap: max_iter, 0 for the extend step's isSatisfied(to) pass)
constexpr int cRun = 0, cIter = 1, cCap = 2, cWords = 4;

template <int N>
__device__ __forceinline__ double row_value(double v) // the value lane N of the lane's own 16-lane row holds
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + N, 0xf, 0xf, false); // row_newbcast:N
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Part E of flat_newton for S systems at once: row r of the wavefront takes slot r — lane c < 14 of the row owns column c,
// lanes 0, 1, 2 of the row form the three serial sums of a Jacobi sweep in solve_minnorm's order (ccmp_solve.h) and hand them
// round the ROW.  A row whose slot takes no step this round (idle, converged, cap reached) computes on whatever its record
// holds and writes nothing.
// `may_step`: the row's slot holds a sample whose iteration counter is still below its cap — read from the control words at the
// TOP of the round (the slot masters rewrite them while this runs).
template <int S>
__device__ __forceinline__ void dense_solve(const ccmp_consts &K, double *lds, bool may_step, int lane)
{
  const int row = lane >> 4, rl = lane & 15;
  const int r = row < S ? row : S - 1;
  double *rec = lds + r * fSlot;
  const double f0 = rec[fF], f1 = rec[fF + 1];
  const bool resid = (f0 > K.tol_pos) || (f1 > K.tol_rot);
  const bool act = may_step && resid; // the slot master's `cont`, from the same operands
  if (__builtin_amdgcn_ballot_w64(act) == 0ull) return;
  const double *jx = rec + fJ + 2 * (rl < 2 ? rl : 2);
  const int me = rl < 14 ? rl : 13;
  const bool wr = act && rl < 14;
  double2 mine = *reinterpret_cast<const double2 *>(rec + fJ + kJCol * me + kJPair);
  double g0 = f0, g1 = f1;
#pragma unroll
  for (int sweep = 0; sweep < 2; sweep++) {
    const double acc = column_sum(jx);
    const double a = row_value<0>(acc), d = row_value<1>(acc), b = row_value<2>(acc);
    if (b != 0.0) { // the same value in every lane of a row
      const double zeta = (d - a) / (2.0 * b);
      double t = 1.0 / (ccmp_abs(zeta) + ccmp_sqrt(CCMP_FMA(zeta, zeta, 1.0)));
      if (zeta < 0.0) t = -t;
      const double cr = 1.0 / ccmp_sqrt(CCMP_FMA(t, t, 1.0));
      const double sr = cr * t;
      const double v0 = mine.x, v1 = mine.y;
      mine.x = CCMP_FMA(cr, v0, -(sr * v1));
      mine.y = CCMP_FMA(sr, v0, cr * v1);
      const double h0 = g0, h1 = g1;
      g0 = CCMP_FMA(cr, h0, -(sr * h1));
      g1 = CCMP_FMA(sr, h0, cr * h1);
      if (wr) store_column(rec + fJ + kJCol * rl, mine.x, mine.y);
    }
    wave_lds_fence();
  }
  const double acc = column_sum(jx);
  const double root = ccmp_sqrt(acc), quot = (rl == 0 ? g0 : g1) / acc;
  const double s0 = row_value<0>(root), s1 = row_value<1>(root);
  const double smax = s0 > s1 ? s0 : s1;
  double thr = smax * (2.0 * 2.220446049250313e-16);
  if (thr < 2.2250738585072014e-308) thr = 2.2250738585072014e-308;
  const double k0 = s0 > thr ? row_value<0>(quot) : 0.0;
  const double k1 = s1 > thr ? row_value<1>(quot) : 0.0;
  const double dxm = CCMP_FMA(k1, mine.y, k0 * mine.x);
  if (wr) rec[fX + rl] = CCMP_FMA(-K.step, dxm, rec[fX + rl]);
}

// One Newton round of a block up to the point where f(x) and J(x) of every running slot sit in its record: angles, chains,
// barrier, residuals + stencil, barrier.  Idle slots' waves only keep the barriers.
template <bool STOCK>
__device__ __forceinline__ void dense_round(const ccmp_consts &K, const ccmp_consts &KC, const double *steptab, double *rec, int arm, int lane,
                                            const FlatLane &L, bool run)
{
  double Tw[12], y = 0.0;
  if (run) {
    y = flat_angles(steptab, rec, L);
    wave_lds_fence(); // an arm's sines and cosines are written and read by the arm's own wave
    if (arm == 0) flat_chain_pose<0, STOCK>(KC, steptab, rec, lane, L.j, L.own, Tw);
    else flat_chain_pose<1, STOCK>(KC, steptab, rec, lane, L.j, L.own, Tw);
  }
  __syncthreads();
  if (run) {
    if (arm == 0) flat_residual_stencil<0, STOCK>(K, rec, lane, L.j, L.head, y, Tw);
    else flat_residual_stencil<1, STOCK>(K, rec, lane, L.j, L.head, y, Tw);
  }
  __syncthreads();
}

__device__ __forceinline__ unsigned long long wave_ticket(unsigned long long *queue, int lane)
{
  unsigned long long t = 0ull;
  if (lane == 0) t = atomicAdd(queue, 1ull);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(t & 0xffffffffull));
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(t >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// ------------------------------------------------------------------------------------------------------------------------
// project_fd_dense_kernel — KinematicChainConstraint::project (ConstraintFunction.h:57-82), S samples per block.
// SRC 0: q_in, SRC 1: ambient sampler, SRC 2: straggler pool (as project_fd_flat_kernel); samples come through an atomic
// ticket per slot.
template <int SRC, bool STOCK, int S>
__global__ __launch_bounds__(128 * S, CCMP_FLAT_MIN_WAVES) void project_fd_dense_kernel(
    const ccmp_consts K, const double *__restrict__ q_in, double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
    uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient, unsigned long long B, unsigned long long *queue,
    unsigned long long seed, unsigned long long first_index, const double *__restrict__ pool,
    const unsigned long long *__restrict__ pool_count, int wrap_output, unsigned long long pool_records)
{
  __shared__ __attribute__((aligned(16))) double lds[S * fSlot];
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ __attribute__((aligned(16))) double steptab[kStepTab];
  __shared__ __attribute__((aligned(16))) int ctl[S * cWords];
  const int tid = threadIdx.x;
  {
    const double *src = reinterpret_cast<const double *>(&K);
    for (int k = tid; k < kConstsDoubles; k += 128 * S) ktab[k] = src[k];
  }
  if (tid < 128) stage_step_table(K, steptab, tid); // its loop strides by 128
  const ccmp_consts &KL = *reinterpret_cast<const ccmp_consts *>(ktab);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), slot = w >> 1, arm = w & 1, lane = tid & 63;
  double *rec = lds + slot * fSlot;
  int *myctl = ctl + slot * cWords;
  const FlatLane L = flat_lane(arm, lane);
  const bool master = arm == 0;
  const unsigned long long n_front = (SRC == 2) ? pool_count[0] : 0ull;
  const unsigned long long total = (SRC == 2) ? n_front + (pool_records ? pool_count[5] : 0ull) : B;

  // slot master's state of the sample in the slot
  unsigned long long idx = 0ull;
  int iter = 0, updates = 0;
  double norm1 = 0.0, norm2 = 0.0;
  bool run = false;

  // takes the next ticket for the slot: x into the record, counters reset (or taken from the pool record)
  auto refill = [&]() {
    const unsigned long long t = wave_ticket(queue, lane);
    run = t < total;
    iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
    if (run) {
      if (SRC == 2) {
        const double *ent = pool + (t < n_front ? t : pool_records - 1ull - (t - n_front)) * kPoolEntry;
        idx = (unsigned long long)__double_as_longlong(ent[14]);
        iter = __double2hiint(ent[15]);
        updates = __double2loint(ent[15]);
        norm1 = ent[16];
        norm2 = ent[17];
        if (lane < 14) rec[fX + lane] = ent[lane];
      } else {
        idx = t;
        if (lane < 14) {
          double v;
          if (SRC == 0) v = q_in[idx * 14 + lane];
          else {
            v = ambient_uniform(KL, seed, first_index + idx, lane);
            if (q_ambient) q_ambient[idx * 14 + lane] = v;
          }
          rec[fX + lane] = v;
        }
      }
    }
    if (lane == 0) { myctl[cRun] = run ? 1 : 0; myctl[cIter] = iter; myctl[cCap] = K.max_iter; }
  };

  __syncthreads(); // tables staged (KL is read by the sampler)
  if (master) refill();
  __syncthreads();
  for (;;) {
    bool any = false;
#pragma unroll
    for (int s = 0; s < S; s++) any = any || ctl[s * cWords + cRun] != 0;
    if (!any) break;
    const bool my_run = myctl[cRun] != 0;
    // the solver's view of its rows' slots, taken now: the masters rewrite the control words behind the round's second barrier
    const int srow = (lane >> 4) < S ? (lane >> 4) : S - 1;
    const bool may_step = (lane >> 4) < S && ctl[srow * cWords + cRun] != 0 && ctl[srow * cWords + cIter] < ctl[srow * cWords + cCap];
    dense_round<STOCK>(K, KL, steptab, rec, arm, lane, L, my_run);
    if (w == 1) dense_solve<S>(K, lds, may_step, lane);
    if (master && my_run) {
      // loop condition of ConstraintFunction.h:68 (slot-uniform): (norm1 = f0 > tol1) || (norm2 = f1) > tol2 — norm2 is assigned
      // only when the first test fails; iter++ only when the residual test holds
      const double f0 = rec[fF], f1 = rec[fF + 1];
      const bool c1 = f0 > K.tol_pos;
      const bool resid = c1 || (f1 > K.tol_rot);
      norm1 = c1 ? 1.0 : 0.0;
      norm2 = c1 ? norm2 : f1;
      const bool cont = resid && iter < K.max_iter;
      iter += resid ? 1 : 0;
      if (cont) {
        updates++;
        if (lane == 0) myctl[cIter] = iter;
      } else {
        // the sample is through: jointValid (ConstraintFunction.h:43-55) on the final iterate, outputs, next ticket.  Nobody
        // writes this slot's x in this round (the solver's `act` is this `cont`).
        const bool conv = (norm1 < K.tol_pos) && (norm2 < K.tol_rot);
        bool bad = false;
        double v = 0.0;
        if (lane < 14) {
          v = rec[fX + lane];
          const int jj = lane < 7 ? lane : lane - 7;
          if (v < KL.lbe[jj]) bad = true;
          if (v > KL.ube[jj]) bad = true;
          q_out[idx * 14 + lane] = wrap_output ? wrap_pi(v) : v;
        }
        const bool jv = __builtin_amdgcn_ballot_w64(bad) == 0ull;
        if (lane == 0) {
          ok_out[idx] = (uint8_t)(jv && conv);
          if (iters_out) iters_out[idx] = (uint16_t)updates;
        }
        refill();
      }
    }
    __syncthreads(); // x of the next round (solver's update or the master's next sample) and the control words are in place
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// geodesic_dense_kernel — jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:32-96), S edges per
// block.  What geodesic_flat_kernel (ccmp_kernels_geo.hip) does with a block per edge, as a slot state machine: the slot
// master walks its edge — interpolate a delta step (KinematicChainSpace::interpolate, KinematicChain.h:145-171), let the
// block's rounds project it, apply the reference's four break tests, record the state — between two rounds, wave-locally;
// isSatisfied(to) (check_target) is one round with an iteration cap of zero.  Resumable exactly as the flat kernel
// (carry_in / carry_out / round_budget: same values at the same places), so first call + continuation on EITHER kernel give the
// states of one uninterrupted traversal.
__device__ __forceinline__ double dense_distance(const double *a, const double *b)
{
  double dist = 0.0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    const double diff = a[i] - b[i];
    dist = CCMP_FMA(diff, diff, dist);
  }
  return ccmp_sqrt(dist);
}

// A wave-uniform condition as the scalar unit sees it: the slot masters' bookkeeping branches on values read from LDS (the
// same in every lane, which the compiler cannot know); through readfirstlane the branches are scalar and the counters that
// change under them stay in scalar registers.
__device__ __forceinline__ bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }

template <bool STOCK, int S>
__global__ __launch_bounds__(128 * S, CCMP_FLAT_MIN_WAVES) void geodesic_dense_kernel(
    const ccmp_consts K, const double delta, const double lambda, const double *__restrict__ from,
    const double *__restrict__ to, unsigned long long E, int max_states, double *__restrict__ states,
    int *__restrict__ n_states, uint8_t *__restrict__ ok_out, int *__restrict__ newton_iters, int check_target,
    unsigned long long *queue, const unsigned int *__restrict__ order, const double *__restrict__ carry_in,
    double *__restrict__ carry_out, int round_budget)
{
  __shared__ __attribute__((aligned(16))) double lds[S * fSlot];
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ __attribute__((aligned(16))) double steptab[kStepTab];
  __shared__ __attribute__((aligned(16))) int ctl[S * cWords];
  // the slot masters' floating-point state between two rounds (registers are what eight blocks per CU live on; these are
  // touched once per round): dist, total, maxd, norm2
  __shared__ __attribute__((aligned(16))) double mst[S * 4];
  const int tid = threadIdx.x;
  {
    const double *src = reinterpret_cast<const double *>(&K);
    for (int k = tid; k < kConstsDoubles; k += 128 * S) ktab[k] = src[k];
  }
  if (tid < 128) stage_step_table(K, steptab, tid);
  const ccmp_consts &KL = *reinterpret_cast<const ccmp_consts *>(ktab);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), slot = w >> 1, arm = w & 1, lane = tid & 63;
  double *rec = lds + slot * fSlot;
  int *myctl = ctl + slot * cWords;
  double *ms = mst + slot * 4;
  const FlatLane L = flat_lane(arm, lane);
  const bool master = arm == 0;
  const double pi = 3.14159265358979323846;

  // slot master's state (scalar): the edge ...
  unsigned long long t = 0ull;
  int n = 1, its = 0, rounds = 0;
  bool suspended = false, fits = true, target_ok = true, target_pass = false;
  // ... and the projection in flight
  int iter = 0, updates = 0, cap = 0;

  if (tid < S) { ctl[tid * cWords + cRun] = 2; ctl[tid * cWords + cIter] = 0; ctl[tid * cWords + cCap] = 0; } // 2: the slot wants its first edge
  __syncthreads();
  for (;;) {
    bool any = false;
#pragma unroll
    for (int s = 0; s < S; s++) any = any || ctl[s * cWords + cRun] != 0;
    if (!any) break;
    const int my_state = __builtin_amdgcn_readfirstlane(myctl[cRun]);
    const bool my_run = my_state == 1;
    const int srow = (lane >> 4) < S ? (lane >> 4) : S - 1;
    const bool may_step = (lane >> 4) < S && ctl[srow * cWords + cRun] == 1 && ctl[srow * cWords + cIter] < ctl[srow * cWords + cCap];
    dense_round<STOCK>(K, KL, steptab, rec, arm, lane, L, my_run);
    if (w == 1) dense_solve<S>(K, lds, may_step, lane);
    if (master && my_state != 0) {
      // What the master does between two rounds, as ONE pass through a small state machine (every step is inlined once):
      // aNone: the projection goes on; aNext: interpolate the next state and project it; aFinish: write the edge's results;
      // aTicket: take the next edge; aBegin: the edge proper (behind the optional isSatisfied(to) round)
      enum { aNone, aNext, aFinish, aTicket, aBegin };
      int action = aTicket; // my_state == 2: first edge
      int acc_at = dPrev;   // where the accepted state sits from which the next one is interpolated
      double dist = 0.0, total = 0.0, maxd = 0.0;
      bool run = true;
      if (my_run) {
        const double f0 = rec[fF], f1 = rec[fF + 1];
        const bool c1 = f0 > K.tol_pos;
        const bool resid = uni(c1 || (f1 > K.tol_rot));
        const double norm1 = c1 ? 1.0 : 0.0;
        const double norm2 = c1 ? ms[3] : f1;
        const bool cont = resid && iter < cap;
        iter += resid ? 1 : 0;
        if (cont) {
          updates++;
          if (lane == 0) ms[3] = norm2;
          action = aNone;
        } else if (target_pass) {
          // KinematicChainConstraint::isSatisfied's test (finite, f0 <= tol1, f1 <= tol2; ConstraintFunction.h:114-120)
          target_ok = uni((f0 - f0 == 0.0) && (f1 - f1 == 0.0) && f0 <= K.tol_pos && f1 <= K.tol_rot);
          action = aBegin;
        } else {
          const bool conv = (norm1 < K.tol_pos) && (norm2 < K.tol_rot);
          dist = ms[0]; total = ms[1]; maxd = ms[2];
          its += updates;
          rounds += updates + 1;
          // the reference's bookkeeping between two projections, in one pass over the 14 joints: jointValid(x), step =
          // |previous - x| and newDist = |x - to| side by side (two independent serial sums, the canonical order each), then the
          // tests in the reference's order (jy_ProjectedStateSpace.cpp:65-84)
          bool jv = true;
          double s_acc = 0.0, d_acc = 0.0;
#pragma unroll
          for (int i = 0; i < 14; i++) {
            const double xi = rec[fX + i];
            const int jj = i < 7 ? i : i - 7;
            if (xi < KL.lbe[jj]) jv = false; // KinematicChainConstraint::jointValid (ConstraintFunction.h:43-55)
            if (xi > KL.ube[jj]) jv = false;
            const double ds = rec[dPrev + i] - xi, dd = xi - rec[dTo + i];
            s_acc = CCMP_FMA(ds, ds, s_acc); // distance(previous, scratch)
            d_acc = CCMP_FMA(dd, dd, d_acc); // distance(scratch, to)
          }
          action = aFinish;
          if (uni(conv && jv)) {                               // else: not on manifold
            const double step = ccmp_sqrt(s_acc), newDist = ccmp_sqrt(d_acc);
            if (uni(!(step > lambda * delta))) {               // else: deviated
              const double total_before = total;
              total += step;
              if (uni(!(total > maxd) && !(newDist >= dist))) { // else: wandered too far / no closer than before
                if (n >= max_states) { // the list is full: the edge stops here and says so (ccmp_kernels_geo.hip)
                  fits = false; n = max_states + 1; total = total_before; its -= updates;
                } else {
                  dist = newDist;
                  if (lane < 14) states[(t * (unsigned long long)max_states + (unsigned long long)n) * 14ull + lane] = rec[fX + lane];
                  n++;
                  if (uni(dist >= delta)) {
                    // the call's budget of Newton rounds for one edge is spent: it stops between two states (ok = 2)
                    if (round_budget > 0 && rounds >= round_budget) suspended = true;
                    else { action = aNext; acc_at = fX; }
                  }
                }
              }
            }
          }
        }
      }
      while (action != aNone) {
        if (action == aFinish) {
          if (lane == 0) {
            n_states[t] = n;
            ok_out[t] = suspended ? (uint8_t)2 : (uint8_t)(target_ok && fits && dist <= delta);
            if (newton_iters) newton_iters[t] = its;
            if (carry_out) { carry_out[2 * t] = total; carry_out[2 * t + 1] = maxd; }
          }
          action = aTicket;
        }
        if (action == aTicket) {
          const unsigned long long tk = wave_ticket(queue, lane);
          if (tk >= E) { run = false; iter = 0; cap = 0; break; }
          t = tk;
          if (order) t = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)order[tk]);
          if (lane < 14) {
            const double a = from[t * 14 + lane];
            rec[dPrev + lane] = a;
            rec[dTo + lane] = to[t * 14 + lane];
            if (max_states > 0) states[t * (unsigned long long)max_states * 14ull + lane] = a; // geodesic->push_back(cloneState(from))
          }
          n = 1; its = 0; rounds = 0;
          suspended = false; fits = true; target_ok = true;
          wave_lds_fence(); // previous / target were written by this wave's lanes < 14
          if (check_target) {
            // ConstrainedMotionValidator::checkMotion (src/planner/stefanBiPRM.cpp:397-398): isSatisfied(s2) first —
            // function(to) through one evaluation round (iteration cap 0: no update)
            if (lane < 14) rec[fX + lane] = rec[dTo + lane];
            if (lane == 0) ms[3] = 0.0;
            iter = 0; updates = 0; cap = 0;
            target_pass = true;
            break;
          }
          action = aBegin;
        }
        if (action == aBegin) { // the edge proper
          dist = dense_distance(rec + dPrev, rec + dTo);
          total = 0.0;
          maxd = dist * lambda;
          // a continuation is in the middle of the reference's do-while: it re-enters on the loop's own condition
          // (dist >= delta) with the running length and the bound of the first call
          bool enter = dist > delta;
          if (carry_in) {
            total = carry_in[2 * t];
            maxd = carry_in[2 * t + 1];
            enter = dist >= delta;
          }
          acc_at = dPrev;
          action = uni(target_ok && enter) ? aNext : aFinish;
        }
        if (action == aNext) {
          // WrapperStateSpace::interpolate(previous, to, delta_ / dist, scratch) by the joints' owners; previous := the
          // accepted state (the lane's joint of it sits at rec[acc_at + lane])
          if (lane < 14) {
            const double tt = delta / dist;
            const double fr = rec[acc_at + lane];
            double diff = rec[dTo + lane] - fr, v;
            if (ccmp_abs(diff) <= pi) v = CCMP_FMA(diff, tt, fr);
            else {
              if (diff > 0.0) diff = 2.0 * pi - diff;
              else diff = -2.0 * pi - diff;
              v = CCMP_FMA(-diff, tt, fr);
              if (v > pi) v -= 2.0 * pi;
              else if (v < -pi) v += 2.0 * pi;
            }
            rec[fX + lane] = v;
            rec[dPrev + lane] = fr;
          }
          if (lane == 0) { ms[0] = dist; ms[1] = total; ms[2] = maxd; ms[3] = 0.0; }
          iter = 0; updates = 0;
          cap = K.max_iter;
          target_pass = false;
          break;
        }
      }
      if (lane == 0) { myctl[cRun] = run ? 1 : 0; myctl[cIter] = iter; myctl[cCap] = cap; }
    }
    __syncthreads();
  }
}

} // namespace

extern "C" {

hipError_t ccmp_launch_geodesic_dense(const ccmp_consts *K, int slots, double delta, double lambda, const double *from, const double *to,
                                      size_t E, int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters,
                                      int check_target, int nblocks, unsigned long long *queue, const unsigned int *order,
                                      const double *carry_in, double *carry_out, int round_budget, hipStream_t st)
{
  if ((slots != 2 && slots != 4) || !queue) return hipErrorInvalidValue;
#define CCMP_LAUNCH_GEO_DENSE(STOCK, S)                                                                                                  \
  hipLaunchKernelGGL((geodesic_dense_kernel<STOCK, S>), dim3(nblocks), dim3(128 * S), 0, st, *K, delta, lambda, from, to, (unsigned long long)E, \
                     max_states, states, n_states, ok, newton_iters, check_target, queue, order, carry_in, carry_out, round_budget)
  if (K->stock) {
    if (slots == 4) CCMP_LAUNCH_GEO_DENSE(true, 4);
    else CCMP_LAUNCH_GEO_DENSE(true, 2);
  } else {
    if (slots == 4) CCMP_LAUNCH_GEO_DENSE(false, 4);
    else CCMP_LAUNCH_GEO_DENSE(false, 2);
  }
#undef CCMP_LAUNCH_GEO_DENSE
  return hipGetLastError();
}

// S = 2 or 4 slots per block; nblocks persistent blocks; queue_head: a zeroed 64-bit word
hipError_t ccmp_launch_project_dense(const ccmp_consts *K, int src, int slots, const double *q_in, double *q_out, uint8_t *ok,
                                     uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue_head,
                                     unsigned long long seed, unsigned long long first, const double *pool,
                                     const unsigned long long *pool_count, int wrap_output, int nblocks, size_t pool_records, hipStream_t st)
{
#define CCMP_LAUNCH_DENSE(SRC, STOCK, S)                                                                                          \
  hipLaunchKernelGGL((project_fd_dense_kernel<SRC, STOCK, S>), dim3(nblocks), dim3(128 * S), 0, st, *K, q_in, q_out, ok, iters,    \
                     q_ambient, (unsigned long long)B, queue_head, seed, first, pool, pool_count, wrap_output,                    \
                     (unsigned long long)pool_records)
#define CCMP_LAUNCH_DENSE_S(SRC, STOCK)        \
  do {                                         \
    if (slots == 4) CCMP_LAUNCH_DENSE(SRC, STOCK, 4); \
    else CCMP_LAUNCH_DENSE(SRC, STOCK, 2);     \
  } while (0)
  if (slots != 2 && slots != 4) return hipErrorInvalidValue;
  if (src == 0) {
    if (K->stock) CCMP_LAUNCH_DENSE_S(0, true);
    else CCMP_LAUNCH_DENSE_S(0, false);
  } else if (src == 1) {
    if (K->stock) CCMP_LAUNCH_DENSE_S(1, true);
    else CCMP_LAUNCH_DENSE_S(1, false);
  } else {
    if (K->stock) CCMP_LAUNCH_DENSE_S(2, true);
    else CCMP_LAUNCH_DENSE_S(2, false);
  }
#undef CCMP_LAUNCH_DENSE_S
#undef CCMP_LAUNCH_DENSE
  return hipGetLastError();
}

} // extern "C"
