// ccmp_kernels_resident.hip — the resident service kernel (ccmp_resident.h: what it is for, what keeps it from hanging anything).
// One persistent 128-thread block on the latency kernels' Newton routine (ccmp_flat_newton.h, built like the extend step's
// latency flavour: machine LICM on, 256-register budget — the block is alone on its CU's SIMDs, registers are free).
// Same arithmetic, same rounding model (-ffp-contract=off -DCCMP_USE_FMA): project / function / isSatisfied / jointValid and ONE edge
// of discreteGeodesic / checkMotion (the per-edge body of geodesic_flat_kernel, included: ccmp_geo_edge_body.inc) through it are
// bit-identical to the launched kernels (tests/test_gpu_resident.py).
#include "ccmp_flat_newton.h"
#include "ccmp_geo_edge.h"
#include "ccmp_resident.h"

namespace {

__device__ __forceinline__ unsigned long long sys_load(const unsigned long long *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(unsigned long long *p, unsigned long long v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long lane_word(unsigned long long v, int src_lane) // wave-uniform broadcast of one lane's word
{
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), src_lane);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), src_lane);
  return ((unsigned long long)hi << 32) | lo;
}

template <bool STOCK>
__global__ __launch_bounds__(128, 1) void resident_service_kernel(unsigned long long *box, unsigned long long last_tag,
                                                                 unsigned long long idle_ticks /* of the 100 MHz wall clock */)
{
  __shared__ __attribute__((aligned(16))) double lds[gRec]; // the Newton routine's record + an edge's previous state and target
  __shared__ double edge_ft[28];                            // an edge's `from` and `to`, as the request brought them
  __shared__ double s_par[4];                               // an edge's parameters: (max_states | round_budget), check_target, delta, lambda
  __shared__ double ktab[kConstsDoubles + 1];
  __shared__ __attribute__((aligned(16))) double steptab[kStepTab];
  __shared__ unsigned long long s_tag;
  __shared__ int s_cmd;
  __shared__ unsigned int s_consts;
  const int tid = threadIdx.x;
  const unsigned long long *consts = box + kResConstsOff / 8;
  const unsigned long long *req = box + kResReqOff / 8;
  unsigned long long *resp = box + kResRespOff / 8;
  unsigned long long *state = box + kResStateOff / 8;
  const ccmp_consts &KL = *reinterpret_cast<const ccmp_consts *>(ktab);
  double *rec = lds;
  unsigned int my_consts = ~0u; // nothing loaded yet: the first request brings its constants
  if (tid == 0) sys_store(state, (unsigned long long)kResRunning);
  unsigned long long idle_since = wall_clock64();

  for (;;) {
    // ---- wait for a request: the first 40 lanes read the five request lines, eight bytes each, over the link ---------------------
    bool fresh = false;
    if (tid < 64) { // wave 0
      unsigned long long w = 0;
      if (tid < kResReqWords) w = sys_load(req + tid);
      // A line's tag = the request's sequence number (low half) and a checksum of the line's seven payload words (high half):
      // whether this read returned each 64-byte line as ONE snapshot is the memory system's business — a line whose payload
      // words and tag are not of one request (read in pieces while the host was writing) fails its checksum and is polled again.
      const int col = tid & 7;
      unsigned long long h = 0;
      if (col < 7) { const int r = 7 * col + 1; h = (w << r) | (w >> (64 - r)); }
      h ^= __shfl_xor(h, 1);
      h ^= __shfl_xor(h, 2);
      h ^= __shfl_xor(h, 4); // every lane of a line holds the line's checksum
      const bool torn = col == 7 && tid < kResReqWords && (unsigned int)(w >> 32) != (unsigned int)(h ^ (h >> 32));
      const bool clean = __builtin_amdgcn_ballot_w64(torn) == 0ull;
      const unsigned int ta = (unsigned int)lane_word(w, 7), tb = (unsigned int)lane_word(w, 15), tc = (unsigned int)lane_word(w, 23),
                         td = (unsigned int)lane_word(w, 31), te = (unsigned int)lane_word(w, 39);
      fresh = clean && ta == tb && tb == tc && tc == td && td == te && te != (unsigned int)last_tag;
      if (fresh) {
        // the state sits in lanes 0..6 and 8..14 of this very read: straight into the Newton routine's record (and, for an edge,
        // `from` and `to` — lanes 16..22, 24..30 — into their staging)
        const double v = __longlong_as_double((long long)w);
        const int line = tid >> 3;
        if (col < 7) {
          if (line < 2) { rec[fX + 7 * line + col] = v; edge_ft[7 * line + col] = v; }
          else if (line < 4) edge_ft[14 + 7 * (line - 2) + col] = v;
          else if (line == 4 && col >= 1 && col <= 4) s_par[col - 1] = v; // words 33..36 (bit patterns; the integers are unpacked below)
        }
        if (tid == 0) {
          const unsigned long long head = lane_word(w, 32);
          s_cmd = (int)(head & 0xffffffffull);
          s_consts = (unsigned int)(head >> 32);
          s_tag = (unsigned long long)te;
        }
        last_tag = te;
      } else if (tid == 0) {
        s_cmd = (wall_clock64() - idle_since > idle_ticks) ? kResStop : kResNone; // nobody has asked for a while: leave by itself
      }
    }
    __syncthreads();
    const int cmd = s_cmd;
    if (cmd == kResNone) {
      __syncthreads(); // (s_cmd is rewritten at the top)
      continue;
    }
    if (cmd == kResStop) break;
    if (tid >= 64) last_tag = s_tag; // wave 1 follows (it never polls, but keeps the value coherent for clarity)
    // ---- the problem in force: its constants are reloaded when the host says they changed ----------------------------------------
    if (s_consts != my_consts) {
      __syncthreads();
      for (int k = tid; k < kConstsDoubles; k += 128) ktab[k] = __longlong_as_double((long long)sys_load(consts + k));
      __syncthreads();
      stage_step_table(KL, steptab, tid);
      my_consts = s_consts;
      __syncthreads();
    }
    // ---- the call ------------------------------------------------------------------------------------------------------------------
    int iter = 0, updates = 0;
    double norm1 = 0.0, norm2 = 0.0;
    bool ok = false;
    if (cmd == kResProject) {
      // KinematicChainConstraint::project (ConstraintFunction.h:57-82): project_fd_flat_kernel's sequence
      const bool conv = flat_newton<STOCK>(KL, KL, steptab, rec, tid, iter, updates, norm1, norm2, KL.max_iter);
      const bool jv = flat_joint_valid(KL, rec, tid);
      ok = jv && conv;
      if (tid < 14) sys_store(resp + kResRespQ + tid, (unsigned long long)__double_as_longlong(rec[fX + tid]));
    } else if (cmd == kResJointValid) {
      ok = flat_joint_valid(KL, rec, tid); // ConstraintFunction.h:43-55
    } else if (cmd == kResGeodesic) {
      // ONE edge of jy_ProjectedStateSpace::discreteGeodesic / checkMotion: geodesic_flat_kernel's per-edge body, the same text
      // (ccmp_geo_edge_body.inc), with its inputs and outputs pointed at the mailbox
      const ccmp_consts &K = KL;
      const double pi = 3.14159265358979323846;
      const unsigned long long pk = (unsigned long long)__double_as_longlong(s_par[0]);
      const int max_states = (int)(pk & 0xffffffffull), round_budget = (int)(pk >> 32);
      const int check_target = (int)(unsigned long long)__double_as_longlong(s_par[1]);
      const double delta = s_par[2], lambda = s_par[3];
      const unsigned long long t = 0;
      const double *from = edge_ft, *to = edge_ft + 14, *carry_in = nullptr, *ent = nullptr;
      double *states = reinterpret_cast<double *>(box + kResStatesOff / 8);
      int *n_states = reinterpret_cast<int *>(resp + kResRespN), *newton_iters = reinterpret_cast<int *>(resp + kResRespIts);
      uint8_t *ok_out = reinterpret_cast<uint8_t *>(resp + kResRespFlags);
      double *carry_out = reinterpret_cast<double *>(resp + kResRespCarry);
      if (tid == 0) sys_store(resp + kResRespFlags, 0ull); // (the body writes the flag as one byte)
      __syncthreads();
#include "ccmp_geo_edge_body.inc"
      __threadfence_system();
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(resp + kResRespDone, s_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        idle_since = wall_clock64();
      }
      __syncthreads();
      continue;
    } else {
      // function(x) through one evaluation pass of the Newton routine (iteration cap 0: no update), as checkMotion's isSatisfied(to)
      // in geodesic_flat_kernel; KinematicChainConstraint::isSatisfied's test on it (ConstraintFunction.h:114-120)
      (void)flat_newton<STOCK>(KL, KL, steptab, rec, tid, iter, updates, norm1, norm2, 0);
      const double f0 = rec[fF], f1 = rec[fF + 1];
      ok = (f0 - f0 == 0.0) && (f1 - f1 == 0.0) && f0 <= KL.tol_pos && f1 <= KL.tol_rot;
      if (tid < 2) sys_store(resp + kResRespF + tid, (unsigned long long)__double_as_longlong(tid ? f1 : f0));
      updates = 0;
    }
    if (tid == 0) sys_store(resp + kResRespFlags, (unsigned long long)(ok ? 1u : 0u) | ((unsigned long long)(unsigned int)updates << 32));
    // every wavefront's result words are on their way before the tag is: system fence, block barrier, then the tag
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(resp + kResRespDone, s_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      idle_since = wall_clock64();
    }
    __syncthreads();
  }
  if (tid == 0) {
    __threadfence_system();
    __hip_atomic_store(state, (unsigned long long)kResExited, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

} // namespace

extern "C" hipError_t ccmp_launch_resident(int stock, void *box_dev, unsigned long long last_tag, unsigned long long idle_ticks, hipStream_t st)
{
  if (stock)
    hipLaunchKernelGGL(resident_service_kernel<true>, dim3(1), dim3(128), 0, st, (unsigned long long *)box_dev, last_tag, idle_ticks);
  else
    hipLaunchKernelGGL(resident_service_kernel<false>, dim3(1), dim3(128), 0, st, (unsigned long long *)box_dev, last_tag, idle_ticks);
  return hipGetLastError();
}
