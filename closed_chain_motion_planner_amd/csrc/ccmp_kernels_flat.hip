// ccmp_kernels_flat.hip — the latency kernel of the reference-arithmetic path: one sample per 128-thread block,
// every residual evaluation of a Newton iteration in ONE round (ccmp_flat_newton.h).  Built like ccmp_kernels_fd.hip
// with -mllvm -disable-machine-licm.
// __launch_bounds__(128, 4): four waves per SIMD, eight blocks per CU (116-118 VGPRs and no scratch for the stock-structure instantiations, 128 and 5-7 spilled dwords outside the Newton loop for the general ones; with a budget of 256
// registers the max-ilp scheduler takes 181 and halves the occupancy).
#include "ccmp_flat_newton.h"

namespace {

// SRC 0: q_in, SRC 1: ambient sampler, SRC 2: straggler pool.  queue == nullptr: static striding (one block per
// sample launches need no queue reset).
template <int SRC, bool STOCK>
__global__ __launch_bounds__(128, CCMP_FLAT_MIN_WAVES) void project_fd_flat_kernel(
    const ccmp_consts K, const double *__restrict__ q_in, double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
    uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient, unsigned long long B, unsigned long long *queue,
    unsigned long long seed, unsigned long long first_index, const double *__restrict__ pool,
    const unsigned long long *__restrict__ pool_count, int wrap_output, unsigned int *done_flag, unsigned int done_seq,
    unsigned long long pool_records, const unsigned int *__restrict__ order, const unsigned long long *__restrict__ total_ptr, int static_first)
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
  // the pool is filled from both ends (ccmp_kernels_fd.hip): pool_count[0] records predicted long at the front — taken
  // first — and pool_count[5] others from the back (pool_records == 0: front only)
  const unsigned long long n_front = (SRC == 2) ? pool_count[0] : 0ull;
  // order / total_ptr (SRC 0, 1; the split launch of mid-size batches, ccmp_api.cpp): the block's tickets run through the
  // first *total_ptr entries of a processing order — the samples the scout predicts longest — beside the throughput kernel
  const unsigned long long total = (SRC == 2) ? n_front + (pool_records ? pool_count[5] : 0ull) : (total_ptr ? *total_ptr : B);
  // static_first (launches that have the chip to themselves with no more blocks than are resident at once: the latency kernel
  // alone): the first ticket of every block is its own index — no atomic: a fetch-add on ONE word costs 12 ns chip-wide whoever
  // issues it (tools/ubench/atomic_rate.hip), so 2 048 blocks taking their first ticket from the queue word stood in line for up to
  // 25 us before the last of them had a sample; the queue word hands out the tickets behind the grid's.  NOT where part of the
  // grid may have to wait for room (beside a throughput kernel, behind a hand-over): a block that is not resident yet would hold
  // its ticket — one of the LONGEST samples of a longest-first order — until a resident block leaves, which is at the very end
  // (measured: bulk extend calls 5-6 % slower with it).
  unsigned long long t = blockIdx.x;
  bool first = static_first != 0;

  for (;;) {
    if (queue && !first) {
      if (tid == 0) ticket = (static_first ? (unsigned long long)gridDim.x : 0ull) + atomicAdd(queue, 1ull);
      __syncthreads();
      t = ticket;
    }
    first = false;
    if (t >= total) break;
    unsigned long long idx;
    int iter = 0, updates = 0;
    double norm1 = 0.0, norm2 = 0.0;
    if (SRC == 2) {
      const double *ent = pool + (t < n_front ? t : pool_records - 1ull - (t - n_front)) * kPoolEntry;
      idx = (unsigned long long)__double_as_longlong(ent[14]);
      iter = __double2hiint(ent[15]);
      updates = __double2loint(ent[15]);
      norm1 = ent[16];
      norm2 = ent[17];
      if (tid < 14) rec[fX + tid] = ent[tid];
    } else {
      idx = order ? (unsigned long long)order[t] : t;
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
#ifdef CCMP_GEO_TRACE
    const int updates_in = updates;
    if (tid == 0 && t < 65536) { g_geo_trace[3 * t] = wall_clock64(); g_geo_trace[3 * t + 2] = ((unsigned long long)blockIdx.x << 32) | (unsigned)updates_in; }
#endif
    const bool conv = flat_newton<STOCK>(K, KL, steptab, rec, tid, iter, updates, norm1, norm2, K.max_iter);
#ifdef CCMP_GEO_TRACE
    if (tid == 0 && t < 65536) { g_geo_trace[3 * t + 1] = wall_clock64(); g_geo_trace[3 * t + 2] |= (unsigned long long)(unsigned)(updates - updates_in) << 16; }
#endif
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

} // namespace

extern "C" {


// one sample per 128-thread block, all evaluations of an iteration in one round; queue_head may be NULL (static striding)
hipError_t ccmp_launch_project_flat(const ccmp_consts *K, int src, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue_head,
                                    unsigned long long seed, unsigned long long first, const double *pool,
                                    const unsigned long long *pool_count, int wrap_output, int nblocks, unsigned int *done_flag,
                                    unsigned int done_seq, size_t pool_records, const unsigned int *order,
                                    const unsigned long long *total_ptr, hipStream_t st)
{
  // the latency kernel alone (no pool, no front of a split launch): the whole grid is resident at once — static first tickets
  const int static_first = (src != 2 && total_ptr == nullptr) ? 1 : 0;
  if (nblocks != 1) done_flag = nullptr; // the completion word is written by the one block of a single-state call
#define CCMP_LAUNCH_FLAT(SRC, STOCK)                                                                                             \
  hipLaunchKernelGGL((project_fd_flat_kernel<SRC, STOCK>), dim3(nblocks), dim3(128), 0, st, *K, q_in, q_out, ok, iters, q_ambient, \
                     (unsigned long long)B, queue_head, seed, first, pool, pool_count, wrap_output, done_flag, done_seq, \
                     (unsigned long long)pool_records, order, total_ptr, static_first)
  if (src == 0) {
    if (K->stock) CCMP_LAUNCH_FLAT(0, true);
    else CCMP_LAUNCH_FLAT(0, false);
  } else if (src == 1) {
    if (!K->stock) return hipErrorInvalidValue; // the fused sampler exists for the stock structure only (ccmp_api.cpp: project_common)
    CCMP_LAUNCH_FLAT(1, true);
  } else {
    if (K->stock) CCMP_LAUNCH_FLAT(2, true);
    else CCMP_LAUNCH_FLAT(2, false);
  }
#undef CCMP_LAUNCH_FLAT
  return hipGetLastError();
}

#ifdef CCMP_GEO_TRACE
hipError_t ccmp_debug_flat_trace(unsigned long long *out, size_t n_edges)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_geo_trace), 3 * n_edges * sizeof(unsigned long long));
}
#endif

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
