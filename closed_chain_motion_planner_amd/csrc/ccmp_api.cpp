// ccmp_api.cpp — host side of libccmp: execution context, scheduling, argument checking and kernel launches
// (problem set-up lives in ccmp_problem.cpp).
//
// There is deliberately no CPU implementation of the hot path in this library: project / function /
// isSatisfied batches run on the GPU or fail with CCMP_ENODEV / CCMP_EHIP.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <set>

#include "../../include/ccmp.h"
#include "ccmp_ctx.h"
#include "ccmp_host.h"
#include "ccmp_policy.h"
#include "ccmp_resident.h"
#include "ccmp_split.h"
#include "ccmp_kin.h"

using ccmp_host::AnalyticPlan;
using ccmp_host::DeviceGuard;
using ccmp_host::FdPlan;
using ccmp_host::GeoPlan;
using ccmp_host::g_hip_err;
using ccmp_host::hip_fail;
using ccmp_host::kPinData;

namespace ccmp_host {
thread_local char g_hip_err[256] = "";
}  // namespace ccmp_host

extern "C" int ccmp_policy_set_option(ccmp_ctx *ctx, const char *name, long value); // ccmp_policy.cpp: the option table

extern "C" {
hipError_t ccmp_launch_clear_words(void *words, size_t n_u32, hipStream_t st);
hipError_t ccmp_launch_project_group(const ccmp_consts *K, int mode, const double *q_in, double *q_out, uint8_t *ok,
                                     uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue,
                                     unsigned long long seed, unsigned long long first, int nblocks, double *pool,
                                     int dump_threshold, const unsigned int *order, const uint16_t *pred, int long_remaining,
                                     size_t pool_records, hipStream_t st);
hipError_t ccmp_launch_project_wave(const ccmp_consts *K, int src, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue_head,
                                    unsigned long long seed, unsigned long long first, const double *pool,
                                    const unsigned long long *pool_count, int wrap_output, int nblocks, hipStream_t st);
hipError_t ccmp_launch_project_flat(const ccmp_consts *K, int src, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue_head,
                                    unsigned long long seed, unsigned long long first, const double *pool,
                                    const unsigned long long *pool_count, int wrap_output, int nblocks, unsigned int *done_flag,
                                    unsigned int done_seq, size_t pool_records, const unsigned int *order,
                                    const unsigned long long *total_ptr, hipStream_t st);
hipError_t ccmp_launch_fd_split(const unsigned int *hist, int pred_min, unsigned int limit, unsigned long long *queue, hipStream_t st);
hipError_t ccmp_launch_geo_split(const unsigned int *hist, int p_min, int p_max, int permille, unsigned long long *queue, hipStream_t st);
hipError_t ccmp_launch_split_count(const unsigned int *hist, int pred_min, unsigned int limit, unsigned int *out, hipStream_t st);
hipError_t ccmp_launch_geodesic_analytic_step(int which, double delta, double lambda, const double *from, const double *to, size_t E,
                                              int max_states, double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters,
                                              const double *carry_in, double *carry_out, const uint8_t *target_ok, void *const *ws,
                                              const double *start14, hipStream_t st);
hipError_t ccmp_launch_project_analytic(const ccmp_consts *K, int mode, const double *q_in, double *q_out, uint8_t *ok, uint16_t *iters,
                                        double *q_ambient, size_t B, unsigned long long *queue, unsigned long long seed,
                                        unsigned long long first, int pair_blocks, int dump_below, int latency_blocks, double *pool,
                                        hipStream_t st);
hipError_t ccmp_launch_scout_order(const ccmp_consts *K, int mode, const double *q_in, size_t B, uint16_t *pred,
                                   unsigned int *hist, unsigned int *order, unsigned long long *queue,
                                   unsigned long long seed, unsigned long long first, int nblocks, int pair_max_blocks,
                                   const ccmp_split_req *split, hipStream_t st);
hipError_t ccmp_launch_function(const ccmp_consts *K, const double *q, double *f, size_t B, unsigned int *done_flag,
                                unsigned int done_seq, hipStream_t st);
hipError_t ccmp_launch_is_satisfied(const ccmp_consts *K, const double *q, uint8_t *ok, size_t B, unsigned int *done_flag,
                                    unsigned int done_seq, hipStream_t st);
hipError_t ccmp_launch_joint_valid(const ccmp_consts *K, const double *q, uint8_t *ok, size_t B, unsigned int *done_flag,
                                   unsigned int done_seq, hipStream_t st);
hipError_t ccmp_launch_ambient_uniform(const ccmp_consts *K, unsigned long long seed, unsigned long long first,
                                       double *q, size_t B, hipStream_t st);
hipError_t ccmp_launch_enforce_bounds(double *q, size_t B, hipStream_t st);
hipError_t ccmp_launch_ambient_ref(const ccmp_consts *K, int kind, unsigned long long seed, unsigned long long first,
                                   const double *ref, int ref_stride, double param, double *q, size_t B, hipStream_t st);
hipError_t ccmp_launch_t_wo(const ccmp_consts *K, const double *q, int q_stride, double *out, size_t B, hipStream_t st);
hipError_t ccmp_launch_geodesic(const ccmp_consts *K, double delta, double lambda, const double *from, const double *to,
                                size_t E, int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters,
                                int check_target, int nblocks, unsigned long long *queue, const unsigned int *order,
                                const double *carry_in, double *carry_out, int round_budget, const unsigned long long *total_ptr,
                                const double *pool, const unsigned long long *pool_count, hipStream_t st);
hipError_t ccmp_launch_geodesic_lat(const ccmp_consts *K, double delta, double lambda, const double *from, const double *to,
                                    size_t E, int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters,
                                    int check_target, int nblocks, unsigned long long *queue, const unsigned int *order,
                                    const double *carry_in, double *carry_out, int round_budget, const unsigned long long *total_ptr,
                                    const double *pool, const unsigned long long *pool_count, hipStream_t st);
hipError_t ccmp_launch_geodesic_group(const ccmp_consts *K, double delta, double lambda, const double *from, const double *to, size_t E,
                                      int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters, int nblocks,
                                      unsigned long long *queue, const unsigned int *order, double *carry_out, int round_budget,
                                      double *pool, unsigned long long *pool_count, int handover_pct, const uint8_t *target_ok, hipStream_t st);
hipError_t ccmp_launch_geodesic_order(const double *from, const double *to, size_t E, double long_dist, unsigned int *counters,
                                      unsigned int *order, hipStream_t st);
hipError_t ccmp_launch_geodesic_scout_order(const ccmp_consts *K, const double *from, const double *to, size_t E, double delta, double lambda,
                                            int max_states, int round_cap, uint16_t *pred, unsigned int *hist, unsigned int *order, int pairs,
                                            const ccmp_split_req *split, hipStream_t st);
hipError_t ccmp_launch_detmath_probe(const double *x, const double *y, double *out, size_t n, hipStream_t st);
hipError_t ccmp_launch_compact(const double *q, const uint8_t *ok, size_t B, double *out, size_t capacity,
                               unsigned int *block_counts, unsigned long long *total, hipStream_t st);
}



namespace {




// single-state calls from the host entry points: the kernel publishes ctx->done_seq in the pinned block behind its
// result and HostIO::finish polls that word (NULL = this launch does not publish)
unsigned int *arm_done_word(ccmp_ctx *ctx, size_t B)
{
  if (!ctx->want_done || B != 1 || !ctx->pin_dev) return nullptr;
  ctx->done_seq++;
  ctx->done_armed = true;
  return (unsigned int *)((char *)ctx->pin_dev + kPinData);
}

// A launch sequence that puts part of a call on the context's side stream.  fork() orders the side stream behind what the
// caller's stream holds; step() launches and remembers the FIRST failure instead of returning (later steps are skipped);
// side_done() marks the end of the side stream's part; join() ALWAYS orders the caller's stream behind the side stream — also
// when a step failed: the kernels already queued there write the caller's buffers and the context's queue words — and returns
// the call's result.
struct ForkJoin {
  ccmp_ctx *ctx;
  hipStream_t st;
  hipError_t err = hipSuccess;
  const char *what = nullptr;
  bool forked = false, recorded = false;
  ForkJoin(ccmp_ctx *c, hipStream_t s) : ctx(c), st(s) {}
  void fail(hipError_t e, const char *w) { if (err == hipSuccess && e != hipSuccess) { err = e; what = w; } }
  bool step(hipError_t e, const char *w) { fail(e, w); return err == hipSuccess; } // (FJ_STEP: not evaluated behind a failure)
  void fork()
  {
    if (err != hipSuccess) return;
    if (!step(hipEventRecord(ctx->fork, st), "hipEventRecord(fork)")) return;
    if (step(hipStreamWaitEvent(ctx->side, ctx->fork, 0), "hipStreamWaitEvent(side, fork)")) forked = true;
  }
  void side_done()
  {
    if (!forked || recorded) return;
    const hipError_t e = hipEventRecord(ctx->join, ctx->side); // (also behind a failed step: what reached the side stream must be joined)
    if (e == hipSuccess) recorded = true;
    fail(e, "hipEventRecord(join)");
  }
  int join()
  {
    if (forked) {
      side_done();
      if (recorded) fail(hipStreamWaitEvent(st, ctx->join, 0), "hipStreamWaitEvent(st, join)");
      else (void)hipStreamSynchronize(ctx->side); // the join event could not be recorded: the only ordering left is the host's
    }
    return err == hipSuccess ? CCMP_OK : hip_fail(err, what);
  }
};

#define FJ_STEP(fj, call)                                  \
  do {                                                     \
    if ((fj).err == hipSuccess) (fj).fail((call), #call);  \
  } while (0)

int check_problem(const ccmp_problem *p)
{
  if (!p) return CCMP_EINVAL;
  if (!(p->tol_pos > 0) || !(p->tol_rot > 0)) return CCMP_EINVAL;
  if (p->max_iter < 0 || p->max_iter > 65535) return CCMP_EINVAL;
  if (p->jacobian_mode != CCMP_JAC_FD && p->jacobian_mode != CCMP_JAC_ANALYTIC) return CCMP_EINVAL;
  return CCMP_OK;
}

} // namespace

extern "C" {

int ccmp_version(void) { return CCMP_VERSION; }
size_t ccmp_problem_sizeof(void) { return sizeof(ccmp_problem); }
const char *ccmp_last_hip_error(void) { return g_hip_err; }

const char *ccmp_strerror(int code)
{
  switch (code) {
    case CCMP_OK: return "ok";
    case CCMP_EINVAL: return "invalid argument";
    case CCMP_EHIP: return "HIP runtime error";
    case CCMP_EIO: return "cannot read file";
    case CCMP_EPARSE: return "YAML key missing or malformed";
    case CCMP_ENODEV: return "no usable HIP device";
    case CCMP_ENOMEM: return "out of memory";
    case CCMP_ECOMM: return "RCCL error or librccl not loadable";
    case CCMP_EOVERFLOW: return "more valid states than the gather blocks hold";
    default: return "unknown error";
  }
}

// ---- context ---------------------------------------------------------------------------------------
// the contexts that exist (ccmp_host::context_alive): +1 registers, -1 removes, 0 asks
static bool live_contexts(const ccmp_ctx *ctx, int op)
{
  static std::mutex mu;
  static std::set<const ccmp_ctx *> live;
  std::lock_guard<std::mutex> hold(mu);
  if (op > 0) live.insert(ctx);
  else if (op < 0) live.erase(ctx);
  return live.count(ctx) != 0;
}

int ccmp_ctx_create(int device, ccmp_ctx **out)
{
  if (!out) return CCMP_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    snprintf(g_hip_err, sizeof g_hip_err, "hipGetDeviceCount: no HIP device");
    return CCMP_ENODEV;
  }
  if (device < 0 || device >= ndev) return CCMP_ENODEV;
  DeviceGuard guard(device);
  if (!guard.ok) return CCMP_ENODEV;
  ccmp_ctx *ctx = new (std::nothrow) ccmp_ctx();
  if (!ctx) return CCMP_ENOMEM;
  ctx->device = device;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { delete ctx; return hip_fail(e, "hipGetDeviceProperties"); }
  ctx->num_cus = prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete ctx; return hip_fail(e, "hipStreamCreate"); }
  e = hipMalloc((void **)&ctx->queue, (kGeoGroupWords + 8) * sizeof(unsigned long long)); // 8 words: reference-arithmetic kernels; kAnalyticWords: analytic kernel; 1: spare; 8: bulk extend
  if (e != hipSuccess) { (void)hipStreamDestroy(ctx->stream); delete ctx; return hip_fail(e, "hipMalloc(queue)"); }
  // side stream of the analytic mode's split launches (latency kernel beside the throughput kernel) and the two events
  // that order it against the caller's stream
  // The side stream must not share a hardware queue with the stream the caller launches on: HIP spreads streams over a
  // handful of hardware queues, and two streams on one queue run their kernels one after the other — seen with a second
  // context in the process, whose side stream landed on the NULL stream's queue: the split launch's front ran alone, in
  // front of the throughput kernel, and a 16 384-sample call took 2.34 ms instead of 1.68 (kernel trace).  Streams of another
  // PRIORITY get queues of their own, so the side stream takes the highest one (callers' streams are normal priority unless
  // they ask otherwise) — which also suits what runs there: the longest samples.
  {
    int prio_least = 0, prio_greatest = 0;
    e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (e == hipSuccess && prio_greatest != prio_least) e = hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, prio_greatest);
    else e = hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->join, hipEventDisableTiming);
  if (e != hipSuccess) { ccmp_ctx_destroy(ctx); return hip_fail(e, "side stream / events"); }
  live_contexts(ctx, +1);
  *out = ctx;
  return CCMP_OK;
}

void ccmp_ctx_destroy(ccmp_ctx *ctx)
{
  if (!ctx) return;
  live_contexts(ctx, -1);
  DeviceGuard guard(ctx->device);
  ccmp_host::resident_destroy(ctx); // first: hipFree below waits for the whole device, a resident kernel included
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->geo_pool) (void)hipFree(ctx->geo_pool);
  if (ctx->geo_an) (void)hipFree(ctx->geo_an);
  if (ctx->queue) (void)hipFree(ctx->queue);
  if (ctx->pool) (void)hipFree(ctx->pool);
  if (ctx->lpt_buf) (void)hipFree(ctx->lpt_buf);
  if (ctx->scan) (void)hipFree(ctx->scan);
  if (ctx->stage) (void)hipFree(ctx->stage);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
  if (ctx->fork) (void)hipEventDestroy(ctx->fork);
  if (ctx->join) (void)hipEventDestroy(ctx->join);
  if (ctx->ev_shard) (void)hipEventDestroy(ctx->ev_shard);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int ccmp_ctx_set_waves_per_cu(ccmp_ctx *ctx, int w)
{
  if (!ctx || w < 0 || w > 32) return CCMP_EINVAL;
  ctx->waves_per_cu = w;
  return CCMP_OK;
}
int ccmp_ctx_set_schedule(ccmp_ctx *ctx, int wave_kernel, size_t small_batch)
{
  if (!ctx || wave_kernel < 0 || wave_kernel > 2) return CCMP_EINVAL;
  ctx->wave_kernel = wave_kernel;
  ctx->small_batch = small_batch == CCMP_DEFAULT ? kDefaultSmallBatch : small_batch;
  return CCMP_OK;
}
int ccmp_ctx_set_option(ccmp_ctx *ctx, const char *name, long value)
{
  if (!ctx || !name) return CCMP_EINVAL;
  if (!strcmp(name, "resident")) return ccmp_host::resident_set(ctx, value); // starts / stops the service kernel (ccmp_resident.cpp)
  return ccmp_policy_set_option(ctx, name, value); // the option table: ccmp_policy.cpp
}
int ccmp_ctx_set_lpt(ccmp_ctx *ctx, int mode, size_t min_batch)
{
  if (!ctx || mode < 0 || mode > 2) return CCMP_EINVAL;
  ctx->lpt = mode;
  ctx->lpt_min_batch = min_batch == CCMP_DEFAULT ? kDefaultLptMinBatch : min_batch;
  return CCMP_OK;
}
#ifdef CCMP_DEBUG_HOOKS // include/ccmp_debug.h: lib/libccmp_debug.so only
int ccmp_ctx_debug_lpt_pred(ccmp_ctx *ctx, uint16_t *host_out, size_t B)
{
  if (!ctx || !host_out || !ctx->lpt_buf || B > ctx->lpt_cap) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  ccmp_host::quiesce(ctx);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host_out, ctx->lpt_buf, B * sizeof(uint16_t), hipMemcpyDeviceToHost));
  return CCMP_OK;
}
int ccmp_ctx_set_order_experimental(ccmp_ctx *ctx, const unsigned int *order_dev)
{
  if (!ctx) return CCMP_EINVAL;
  ctx->order = order_dev;
  return CCMP_OK;
}
int ccmp_debug_fail_calls(ccmp_ctx *ctx, int n)
{
  if (!ctx || n < 0) return CCMP_EINVAL;
  ctx->debug_fail_calls = n;
  return CCMP_OK;
}
#endif
int ccmp_ctx_device(const ccmp_ctx *ctx) { return ctx ? ctx->device : -1; }
int ccmp_ctx_num_cus(const ccmp_ctx *ctx) { return ctx ? ctx->num_cus : 0; }

#ifdef CCMP_DEBUG_HOOKS
#define CCMP_DEBUG_FAIL_POINT() if (ctx->debug_fail_calls > 0) { ctx->debug_fail_calls--; return CCMP_EHIP; }
#define CCMP_FAIL_AFTER_FORK(fj, where) if (ctx->fail_after_fork == (where)) (fj).fail(hipErrorLaunchFailure, "fail_after_fork (debug option)")
#else
#define CCMP_DEBUG_FAIL_POINT()
#define CCMP_FAIL_AFTER_FORK(fj, where)
#endif
#define CCMP_PROLOGUE()                                        \
  if (!ctx) return CCMP_EINVAL;                                \
  CCMP_DEBUG_FAIL_POINT()                                      \
  { int rc_ = check_problem(p); if (rc_ != CCMP_OK) return rc_; } \
  DeviceGuard guard(ctx->device);                              \
  if (!guard.ok) return CCMP_ENODEV;                           \
  hipStream_t st = (hipStream_t)hip_stream; \
  ccmp_consts K;                                               \
  ccmp_host::make_consts(*p, K);                               \
  if (!ctx->stock_kernels) K.stock = K.twin_arms = 0

int ccmp_function_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, double *f, size_t B, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q || !f) return CCMP_EINVAL;
  unsigned int *flag = arm_done_word(ctx, B);
  HIP_TRY(ccmp_launch_function(&K, q, f, B, flag, ctx->done_seq, st));
  return CCMP_OK;
}

// workspaces owned by the context; they grow outside any stream capture (the first call at a size is never captured)
static int ensure_pool(ccmp_ctx *ctx, size_t records)
{
  if (ctx->pool_cap >= records) return CCMP_OK;
  ccmp_host::quiesce(ctx); // (hipFree waits for the whole device: a resident service kernel must be gone first)
  if (ctx->pool) (void)hipFree(ctx->pool);
  ctx->pool = nullptr;
  ctx->pool_cap = 0;
  HIP_TRY(hipMalloc((void **)&ctx->pool, records * 18 * sizeof(double)));
  ctx->pool_cap = records;
  return CCMP_OK;
}
static int ensure_lpt_buffers(ccmp_ctx *ctx, size_t B)
{
  if (ctx->lpt_cap >= B) return CCMP_OK;
  ccmp_host::quiesce(ctx);
  if (ctx->lpt_buf) (void)hipFree(ctx->lpt_buf);
  ctx->lpt_buf = nullptr;
  ctx->lpt_cap = 0;
  HIP_TRY(hipMalloc(&ctx->lpt_buf, ((B * 2 + 255) & ~(size_t)255) + 4096 + B * 4 + B)); // pred u16 | hist 1024 x u32 | order u32 | flags u8 (bulk checkMotion)
  ctx->lpt_cap = B;
  return CCMP_OK;
}

// the scout's workspace inside ctx->lpt_buf: pred (u16 x cap) | hist (u32 x 1024) | order (u32 x cap) | flags (u8 x cap)
struct ScoutBuffers {
  uint16_t *pred;
  unsigned int *hist, *order;
  explicit ScoutBuffers(const ccmp_ctx *ctx)
  {
    char *base = (char *)ctx->lpt_buf;
    pred = (uint16_t *)base;
    hist = (unsigned int *)(base + ((ctx->lpt_cap * 2 + 255) & ~(size_t)255));
    order = (unsigned int *)((char *)hist + 4096);
  }
};

static int project_common(ccmp_ctx *ctx, const ccmp_problem *p, int mode, const double *q_in, double *q_out,
                          uint8_t *ok, uint16_t *iters, double *q_ambient, size_t B, uint64_t seed, uint64_t first,
                          void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q_out || !ok || (mode == 0 && !q_in)) return CCMP_EINVAL;
  if ((((uintptr_t)q_in) | ((uintptr_t)q_out)) & 15u) return CCMP_EINVAL; // rows are moved in 16-byte pieces
  const int scout_pair_blocks = ctx->scout_pairs ? ctx->num_cus * ctx->scout_pair_blocks_per_cu : 0;
  if (p->jacobian_mode != CCMP_JAC_FD) { // analytic mode: the plan is ccmp_policy.cpp's plan_analytic_batch
    const AnalyticPlan pl = ccmp_host::plan_analytic_batch(ctx, B);
    if (pl.pool_records > 0) {
      int rc = ensure_pool(ctx, pl.pool_records); // sized before anything of the call is in flight
      if (rc != CCMP_OK) return rc;
    }
    HIP_TRY(ccmp_launch_project_analytic(&K, mode, q_in, q_out, ok, iters, q_ambient, B, ctx->queue + 8, seed, first, pl.pair_blocks, pl.dump,
                                         pl.latency_blocks, ctx->pool, st));
    return CCMP_OK;
  }
  // Reference arithmetic.  sampleUniform on arms WITHOUT the stock structure (calibrated arms, tilted bases) is not fused: the
  // ambient sampler writes the states, the projector runs on them in place, enforceBounds wraps them — the same values through
  // 60 MB more traffic at C3 (0.03 ms beside a 20 ms kernel).  The fused general instantiation project_fd_kernel<1, false>
  // spilled 2.3 KB per lane (its sampler prologue and the general chain's pose arrays overlap) and was removed in round 6.
  if (mode == 1 && !(K.stock && K.twin_arms)) {
    double *amb = q_ambient ? q_ambient : q_out;
    HIP_TRY(ccmp_launch_ambient_uniform(&K, seed, first, amb, B, st));
    const int rc = project_common(ctx, p, 0, amb, q_out, ok, iters, nullptr, B, seed, first, hip_stream);
    if (rc != CCMP_OK) return rc;
    HIP_TRY(ccmp_launch_enforce_bounds(q_out, B, st));
    return CCMP_OK;
  }
  // the plan is ccmp_policy.cpp's plan_fd_batch (what ccmp_ctx_describe prints)
  const FdPlan pl = ccmp_host::plan_fd_batch(ctx, B, ctx->order != nullptr);
  // every workspace of the call is sized before anything of it is in flight
  if (pl.scout || pl.latency_order) {
    int rc = ensure_lpt_buffers(ctx, B);
    if (rc != CCMP_OK) return rc;
  }
  if (pl.handover) {
    int rc = ensure_pool(ctx, (size_t)pl.group_blocks * 10);
    if (rc != CCMP_OK) return rc;
  }
  // queue[0]: sample queue of the throughput kernel; queue[1]: pool fill count; queue[2]: read head of the latency kernel
  unsigned long long *const q_group = ctx->queue, *const q_pool_count = ctx->queue + 1, *const q_latency = ctx->queue + 2;
  if (!pl.latency_static) HIP_TRY(ccmp_launch_clear_words(ctx->queue, 16, st)); // the eight 64-bit words of this path (6: pool count from the back)

  if (pl.group_blocks == 0) { // small batches and single states
    if (ctx->flat_kernel) {
      unsigned int *flag = arm_done_word(ctx, B);
      const unsigned int *lat_order = nullptr;
      // (Round 6 measured and dropped a head start — the first resident-blocks samples in index order on this stream while scout
      // and sort of the rest, and then the rest, ran on the side stream: 4 096 Wine_Bottle samples 0.656 -> 0.704 ms, stefan 0.977 ->
      // 1.163 ms over six seeds, bit-identical; the longest sample is as likely in the rest, where it then starts later than behind
      // the scout — profiles/r06_head_start_ab.log.)
      if (pl.latency_order) { // longest-predicted-first on the latency kernel alone
        const ScoutBuffers sb(ctx);
        HIP_TRY(ccmp_launch_scout_order(&K, mode, q_in, B, sb.pred, sb.hist, sb.order, ctx->queue + 5, seed, first, ctx->num_cus, scout_pair_blocks, nullptr, st));
        lat_order = sb.order;
      }
      HIP_TRY(ccmp_launch_project_flat(&K, mode, q_in, q_out, ok, iters, q_ambient, B, pl.latency_static ? nullptr : q_latency, seed, first,
                                       ctx->pool, q_pool_count, mode, pl.latency_blocks, flag, ctx->done_seq, 0, lat_order, nullptr, st));
    }
    else
      HIP_TRY(ccmp_launch_project_wave(&K, mode, q_in, q_out, ok, iters, q_ambient, B, q_latency, seed, first, ctx->pool, q_pool_count,
                                       mode, pl.latency_blocks, st));
    return CCMP_OK;
  }

  const unsigned int *order = ctx->order;
  const uint16_t *pred = nullptr;
  ForkJoin fj(ctx, st);
  if (pl.scout) { // FP32 scout pass -> predicted iteration counts -> descending counting sort -> processing order
    const ScoutBuffers sb(ctx);
    // one 256-thread block per CU, 4 samples per lane at 262144: more lanes only lengthen the per-wave maximum
    // (a split launch's cut of the order — fd_split_kernel's rule — is decided by the sort's own kernel: ccmp_split.h)
    const ccmp_split_req cut{ctx->queue, 1, pl.shape.pred, 0, 0, pl.shape.samples, 0};
    HIP_TRY(ccmp_launch_scout_order(&K, mode, q_in, B, sb.pred, sb.hist, sb.order, ctx->queue + 5, seed, first, ctx->num_cus, scout_pair_blocks,
                                    pl.split ? &cut : nullptr, st));
    order = sb.order;
    // hand-over in two classes (scout's prediction minus the iterations done): the pool is filled from both ends and the
    // latency kernel takes the long samples first
    if (pl.two_class_pool) pred = sb.pred;
    // Split launch (round 4).  A mid-size batch ends on the serial chain of its longest samples: they start first in the
    // throughput kernel, do ~22 iterations there at 26-62 us each, and only after the hand-over — a millisecond into the call —
    // go on at the latency kernel's pace.  With the split the front of the order runs on latency blocks on the side stream FROM
    // THE START, beside the throughput kernel, which takes the rest of the order with a few wavefronts per CU fewer and hands
    // over as before.  From the fork on a failure is reported only after the side stream has been joined back (ForkJoin).
    if (pl.split) {
      fj.fork();
      CCMP_FAIL_AFTER_FORK(fj, 1);
      FJ_STEP(fj, ccmp_launch_project_flat(&K, mode, q_in, q_out, ok, iters, q_ambient, B, ctx->queue + 7, seed, first, ctx->pool, q_pool_count, mode,
                                           pl.shape.blocks, nullptr, 0, 0, sb.order, ctx->queue + 4, ctx->side));
      fj.side_done();
      CCMP_FAIL_AFTER_FORK(fj, 2);
    }
  }
  const size_t pool_records = pred ? (size_t)pl.group_blocks * 10 : 0;
  FJ_STEP(fj, ccmp_launch_project_group(&K, mode, q_in, q_out, ok, iters, q_ambient, B, q_group, seed, first, pl.group_blocks,
                                        pl.handover ? ctx->pool : nullptr, pl.dump_threshold, order, pred, ctx->pool_long_remaining, pool_records, st));
  if (pl.handover) { // the pool's fill count is read on the device: the latency kernel's surplus blocks exit at once
    if (ctx->flat_kernel)
      FJ_STEP(fj, ccmp_launch_project_flat(&K, 2, q_in, q_out, ok, iters, q_ambient, B, q_latency, seed, first, ctx->pool, q_pool_count, mode,
                                           pl.latency_blocks, nullptr, 0, pool_records, nullptr, nullptr, st));
    else
      FJ_STEP(fj, ccmp_launch_project_wave(&K, 2, q_in, q_out, ok, iters, q_ambient, B, q_latency, seed, first, ctx->pool, q_pool_count, mode,
                                           pl.latency_blocks, st));
  }
  return fj.join(); // the call is complete on `st` when the front is
}

int ccmp_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q_in, double *q_out, uint8_t *ok,
                       uint16_t *iters, size_t B, void *hip_stream)
{
  return project_common(ctx, p, 0, q_in, q_out, ok, iters, nullptr, B, 0, 0, hip_stream);
}

int ccmp_sample_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index, double *q_out,
                              uint8_t *ok, uint16_t *iters, double *q_ambient, size_t B, void *hip_stream)
{
  return project_common(ctx, p, 1, nullptr, q_out, ok, iters, q_ambient, B, seed, first_index, hip_stream);
}

static int sample_ref_common(ccmp_ctx *ctx, const ccmp_problem *p, int kind, uint64_t seed, uint64_t first_index,
                             const double *ref, int ref_stride, double param, double *q_out, uint8_t *ok, uint16_t *iters,
                             double *q_ambient, size_t B, void *hip_stream)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!ref || !q_out || !ok || (ref_stride != 0 && ref_stride != 14) || !(param >= 0)) return CCMP_EINVAL;
  {
    CCMP_PROLOGUE();
    HIP_TRY(ccmp_launch_ambient_ref(&K, kind, seed, first_index, ref, ref_stride, param, q_out, B, st));
    if (q_ambient) HIP_TRY(hipMemcpyAsync(q_ambient, q_out, B * 14 * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  int rc = ccmp_project_batch(ctx, p, q_out, q_out, ok, iters, B, hip_stream);
  if (rc != CCMP_OK) return rc;
  return ccmp_enforce_bounds_batch(ctx, q_out, B, hip_stream);
}

int ccmp_sample_near_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                                   const double *near, int near_stride, double distance, double *q_out, uint8_t *ok,
                                   uint16_t *iters, double *q_ambient, size_t B, void *hip_stream)
{
  return sample_ref_common(ctx, p, 0, seed, first_index, near, near_stride, distance, q_out, ok, iters, q_ambient, B, hip_stream);
}

int ccmp_sample_gaussian_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                                       const double *mean, int mean_stride, double std_dev, double *q_out, uint8_t *ok,
                                       uint16_t *iters, double *q_ambient, size_t B, void *hip_stream)
{
  return sample_ref_common(ctx, p, 1, seed, first_index, mean, mean_stride, std_dev, q_out, ok, iters, q_ambient, B, hip_stream);
}

int ccmp_compute_t_wo_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, int q_stride, double *t_wo, size_t B,
                            void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q || !t_wo || q_stride < 7) return CCMP_EINVAL;
  HIP_TRY(ccmp_launch_t_wo(&K, q, q_stride, t_wo, B, st));
  return CCMP_OK;
}

static int geodesic_common(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                           double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters, const double *carry_in,
                           double *carry_out, int round_budget, int check_target, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (E == 0) return CCMP_OK;
  if (!from || !to || !states || !n_states || !ok || max_states < 1 || round_budget < 0) return CCMP_EINVAL;
  if (!(p->delta > 0) || !(p->lambda > 0)) return CCMP_EINVAL;
  if (newton_iters == nullptr && p->jacobian_mode != CCMP_JAC_FD) return CCMP_EINVAL; // (the analytic step loop keeps its counts there)
  if (carry_in && check_target) return CCMP_EINVAL;        // a continuation's target was tested by the call it continues
  if (round_budget > 0 && !carry_out) return CCMP_EINVAL;  // a suspended edge is useless without what its continuation needs
  // a resumable call needs room for one state besides `from`: with a one-entry list the first accepted state already reports
  // max_states + 1 with `from` as its last stored state, and a caller following the protocol would continue from `from` for ever
  if ((carry_in || carry_out || round_budget > 0) && max_states < 2) return CCMP_EINVAL;
  if (p->jacobian_mode != CCMP_JAC_FD) {
    // Analytic mode (round 6): the traversal is a step loop around the batched analytic projector (ccmp_kernels_fast.hip) — per
    // step: every live edge's interpolated state, the projection of all of them in place, the reference's bookkeeping — at most
    // max_states steps, no host synchronisation, capturable.  A round budget is not enforced here (ok is never 2).
    if (ctx->geo_an_cap < E) { // (grows outside any stream capture: the first call at a size is never captured)
      ccmp_host::quiesce(ctx);
      if (ctx->geo_an) (void)hipFree(ctx->geo_an);
      ctx->geo_an = nullptr;
      ctx->geo_an_cap = 0;
      HIP_TRY(hipMalloc(&ctx->geo_an, E * ((14 + 14 + 3) * sizeof(double) + 2 + 1 + 1 + 1)));
      ctx->geo_an_cap = E;
    }
    char *base = (char *)ctx->geo_an;
    const size_t cap = ctx->geo_an_cap;
    double *prev = (double *)base, *scr = prev + cap * 14, *dtm = scr + cap * 14;
    uint16_t *itp = (uint16_t *)(dtm + cap * 3);
    uint8_t *okp = (uint8_t *)(itp + cap), *live = okp + cap, *tok = live + cap;
    void *ws[6] = {prev, scr, dtm, itp, okp, live};
    const uint8_t *target_ok = nullptr;
    if (check_target) { // ConstrainedMotionValidator::checkMotion: isSatisfied(to) first
      HIP_TRY(ccmp_launch_is_satisfied(&K, to, tok, E, nullptr, 0, st));
      target_ok = tok;
    }
    HIP_TRY(ccmp_launch_geodesic_analytic_step(0, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, carry_in, carry_out,
                                               target_ok, ws, p->start_joint, st));
    for (int s = 0; s < max_states; s++) {
      HIP_TRY(ccmp_launch_geodesic_analytic_step(1, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, nullptr, carry_out,
                                                 nullptr, ws, p->start_joint, st));
      const int rc = project_common(ctx, p, 0, scr, scr, okp, itp, nullptr, E, 0, 0, hip_stream);
      if (rc != CCMP_OK) return rc;
      HIP_TRY(ccmp_launch_geodesic_analytic_step(2, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, nullptr, carry_out,
                                                 nullptr, ws, p->start_joint, st));
    }
    return CCMP_OK;
  }
  // what the call is going to launch is decided (ccmp_policy.cpp: plan_geodesic — what ccmp_ctx_describe prints) and every workspace
  // it needs is sized before anything of it is in flight
  const GeoPlan pl = ccmp_host::plan_geodesic(ctx, E, round_budget, carry_in != nullptr);
  unsigned long long *queue = nullptr;
  const unsigned int *order = nullptr;
  if (pl.ordered) {
    int rc = ensure_lpt_buffers(ctx, E);
    if (rc != CCMP_OK) return rc;
  }
  if (pl.bulk && pl.handover_pct > 0 && ctx->geo_pool_cap < pl.group_waves * 10) { // (grows outside any stream capture: the first call at a size is never captured)
    ccmp_host::quiesce(ctx);
    if (ctx->geo_pool) (void)hipFree(ctx->geo_pool);
    ctx->geo_pool = nullptr;
    ctx->geo_pool_cap = 0;
    HIP_TRY(hipMalloc((void **)&ctx->geo_pool, pl.group_waves * 10 * kGeoPoolDoubles * sizeof(double)));
    ctx->geo_pool_cap = pl.group_waves * 10;
  }
  if (pl.queued) {
    queue = ctx->queue + 3; // word 3: ticket; word 4: the two counters of the ordering pass
    HIP_TRY(ccmp_launch_clear_words(queue, 4, st));
    if (pl.ordered) {
      const ScoutBuffers sb(ctx);
      if (pl.scouted) {
        // FP32 scout of every edge (the traversal in single precision with the exact Jacobian, one edge per lane, rounds
        // capped) -> predicted Newton rounds -> descending counting sort: longest-predicted-first
        // (a bulk call's default cut of the order — apply_split's kind 2, ccmp_kernels_scout.hip — is decided, and the launch's block of queue words
        // cleared, by the sort's own kernel: ccmp_split.h)
        const ccmp_split_req cut{ctx->queue + kGeoGroupWords, 2, pl.low_cut, 64, ctx->geodesic_group_heavy_permille, 0u, 8};
        HIP_TRY(ccmp_launch_geodesic_scout_order(&K, from, to, E, p->delta, p->lambda, max_states, ctx->geodesic_scout_rounds, sb.pred, sb.hist,
                                                 sb.order, pl.scout_pairs, pl.bulk && pl.default_cut ? &cut : nullptr, st));
      } else {
        HIP_TRY(ccmp_launch_geodesic_order(from, to, E, ctx->geodesic_long_steps * p->delta, (unsigned int *)(ctx->queue + 4), sb.order, st));
      }
      order = sb.order;
    }
  }
  // Bulk calls (round budget, thousands of edges, scout order): the SHORT edges run on the throughput layout — ten edges per
  // wavefront, geodesic_group_kernel, less than half the instructions per Newton round — and the front of the order on this
  // kernel's blocks on the side stream, both from the start.
  if (pl.bulk) {
    unsigned long long *gq = ctx->queue + kGeoGroupWords; // [0] group kernel's ticket (starts behind the front), [3] finished edges, [4] front length, [5] front's ticket, [6] pool count, [7] pool ticket
    const ScoutBuffers sb(ctx);
    const int pct = pl.handover_pct;
    if (!pl.default_cut) HIP_TRY(ccmp_launch_clear_words(gq, 16, st));
    // checkMotion: isSatisfied(to) of every edge up front (one lane per edge) for the group kernel; the front's blocks test their own
    uint8_t *target_ok = nullptr;
    if (check_target) {
      target_ok = (uint8_t *)sb.hist + 4096 + ctx->lpt_cap * 4;
      HIP_TRY(ccmp_launch_is_satisfied(&K, to, target_ok, E, nullptr, 0, st));
    }
    // the cut of the order: by default one of two, by what the batch looks like — at the scout's cap where the edges beyond it carry
    // a tenth of the predicted work (stefan, dumbbell), lower where they do not (Wine_Bottle) — decided by the sort's kernel above;
    // the options that fix it themselves get a launch of their own
    if (ctx->geodesic_group_permille > 0)
      HIP_TRY(ccmp_launch_geo_split(sb.hist, 8, ctx->geodesic_group_pred > 0 ? ctx->geodesic_group_pred : 64, ctx->geodesic_group_permille, gq, st));
    else if (ctx->geodesic_group_pred > 0)
      HIP_TRY(ccmp_launch_fd_split(sb.hist, ctx->geodesic_group_pred, 0xffffffffu, gq, st));
    // Fork.  From here on a failure no longer returns at once: whatever was queued on the side stream is joined back into the
    // caller's stream first (an early return left the side stream's kernels writing the caller's buffers unordered against
    // `st`), then the first error is reported.
    ForkJoin fj(ctx, st);
    fj.fork();
    CCMP_FAIL_AFTER_FORK(fj, 1);
    FJ_STEP(fj, ccmp_launch_geodesic(&K, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, check_target,
                                     pl.front_blocks, gq + 5, order, carry_in, carry_out, round_budget, gq + 4, nullptr, nullptr, ctx->side));
    fj.side_done();
    CCMP_FAIL_AFTER_FORK(fj, 2);
    FJ_STEP(fj, ccmp_launch_geodesic_group(&K, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, (int)pl.group_waves, gq,
                                           order, carry_out, round_budget, pct > 0 ? ctx->geo_pool : nullptr, gq + 6, pct, target_ok, st));
    if (pct > 0) // the handed-over edges: latency blocks behind the group kernel; the pool's fill count is read on the device
      FJ_STEP(fj, ccmp_launch_geodesic(&K, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, 0, pl.drain_blocks, gq + 7,
                                       nullptr, nullptr, carry_out, round_budget, nullptr, ctx->geo_pool, gq + 6, st));
    return fj.join();
  }
  HIP_TRY((pl.latency_flavour ? ccmp_launch_geodesic_lat : ccmp_launch_geodesic)(&K, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok,
                                                                                newton_iters, check_target, (int)pl.blocks, queue, order, carry_in,
                                                                                carry_out, round_budget, nullptr, nullptr, nullptr, st));
  return CCMP_OK;
}

int ccmp_geodesic_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                        double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters, void *hip_stream)
{
  return geodesic_common(ctx, p, from, to, E, max_states, states, n_states, ok, newton_iters, nullptr, nullptr, 0, 0, hip_stream);
}

int ccmp_check_motion_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                            double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters, void *hip_stream)
{
  return geodesic_common(ctx, p, from, to, E, max_states, states, n_states, ok, newton_iters, nullptr, nullptr, 0, 1, hip_stream);
}

int ccmp_geodesic_batch_ex(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                           double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters, const double *carry_in,
                           double *carry_out, int round_budget, int check_target, void *hip_stream)
{
  return geodesic_common(ctx, p, from, to, E, max_states, states, n_states, ok, newton_iters, carry_in, carry_out, round_budget,
                         check_target, hip_stream);
}

int ccmp_is_satisfied_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q || !ok) return CCMP_EINVAL;
  unsigned int *flag = arm_done_word(ctx, B);
  HIP_TRY(ccmp_launch_is_satisfied(&K, q, ok, B, flag, ctx->done_seq, st));
  return CCMP_OK;
}

int ccmp_joint_valid_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q || !ok) return CCMP_EINVAL;
  unsigned int *flag = arm_done_word(ctx, B);
  HIP_TRY(ccmp_launch_joint_valid(&K, q, ok, B, flag, ctx->done_seq, st));
  return CCMP_OK;
}

int ccmp_ambient_uniform_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index, double *q_out,
                               size_t B, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q_out) return CCMP_EINVAL;
  HIP_TRY(ccmp_launch_ambient_uniform(&K, seed, first_index, q_out, B, st));
  return CCMP_OK;
}

int ccmp_enforce_bounds_batch(ccmp_ctx *ctx, double *q, size_t B, void *hip_stream)
{
  if (!ctx) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(ccmp_launch_enforce_bounds(q, B, st));
  return CCMP_OK;
}

int ccmp_compact_valid(ccmp_ctx *ctx, const double *q, const uint8_t *ok, size_t B, double *q_valid, uint64_t *count_dev,
                       void *hip_stream)
{
  return ccmp_compact_valid_capped(ctx, q, ok, B, q_valid, B, count_dev, hip_stream);
}

int ccmp_compact_valid_capped(ccmp_ctx *ctx, const double *q, const uint8_t *ok, size_t B, double *q_valid, size_t capacity,
                              uint64_t *count_dev, void *hip_stream)
{
  if (!ctx || !count_dev) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  hipStream_t st = (hipStream_t)hip_stream;
  if (B == 0) {
    HIP_TRY(ccmp_launch_clear_words(count_dev, 2, st));
    return CCMP_OK;
  }
  if (!q || !ok || !q_valid) return CCMP_EINVAL;
  const size_t nblocks = (B + 255) / 256;
  if (ctx->scan_cap < nblocks) {
    // growth happens outside any capture: callers that capture graphs call once un-captured first
    ccmp_host::quiesce(ctx);
    if (ctx->scan) (void)hipFree(ctx->scan);
    ctx->scan = nullptr;
    ctx->scan_cap = 0;
    HIP_TRY(hipMalloc((void **)&ctx->scan, nblocks * sizeof(unsigned int)));
    ctx->scan_cap = nblocks;
  }
  HIP_TRY(ccmp_launch_compact(q, ok, B, q_valid, capacity, ctx->scan, (unsigned long long *)count_dev, st));
  return CCMP_OK;
}

#ifdef CCMP_DEBUG_HOOKS
int ccmp_detmath_probe(ccmp_ctx *ctx, const double *x_dev, const double *y_dev, double *out_dev, size_t n, void *hip_stream)
{
  if (!ctx || !x_dev || !y_dev || !out_dev) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  hipStream_t st = (hipStream_t)hip_stream;
  if (n == 0) return CCMP_OK;
  HIP_TRY(ccmp_launch_detmath_probe(x_dev, y_dev, out_dev, n, st));
  return CCMP_OK;
}
#endif

} // extern "C"

namespace ccmp_host {
bool context_alive(const ccmp_ctx *ctx) { return ctx && live_contexts(ctx, 0); }
}  // namespace ccmp_host
