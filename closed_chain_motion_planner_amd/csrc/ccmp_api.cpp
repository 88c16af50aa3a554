// ccmp_api.cpp — host side of libccmp: execution context, scheduling, argument checking and kernel launches
// (problem set-up lives in ccmp_problem.cpp).
//
// There is deliberately no CPU implementation of the hot path in this library: project / function /
// isSatisfied batches run on the GPU or fail with CCMP_ENODEV / CCMP_EHIP.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/ccmp.h"
#include "ccmp_ctx.h"
#include "ccmp_host.h"
#include "ccmp_kin.h"

using ccmp_host::DeviceGuard;
using ccmp_host::g_hip_err;
using ccmp_host::hip_fail;
using ccmp_host::kPinData;

namespace ccmp_host {
thread_local char g_hip_err[256] = "";
}  // namespace ccmp_host

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
hipError_t ccmp_launch_geo_split2(const unsigned int *hist, int p_low, int p_high, int permille, unsigned long long *queue, hipStream_t st);
hipError_t ccmp_launch_geo_split(const unsigned int *hist, int p_min, int p_max, int permille, unsigned long long *queue, hipStream_t st);
hipError_t ccmp_launch_split_count(const unsigned int *hist, int pred_min, unsigned int limit, unsigned int *out, hipStream_t st);
hipError_t ccmp_launch_project_fast(const ccmp_consts *K, int mode, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue,
                                    unsigned long long seed, unsigned long long first, int lane_blocks, int rows_blocks,
                                    double *pool, int cap_iter, const unsigned int *order, const unsigned int *split_ptr,
                                    int front_blocks, hipStream_t side, hipEvent_t fork, hipEvent_t join, hipStream_t st);
hipError_t ccmp_launch_scout_order(const ccmp_consts *K, int mode, const double *q_in, size_t B, uint16_t *pred,
                                   unsigned int *hist, unsigned int *order, unsigned long long *queue,
                                   unsigned long long seed, unsigned long long first, int nblocks, int pair_max_blocks, hipStream_t st);
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
                                            hipStream_t st);
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

int projector_blocks(const ccmp_ctx *ctx, size_t B, int samples_per_wave, int default_wpc)
{
  const int wpc = ctx->waves_per_cu > 0 ? ctx->waves_per_cu : default_wpc;
  size_t want = (B + samples_per_wave - 1) / samples_per_wave;
  size_t cap = (size_t)ctx->num_cus * (size_t)wpc;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

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
  e = hipMalloc((void **)&ctx->queue, (kGeoGroupWords + 8) * sizeof(unsigned long long)); // 8 words: reference-arithmetic kernels; 64 + 3: analytic kernels; 1: split count
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
  *out = ctx;
  return CCMP_OK;
}

void ccmp_ctx_destroy(ccmp_ctx *ctx)
{
  if (!ctx) return;
  DeviceGuard guard(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->geo_pool) (void)hipFree(ctx->geo_pool);
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
  if (!strcmp(name, "flat_kernel")) { // latency work: 1 = one-round 128-thread kernel (default), 0 = single-wave kernel
    if (value != 0 && value != 1) return CCMP_EINVAL;
    ctx->flat_kernel = (int)value;
  } else if (!strcmp(name, "stock_kernels")) { // 1 = kernels specialised for the stock Panda structure when it applies (default)
    if (value != 0 && value != 1) return CCMP_EINVAL;
    ctx->stock_kernels = (int)value;
  } else if (!strcmp(name, "analytic_cap")) { // analytic mode: hand samples past this many iterations to the rows kernel (0 = never)
    if (value < 0 || value > 65535) return CCMP_EINVAL;
    ctx->analytic_cap = (int)value;
  } else if (!strcmp(name, "analytic_handover_max")) { // analytic mode: hand-over for batches up to this many samples
    if (value < 0) return CCMP_EINVAL;
    ctx->analytic_handover_max = (size_t)value;
  } else if (!strcmp(name, "analytic_split")) { // analytic mode, large batches: latency kernel beside the throughput kernel (0/1)
    if (value != 0 && value != 1) return CCMP_EINVAL;
    ctx->analytic_split = (int)value;
  } else if (!strcmp(name, "analytic_split_min")) {
    if (value < 0) return CCMP_EINVAL;
    ctx->analytic_split_min = (size_t)value;
  } else if (!strcmp(name, "analytic_split_max")) {
    if (value < 0) return CCMP_EINVAL;
    ctx->analytic_split_max = (size_t)value;
  } else if (!strcmp(name, "analytic_split_cap")) { // split launch: the one-lane kernel hands over past this many iterations
    if (value < 1 || value > 65535) return CCMP_EINVAL;
    ctx->analytic_split_cap = (int)value;
  } else if (!strcmp(name, "analytic_split_front")) { // wavefronts (= SIMDs) of the front kernel in a split launch
    if (value < 1 || value > 512) return CCMP_EINVAL;
    ctx->analytic_split_front = (int)value;
  } else if (!strcmp(name, "analytic_split_pred")) { // predicted iterations from which a sample goes to the front kernel
    if (value < 1 || value > 1023) return CCMP_EINVAL;
    ctx->analytic_split_pred = (int)value;
  } else if (!strcmp(name, "analytic_small_batch")) { // analytic mode: at or below this many samples the rows kernel alone
    if (value < 0) return CCMP_EINVAL;
    ctx->analytic_small_batch = (size_t)value;
  } else if (!strcmp(name, "clearance_per_state_max")) { // proxy clearance: one block per state up to this many states
    if (value < 0) return CCMP_EINVAL;
    ctx->clearance_per_state_max = (size_t)value;
  } else if (!strcmp(name, "host_zero_copy")) { // *_host calls on page-locked caller buffers: 0 staged, 1 q_out direct, 2 q_in too
    if (value < 0 || value > 2) return CCMP_EINVAL;
    ctx->host_zero_copy = (int)value;
  } else if (!strcmp(name, "scout_pairs")) { // FP32 scouts on lane pairs (one arm per lane) where lanes are plentiful: 1 = on
    if (value != 0 && value != 1) return CCMP_EINVAL;
    ctx->scout_pairs = (int)value;
  } else if (!strcmp(name, "scout_pair_blocks_per_cu")) { // ... projector: up to this many 256-thread blocks per CU
    if (value < 1 || value > 64) return CCMP_EINVAL;
    ctx->scout_pair_blocks_per_cu = (int)value;
  } else if (!strcmp(name, "scout_pair_max_edges")) { // ... extend step: up to this many edges
    if (value < 0) return CCMP_EINVAL;
    ctx->scout_pair_max_edges = (size_t)value;
  } else if (!strcmp(name, "latency_order_min")) { // latency kernel alone: FP32 scout order from this many samples on
    if (value < 0) return CCMP_EINVAL;
    ctx->latency_order_min = (size_t)value;
  } else if (!strcmp(name, "fd_split")) { // reference arithmetic, mid-size batches: the predicted-longest samples on latency blocks beside the throughput kernel
    if (value != 0 && value != 1) return CCMP_EINVAL;
    ctx->fd_split = (int)value;
  } else if (!strcmp(name, "fd_split_min")) {
    if (value < 0) return CCMP_EINVAL;
    ctx->fd_split_min = (size_t)value;
  } else if (!strcmp(name, "fd_split_max")) {
    if (value < 0) return CCMP_EINVAL;
    ctx->fd_split_max = (size_t)value;
  } else if (!strcmp(name, "fd_split_pred")) { // predicted iterations from which a sample belongs to the front
    if ((value < 1 && value != -1) || value > 1023) return CCMP_EINVAL; // -1: by the batch size (ccmp_ctx.h)
    ctx->fd_split_pred = (int)value;
  } else if (!strcmp(name, "fd_split_group_cut")) { // throughput wavefronts per CU given up for the front's blocks
    if (value < -1 || value > 8) return CCMP_EINVAL;
    ctx->fd_split_group_cut = (int)value;
  } else if (!strcmp(name, "fd_split_front")) { // latency blocks (= samples at most) of the front
    if (value < -1 || value > 4096) return CCMP_EINVAL;
    ctx->fd_split_front = (int)value;
  } else if (!strcmp(name, "geodesic_group")) { // bulk extend calls: short edges on the throughput layout (1 = on)
    if (value != 0 && value != 1) return CCMP_EINVAL;
    ctx->geodesic_group = (int)value;
  } else if (!strcmp(name, "geodesic_group_min")) {
    if (value < 0) return CCMP_EINVAL;
    ctx->geodesic_group_min = (size_t)value;
  } else if (!strcmp(name, "geodesic_group_pred")) { // predicted Newton rounds from which an edge goes to the latency blocks
    if ((value < 1 && value != -1) || value > 1023) return CCMP_EINVAL;
    ctx->geodesic_group_pred = (int)value;
  } else if (!strcmp(name, "geodesic_group_permille")) { // share of the predicted work the front must carry (0 = cut at geodesic_group_pred)
    if (value < 0 || value > 1000) return CCMP_EINVAL;
    ctx->geodesic_group_permille = (int)value;
  } else if (!strcmp(name, "geodesic_group_low_cut")) {
    if ((value < 1 && value != -1) || value > 64) return CCMP_EINVAL;
    ctx->geodesic_group_low_cut = (int)value;
  } else if (!strcmp(name, "geodesic_group_heavy_permille")) {
    if (value < 0 || value > 1001) return CCMP_EINVAL;
    ctx->geodesic_group_heavy_permille = (int)value;
  } else if (!strcmp(name, "geodesic_group_handover_pct")) {
    if (value < 0 || value > 100) return CCMP_EINVAL;
    ctx->geodesic_group_handover_pct = (int)value;
  } else if (!strcmp(name, "geodesic_group_front_per_cu")) {
    if ((value < 1 && value != -1) || value > 8) return CCMP_EINVAL;
    ctx->geodesic_group_front_per_cu = (int)value;
  } else if (!strcmp(name, "geodesic_group_waves_per_cu")) {
    if (value < 1 || value > 10) return CCMP_EINVAL;
    ctx->geodesic_group_waves_per_cu = (int)value;
  } else if (!strcmp(name, "fd_split_samples")) { // samples of the front (0 = as many as blocks)
    if (value < -1 || value > 0x7fffffffll) return CCMP_EINVAL;
    ctx->fd_split_samples = value;
  } else if (!strcmp(name, "latency_blocks_per_cu")) { // persistent blocks of the projector's latency kernel per CU (8 resident)
    if (value < 1 || value > 32) return CCMP_EINVAL;
    ctx->latency_blocks_per_cu = (int)value;
  } else if (!strcmp(name, "geodesic_blocks_per_cu")) { // persistent blocks of the extend step's latency flavour per CU (4 resident)
    if (value < 1 || value > 32) return CCMP_EINVAL;
    ctx->geodesic_blocks_per_cu = (int)value;
  } else if (!strcmp(name, "geodesic_flavour")) { // 0: by call shape, 1: throughput build always, 2: latency build always (both: same bits)
    if (value < 0 || value > 2) return CCMP_EINVAL;
    ctx->geodesic_flavour = (int)value;
  } else if (!strcmp(name, "pool_long_remaining")) { // hand-over: samples predicted to need this many more iterations go first (0 = one class)
    if (value < 0 || value > 1000) return CCMP_EINVAL;
    ctx->pool_long_remaining = (int)value;
  } else if (!strcmp(name, "geodesic_order")) { // extend step, batches beyond the resident blocks: 0 = index order, 1 = far-apart
    if (value < 0 || value > 2) return CCMP_EINVAL; // edges first, 2 = FP32 scout + longest-predicted-first (falls back to 1 below geodesic_scout_min)
    ctx->geodesic_order = (int)value;
  } else if (!strcmp(name, "geodesic_scout_min")) {
    if (value < 0) return CCMP_EINVAL;
    ctx->geodesic_scout_min = (size_t)value;
  } else if (!strcmp(name, "geodesic_scout_rounds")) { // the scout stops an edge after this many Newton rounds ("long")
    if (value < 1 || value > 1023) return CCMP_EINVAL;
    ctx->geodesic_scout_rounds = (int)value;
  } else if (!strcmp(name, "geodesic_order_min")) { // ... from this many edges on
    if (value < 0) return CCMP_EINVAL;
    ctx->geodesic_order_min = (size_t)value;
  } else if (!strcmp(name, "geodesic_long_steps")) { // ... an edge is long when |to - from| exceeds this many delta
    if (value < 0) return CCMP_EINVAL;
    ctx->geodesic_long_steps = (double)value;
  } else if (!strcmp(name, "handover_threshold")) { // -1 = automatic, 0..10 = hand a wave over once <= this many groups are busy
    if (value < -1 || value > 110) return CCMP_EINVAL; // 11..110: occupancy-driven, hand over below (value - 10) % of the group slots
    ctx->dump_threshold = (int)value;
  } else {
    return CCMP_EINVAL;
  }
  return CCMP_OK;
}
int ccmp_ctx_set_lpt(ccmp_ctx *ctx, int mode, size_t min_batch)
{
  if (!ctx || mode < 0 || mode > 2) return CCMP_EINVAL;
  ctx->lpt = mode;
  ctx->lpt_min_batch = min_batch == CCMP_DEFAULT ? kDefaultLptMinBatch : min_batch;
  return CCMP_OK;
}
int ccmp_ctx_debug_lpt_pred(ccmp_ctx *ctx, uint16_t *host_out, size_t B)
{
  if (!ctx || !host_out || !ctx->lpt_buf || B > ctx->lpt_cap) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
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
int ccmp_ctx_device(const ccmp_ctx *ctx) { return ctx ? ctx->device : -1; }
int ccmp_ctx_num_cus(const ccmp_ctx *ctx) { return ctx ? ctx->num_cus : 0; }

#define CCMP_PROLOGUE()                                        \
  if (!ctx) return CCMP_EINVAL;                                \
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

// ---- scheduling of a reference-arithmetic batch ---------------------------------------------------------------------
// Which kernels a batch of B samples runs on.  Pure function of the context's settings and B; never changes a result.
//   latency kernel alone            B <= small_batch (or schedule 2): one sample per block, lowest latency per sample
//   throughput kernel (+ hand-over) otherwise: 10 samples per wavefront from a queue; when the queue runs dry the samples
//                                   still in flight go to the latency kernel
//   scout + longest-first order     from lpt_min_batch on; from 120000 samples on without hand-over (see below)
struct FdPlan {
  int group_blocks = 0;    // persistent wavefronts of the throughput kernel; 0 = latency kernel alone
  bool handover = false;   // throughput kernel dumps its last samples to the pool, the latency kernel finishes them
  bool scout = false;      // FP32 scout pass + descending counting sort -> processing order
  int dump_threshold = 10; // a wave hands over once the queue is dry and at most this many of its 10 groups are busy
  int latency_blocks = 0;  // grid of the latency kernel (direct launch or hand-over)
  bool latency_static = false; // one block per sample, static striding: no queue word to reset
};

static FdPlan plan_fd_batch(const ccmp_ctx *ctx, size_t B, bool external_order)
{
  FdPlan pl;
  const int wpc = ctx->waves_per_cu > 0 ? ctx->waves_per_cu : 12;
  // latency kernels: the flat kernel runs 8 blocks of 128 threads per CU (16 waves), the one-wave kernel wpc waves
  const size_t lat_cap = ctx->flat_kernel ? (size_t)ctx->num_cus * (size_t)ctx->latency_blocks_per_cu : (size_t)ctx->num_cus * (size_t)wpc;
  const bool latency_only = ctx->wave_kernel == 2 || (ctx->wave_kernel == 1 && B <= ctx->small_batch);
  if (latency_only) {
    pl.latency_blocks = (int)(B < lat_cap ? B : lat_cap);
    pl.latency_static = ctx->flat_kernel && B <= lat_cap;
    return pl;
  }
  pl.group_blocks = projector_blocks(ctx, B, 10, 12);
  pl.handover = ctx->wave_kernel == 1;
  // the scout pays from lpt_min_batch samples on by its order alone, and earlier where the split launch (project_common) uses
  // its predictions to start the longest samples on latency blocks at once
  const bool split_range = ctx->fd_split && ctx->flat_kernel && ctx->wave_kernel == 1 && B >= ctx->fd_split_min && B <= ctx->fd_split_max;
  pl.scout = !external_order && ctx->lpt > 0 && (B >= ctx->lpt_min_batch || split_range) && B < 0xffffffffull;
  // Large ordered batches end on their shortest samples, and the scout is accurate there (tools/scout_tail.py: in the
  // last fill of a 262144-sample batch it predicts <= 21 iterations and the truth is <= 23): nothing worth handing
  // over is left (-3 % at 262144 Wine_Bottle without it, tools/time_lpt3.py).  An explicit threshold keeps hand-over.
  if (pl.scout && (ctx->lpt == 2 || (ctx->dump_threshold < 0 && B >= 120000))) pl.handover = false;
  // the samples still in flight go to the latency kernel (it iterates ~15x faster than a fully occupied throughput wave but
  // spends twice the SIMD-cycles per iteration): at once for large batches; for batches of about one fill of the
  // throughput kernel only when the samples in flight no longer fill 70 % of its group slots (value 80 = 10 + 70 %)
  pl.dump_threshold = ctx->dump_threshold >= 0 ? ctx->dump_threshold : (B < kOccupancyHandoverBelow ? kOccupancyHandoverValue : 10);
  if (pl.handover) {
    const size_t in_flight = (size_t)pl.group_blocks * 10;
    pl.latency_blocks = (int)(in_flight < lat_cap ? in_flight : lat_cap);
  }
  return pl;
}

// shape of the split launch's front for a batch of B samples (ccmp_ctx.h: fd_split*; an option that was set wins)
struct SplitShape { int pred, blocks, cut; unsigned int samples; };
static SplitShape split_shape(const ccmp_ctx *ctx, size_t B)
{
  const bool wide = B <= kSplitWideMax;
  SplitShape s;
  s.pred = ctx->fd_split_pred >= 0 ? ctx->fd_split_pred : (wide ? 40 : 56);
  s.blocks = ctx->fd_split_front >= 0 ? ctx->fd_split_front : ctx->num_cus * (wide ? 2 : 1);
  s.cut = ctx->fd_split_group_cut >= 0 ? ctx->fd_split_group_cut : (wide ? 3 : 2);
  const long long per_cu = wide ? 4 : (B < 40960 ? 3 : 4);
  const long long n = ctx->fd_split_samples > 0 ? ctx->fd_split_samples : (ctx->fd_split_samples == 0 ? s.blocks : per_cu * ctx->num_cus);
  s.samples = (unsigned int)(n < s.blocks ? s.blocks : n);
  return s;
}

// workspaces owned by the context; they grow outside any stream capture (the first call at a size is never captured)
static int ensure_pool(ccmp_ctx *ctx, size_t records)
{
  if (ctx->pool_cap >= records) return CCMP_OK;
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
  if (ctx->lpt_buf) (void)hipFree(ctx->lpt_buf);
  ctx->lpt_buf = nullptr;
  ctx->lpt_cap = 0;
  HIP_TRY(hipMalloc(&ctx->lpt_buf, ((B * 2 + 255) & ~(size_t)255) + 4096 + B * 4 + B)); // pred u16 | hist 1024 x u32 | order u32 | flags u8 (bulk checkMotion)
  ctx->lpt_cap = B;
  return CCMP_OK;
}

static int project_common(ccmp_ctx *ctx, const ccmp_problem *p, int mode, const double *q_in, double *q_out,
                          uint8_t *ok, uint16_t *iters, double *q_ambient, size_t B, uint64_t seed, uint64_t first,
                          void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q_out || !ok || (mode == 0 && !q_in)) return CCMP_EINVAL;
  if ((((uintptr_t)q_in) | ((uintptr_t)q_out)) & 15u) return CCMP_EINVAL; // rows are moved in 16-byte pieces
  if (p->jacobian_mode != CCMP_JAC_FD) {
    // Analytic mode.  One sample per lane (one wavefront per SIMD) is the throughput kernel; it runs a sample at ~7 us per
    // iteration whatever the occupancy, so its longest sample bounds the launch.  With twin stock arms the six-lanes-per-
    // sample rows kernel (~4 us per iteration, 2.5x the SIMD-cycles per sample-iteration) takes small batches alone and,
    // for mid-size batches, the samples that pass analytic_cap iterations.  In-process sweep (Wine_Bottle / stefan, ms;
    // one-lane alone | hand-over past 96 iterations | rows alone): 4096: 1.43 / 1.79 | 1.23 / 1.48 | 0.95 / 1.19;
    // 16384: 1.44 / 1.81 | 1.25 / 1.51 | 1.00 / 1.48; 65536: 1.91 / 1.96 | 1.63 / 1.65 | 1.73 / 2.38; 262144: 2.71 / 3.41 |
    // 2.56 / 3.43 | -; 1048576: 5.75 / 8.68 | 5.83 / 8.98 | -.  Other problems (calibrated arms, tilted bases): one-lane alone.
    const int lane_blocks = projector_blocks(ctx, B, 64, 4);
    // Large batches, twin stock arms: the FP32 scout orders the batch longest-predicted-first; the samples predicted past
    // analytic_split_pred iterations (at most the front kernel's resident capacity) run on the six-lane kernel on the side
    // stream WHILE the one-lane kernel takes the rest, longest first — the launch no longer ends on the serial chain of a
    // 250-iteration sample started late — and what the one-lane kernel still hands over is finished behind both.
    if (K.twin_arms && ctx->analytic_split && ctx->analytic_cap > 0 && ctx->lpt > 0 && !ctx->order && B >= ctx->analytic_split_min &&
        B <= ctx->analytic_split_max && B < 0xffffffffull) {
      int rc = ensure_lpt_buffers(ctx, B);
      if (rc == CCMP_OK) rc = ensure_pool(ctx, B);
      if (rc != CCMP_OK) return rc;
      char *base = (char *)ctx->lpt_buf;
      uint16_t *pred = (uint16_t *)base;
      unsigned int *hist = (unsigned int *)(base + ((ctx->lpt_cap * 2 + 255) & ~(size_t)255));
      unsigned int *ord = (unsigned int *)((char *)hist + 4096);
      unsigned int *split = (unsigned int *)(ctx->queue + 8 + 64 + 3);
      HIP_TRY(ccmp_launch_scout_order(&K, mode, q_in, B, pred, hist, ord, ctx->queue + 5, seed, first, ctx->num_cus,
                                    ctx->scout_pairs ? ctx->num_cus * ctx->scout_pair_blocks_per_cu : 0, st));
      // the front kernel gets analytic_split_front wavefronts, one SIMD each (a one-lane wave fills a SIMD's registers, so
      // the one-lane kernel is launched that many wavefronts short); ten samples per wavefront, one round
      const int front_blocks = ctx->analytic_split_front;
      HIP_TRY(ccmp_launch_split_count(hist, ctx->analytic_split_pred, (unsigned int)front_blocks * 10u, split, st));
      const int lanes = lane_blocks > ctx->num_cus * 4 - front_blocks ? ctx->num_cus * 4 - front_blocks : lane_blocks;
      HIP_TRY(ccmp_launch_project_fast(&K, mode, q_in, q_out, ok, iters, q_ambient, B, ctx->queue + 8, seed, first, lanes, ctx->num_cus * 8,
                                       ctx->pool, ctx->analytic_split_cap, ord, split, front_blocks, ctx->side, ctx->fork, ctx->join, st));
      return CCMP_OK;
    }
    if (!K.twin_arms || ctx->analytic_cap <= 0 || (B > ctx->analytic_handover_max && B > ctx->analytic_small_batch)) {
      HIP_TRY(ccmp_launch_project_fast(&K, mode, q_in, q_out, ok, iters, q_ambient, B, ctx->queue + 8, seed, first, lane_blocks, 0, nullptr,
                                       0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, st));
      return CCMP_OK;
    }
    const size_t rows_cap = (size_t)ctx->num_cus * 8; // waves of the rows kernel: two per SIMD (16.5 KB of LDS each)
    if (B <= ctx->analytic_small_batch) {
      const size_t want = (B + 9) / 10;
      HIP_TRY(ccmp_launch_project_fast(&K, mode, q_in, q_out, ok, iters, q_ambient, B, ctx->queue + 8, seed, first, 0,
                                       (int)(want < rows_cap ? want : rows_cap), nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, st));
      return CCMP_OK;
    }
    int rc = ensure_pool(ctx, B); // every sample may be handed over
    if (rc != CCMP_OK) return rc;
    HIP_TRY(ccmp_launch_project_fast(&K, mode, q_in, q_out, ok, iters, q_ambient, B, ctx->queue + 8, seed, first, lane_blocks, (int)rows_cap,
                                     ctx->pool, ctx->analytic_cap, nullptr, nullptr, 0, nullptr, nullptr, nullptr, st));
    return CCMP_OK;
  }
  const FdPlan pl = plan_fd_batch(ctx, B, ctx->order != nullptr);
  // queue[0]: sample queue of the throughput kernel; queue[1]: pool fill count; queue[2]: read head of the latency kernel
  unsigned long long *const q_group = ctx->queue, *const q_pool_count = ctx->queue + 1, *const q_latency = ctx->queue + 2;
  if (!pl.latency_static) HIP_TRY(ccmp_launch_clear_words(ctx->queue, 16, st)); // the eight 64-bit words of this path (6: pool count from the back)

  if (pl.group_blocks == 0) { // small batches and single states
    if (ctx->flat_kernel) {
      unsigned int *flag = arm_done_word(ctx, B);
      // Longest-predicted-first on the latency kernel alone (round 4): a batch of a few fills of its blocks ends on the serial
      // chain of whichever long sample the index order happened to start late (8 192 samples = 4 fills: a 250-round sample
      // starts anywhere in the first 0.9 ms and needs 0.8 ms alone); the FP32 scout's order starts them first.
      const unsigned int *lat_order = nullptr;
      if (!pl.latency_static && ctx->lpt > 0 && !ctx->order && B >= ctx->latency_order_min && B < 0xffffffffull) {
        int rc = ensure_lpt_buffers(ctx, B);
        if (rc != CCMP_OK) return rc;
        char *base = (char *)ctx->lpt_buf;
        unsigned int *hist = (unsigned int *)(base + ((ctx->lpt_cap * 2 + 255) & ~(size_t)255));
        unsigned int *ord = (unsigned int *)((char *)hist + 4096);
        HIP_TRY(ccmp_launch_scout_order(&K, mode, q_in, B, (uint16_t *)base, hist, ord, ctx->queue + 5, seed, first, ctx->num_cus,
                                    ctx->scout_pairs ? ctx->num_cus * ctx->scout_pair_blocks_per_cu : 0, st));
        lat_order = ord;
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
  bool fd_split = false;
  int group_blocks = pl.group_blocks;
  if (pl.handover) { // (workspace first: nothing of this call is in flight yet if an allocation fails)
    int rc = ensure_pool(ctx, (size_t)pl.group_blocks * 10);
    if (rc != CCMP_OK) return rc;
  }
  if (pl.scout) { // FP32 scout pass -> predicted iteration counts -> descending counting sort -> processing order
    int rc = ensure_lpt_buffers(ctx, B);
    if (rc != CCMP_OK) return rc;
    char *base = (char *)ctx->lpt_buf;
    uint16_t *pred = (uint16_t *)base;
    unsigned int *hist = (unsigned int *)(base + ((ctx->lpt_cap * 2 + 255) & ~(size_t)255));
    unsigned int *ord = (unsigned int *)((char *)hist + 4096);
    // one 256-thread block per CU, 4 samples per lane at 262144: more lanes only lengthen the per-wave maximum
    HIP_TRY(ccmp_launch_scout_order(&K, mode, q_in, B, pred, hist, ord, ctx->queue + 5, seed, first, ctx->num_cus,
                                    ctx->scout_pairs ? ctx->num_cus * ctx->scout_pair_blocks_per_cu : 0, st));
    order = ord;
    // Split launch (round 4).  A mid-size batch ends on the serial chain of its longest samples: they start first in the
    // throughput kernel, do ~22 iterations there at 26-62 us each, and only after the hand-over — a millisecond into the call —
    // go on at the latency kernel's pace (a 250-round sample: 125 rounds while that kernel is loaded, then 100 more alone =
    // 0.3 ms behind everybody else).  With the split the front of the order — the samples predicted >= fd_split_pred
    // iterations, at most fd_split_front — runs on latency blocks on the side stream FROM THE START, beside the throughput
    // kernel, which takes the rest of the order (two wavefronts per CU fewer: a latency block needs two SIMDs with a free
    // register slot, and the persistent throughput waves never leave theirs) and hands over as before.
    const SplitShape sh = split_shape(ctx, B);
    if (pl.handover && ctx->flat_kernel && ctx->fd_split && B >= ctx->fd_split_min && B <= ctx->fd_split_max && sh.blocks > 0) {
      fd_split = true;
      HIP_TRY(ccmp_launch_fd_split(hist, sh.pred, sh.samples, ctx->queue, st));
      HIP_TRY(hipEventRecord(ctx->fork, st));
      HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->fork, 0));
      HIP_TRY(ccmp_launch_project_flat(&K, mode, q_in, q_out, ok, iters, q_ambient, B, ctx->queue + 7, seed, first, ctx->pool, q_pool_count, mode,
                                       sh.blocks, nullptr, 0, 0, ord, ctx->queue + 4, ctx->side));
      HIP_TRY(hipEventRecord(ctx->join, ctx->side));
      const int room = ctx->num_cus * ((ctx->waves_per_cu > 0 ? ctx->waves_per_cu : 12) - sh.cut);
      if (group_blocks > room) group_blocks = room;
    }
  }
  // hand-over in two classes (scout's prediction minus the iterations done): the pool is filled from both ends and the
  // latency kernel takes the long samples first; only with the scout's predictions and the default latency kernel
  const uint16_t *pred = (pl.scout && pl.handover && ctx->flat_kernel && ctx->pool_long_remaining > 0) ? (const uint16_t *)ctx->lpt_buf : nullptr;
  const size_t pool_records = pred ? (size_t)group_blocks * 10 : 0;
  HIP_TRY(ccmp_launch_project_group(&K, mode, q_in, q_out, ok, iters, q_ambient, B, q_group, seed, first, group_blocks,
                                    pl.handover ? ctx->pool : nullptr, pl.dump_threshold, order, pred, ctx->pool_long_remaining, pool_records,
                                    st));
  if (pl.handover) { // the pool's fill count is read on the device: the latency kernel's surplus blocks exit at once
    if (ctx->flat_kernel)
      HIP_TRY(ccmp_launch_project_flat(&K, 2, q_in, q_out, ok, iters, q_ambient, B, q_latency, seed, first, ctx->pool, q_pool_count, mode,
                                       pl.latency_blocks, nullptr, 0, pool_records, nullptr, nullptr, st));
    else
      HIP_TRY(ccmp_launch_project_wave(&K, 2, q_in, q_out, ok, iters, q_ambient, B, q_latency, seed, first, ctx->pool, q_pool_count, mode,
                                       pl.latency_blocks, st));
  }
  if (fd_split) HIP_TRY(hipStreamWaitEvent(st, ctx->join, 0)); // the call is complete on `st` when the front is
  return CCMP_OK;
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
  if (p->jacobian_mode != CCMP_JAC_FD) return CCMP_EINVAL; // the extend step exists in reference arithmetic only
  if (carry_in && check_target) return CCMP_EINVAL;        // a continuation's target was tested by the call it continues
  if (round_budget > 0 && !carry_out) return CCMP_EINVAL;  // a suspended edge is useless without what its continuation needs
  // a resumable call needs room for one state besides `from`: with a one-entry list the first accepted state already reports
  // max_states + 1 with `from` as its last stored state, and a caller following the protocol would continue from `from` for ever
  if ((carry_in || carry_out || round_budget > 0) && max_states < 2) return CCMP_EINVAL;
  // One 128-thread block per edge.  Up to the resident capacity every edge has its block at once and
  // the hardware dispatcher is the queue.  Beyond it the blocks are persistent and take tickets from an atomic word,
  // handed out through a long-edges-first order when the batch is large enough for the ordering pass to pay: the
  // launch then ends on short edges (16384 near-neighbour edges, 16-state lists: 3.15 -> 2.1 ms).
  // Two builds of the kernel (ccmp_kernels_geo.hip): a call that bounds the rounds per edge is bound by the chip's turnover
  // of Newton rounds and takes the throughput flavour (8 blocks per CU); a call that ends on one edge's serial chain — no
  // round budget, or no more edges than the latency flavour has blocks — takes the latency flavour (4 blocks per CU,
  // fewer instructions per round).
  const size_t lat_resident = (size_t)ctx->num_cus * (size_t)ctx->geodesic_blocks_per_cu;
  const bool latency_flavour = ctx->geodesic_flavour == 2 || (ctx->geodesic_flavour == 0 && (round_budget == 0 || E <= lat_resident));
  const size_t resident = latency_flavour ? lat_resident : (size_t)ctx->num_cus * (size_t)ctx->latency_blocks_per_cu;
  size_t nb = E;
  unsigned long long *queue = nullptr;
  const unsigned int *order = nullptr;
  bool scouted = false; // `order` is the FP32 scout's longest-first order and its histogram is in place
  if (E > resident) {
    nb = resident;
    queue = ctx->queue + 3; // word 3: ticket; word 4: the two counters of the ordering pass
    HIP_TRY(ccmp_launch_clear_words(queue, 4, st));
    if (ctx->geodesic_order && E >= ctx->geodesic_order_min && E < 0xffffffffull) {
      int rc = ensure_lpt_buffers(ctx, E);
      if (rc != CCMP_OK) return rc;
      char *base = (char *)ctx->lpt_buf;
      unsigned int *hist = (unsigned int *)(base + ((ctx->lpt_cap * 2 + 255) & ~(size_t)255));
      unsigned int *ord = (unsigned int *)((char *)hist + 4096);
      if (ctx->geodesic_order == 2 && E >= ctx->geodesic_scout_min && !carry_in) {
        // FP32 scout of every edge (the traversal in single precision with the exact Jacobian, one edge per lane, rounds
        // capped) -> predicted Newton rounds -> descending counting sort: longest-predicted-first
        HIP_TRY(ccmp_launch_geodesic_scout_order(&K, from, to, E, p->delta, p->lambda, max_states, ctx->geodesic_scout_rounds,
                                                 (uint16_t *)base, hist, ord, ctx->scout_pairs && E <= ctx->scout_pair_max_edges, st));
        scouted = true;
      } else {
        HIP_TRY(ccmp_launch_geodesic_order(from, to, E, ctx->geodesic_long_steps * p->delta, (unsigned int *)(ctx->queue + 4), ord, st));
      }
      order = ord;
    }
  }
  // Bulk calls (round budget, thousands of edges, scout order): the SHORT edges run on the throughput layout — ten edges per
  // wavefront, geodesic_group_kernel, less than half the instructions per Newton round — and the front of the order, the edges
  // predicted to need geodesic_group_pred rounds or more, on this kernel's blocks on the side stream, both from the start.
  if (scouted && round_budget > 0 && !carry_in && ctx->geodesic_group && E >= ctx->geodesic_group_min) {
    // (workspace first: nothing of this call is in flight yet if the allocation fails)
    size_t waves = (E + 9) / 10;
    const size_t cap = (size_t)ctx->num_cus * (size_t)ctx->geodesic_group_waves_per_cu;
    if (waves > cap) waves = cap;
    const int pct = ctx->geodesic_group_handover_pct;
    if (pct > 0 && ctx->geo_pool_cap < waves * 10) { // (grows outside any stream capture: the first call at a size is never captured)
      if (ctx->geo_pool) (void)hipFree(ctx->geo_pool);
      ctx->geo_pool = nullptr;
      ctx->geo_pool_cap = 0;
      HIP_TRY(hipMalloc((void **)&ctx->geo_pool, waves * 10 * kGeoPoolDoubles * sizeof(double)));
      ctx->geo_pool_cap = waves * 10;
    }
    unsigned long long *gq = ctx->queue + kGeoGroupWords; // [0] group kernel's ticket (starts behind the front), [4] front length, [5] front's ticket
    unsigned int *hist = (unsigned int *)((char *)ctx->lpt_buf + ((ctx->lpt_cap * 2 + 255) & ~(size_t)255));
    HIP_TRY(ccmp_launch_clear_words(gq, 16, st));
    // checkMotion: isSatisfied(to) of every edge up front (one lane per edge) for the group kernel; the front's blocks test their own
    uint8_t *target_ok = nullptr;
    if (check_target) {
      target_ok = (uint8_t *)hist + 4096 + ctx->lpt_cap * 4;
      HIP_TRY(ccmp_launch_is_satisfied(&K, to, target_ok, E, nullptr, 0, st));
    }
    // the cut of the order: by default one of two, by what the batch looks like (geo_split2_kernel) — at the scout's cap where the edges
    // beyond it carry a tenth of the predicted work (stefan, dumbbell), lower where they do not (Wine_Bottle)
    const int low_cut = ctx->geodesic_group_low_cut > 0 ? ctx->geodesic_group_low_cut : (E < kGeoGroupHighCut ? 40 : 48);
    const int front_per_cu = ctx->geodesic_group_front_per_cu > 0 ? ctx->geodesic_group_front_per_cu : 8;
    if (ctx->geodesic_group_permille > 0)
      HIP_TRY(ccmp_launch_geo_split(hist, 8, ctx->geodesic_group_pred > 0 ? ctx->geodesic_group_pred : 64, ctx->geodesic_group_permille, gq, st));
    else if (ctx->geodesic_group_pred <= 0)
      HIP_TRY(ccmp_launch_geo_split2(hist, low_cut, 64, ctx->geodesic_group_heavy_permille, gq, st));
    else
      HIP_TRY(ccmp_launch_fd_split(hist, ctx->geodesic_group_pred, 0xffffffffu, gq, st));
    HIP_TRY(hipEventRecord(ctx->fork, st));
    HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->fork, 0));
    HIP_TRY(ccmp_launch_geodesic(&K, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, check_target,
                                 ctx->num_cus * front_per_cu, gq + 5, order, carry_in, carry_out, round_budget, gq + 4, nullptr, nullptr, ctx->side));
    HIP_TRY(hipEventRecord(ctx->join, ctx->side));
    HIP_TRY(ccmp_launch_geodesic_group(&K, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, (int)waves, gq, order,
                                       carry_out, round_budget, pct > 0 ? ctx->geo_pool : nullptr, gq + 6, pct, target_ok, st));
    if (pct > 0) { // the handed-over edges: latency blocks behind the group kernel; the pool's fill count is read on the device
      const size_t lat = (size_t)ctx->num_cus * (size_t)ctx->latency_blocks_per_cu;
      HIP_TRY(ccmp_launch_geodesic(&K, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, 0,
                                   (int)(waves * 10 < lat ? waves * 10 : lat), gq + 7, nullptr, nullptr, carry_out, round_budget, nullptr, ctx->geo_pool,
                                   gq + 6, st));
    }
    HIP_TRY(hipStreamWaitEvent(st, ctx->join, 0));
    return CCMP_OK;
  }
  HIP_TRY((latency_flavour ? ccmp_launch_geodesic_lat : ccmp_launch_geodesic)(&K, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok,
                                                                             newton_iters, check_target, (int)nb, queue, order, carry_in, carry_out,
                                                                             round_budget, nullptr, nullptr, nullptr, st));
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
    if (ctx->scan) (void)hipFree(ctx->scan);
    ctx->scan = nullptr;
    ctx->scan_cap = 0;
    HIP_TRY(hipMalloc((void **)&ctx->scan, nblocks * sizeof(unsigned int)));
    ctx->scan_cap = nblocks;
  }
  HIP_TRY(ccmp_launch_compact(q, ok, B, q_valid, capacity, ctx->scan, (unsigned long long *)count_dev, st));
  return CCMP_OK;
}

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

} // extern "C"
