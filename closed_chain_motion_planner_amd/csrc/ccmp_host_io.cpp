// ccmp_host_io.cpp — the host-pointer side of libccmp: the *_host conveniences (the reference's entry is host memory: an
// Eigen::Ref over the OMPL state's values), their staging / pinned block / page-locked caller buffers, and the sharded
// host calls that drive several GPUs from the planner's one process.  Everything here ends in the device-pointer *_batch
// entry points of ccmp_api.cpp; there is no CPU implementation of the path.
#include <hip/hip_runtime.h>

#include <chrono>
#include <thread>
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/ccmp.h"
#include "ccmp_ctx.h"
#include "ccmp_resident.h"

using ccmp_host::DeviceGuard;
using ccmp_host::ensure_stage;
using ccmp_host::g_hip_err;
using ccmp_host::hip_fail;

namespace ccmp_host {

int ensure_stage(ccmp_ctx *ctx, size_t bytes)
{
  if (ctx->stage_cap >= bytes) return CCMP_OK;
  quiesce(ctx); // (hipFree waits for the whole device: a resident service kernel must be gone first)
  if (ctx->stage) (void)hipFree(ctx->stage);
  ctx->stage = nullptr;
  ctx->stage_cap = 0;
  HIP_TRY(hipMalloc(&ctx->stage, bytes));
  ctx->stage_cap = bytes;
  return CCMP_OK;
}

void *pinned_alias(const void *host, size_t bytes)
{
  if (!host || bytes == 0) return nullptr;
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  // an unregistered (pageable) pointer is reported as an error by older runtimes and as hipMemoryTypeUnregistered by newer
  // ones; the error is sticky for hipGetLastError only and is cleared here
  if (hipPointerGetAttributes(&a, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
  // the mapping must be this device's: page-locked memory of another device's context is only usable here when it was
  // allocated portable, which the attributes do not tell — such a buffer takes the staged path
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || a.device != dev) return nullptr;
  // [host, host + bytes) must lie inside ONE page-locked allocation.  (Round 4 tested the first and the last byte only: with
  // hipHostRegister the device address equals the host address, so two separately registered regions with a pageable gap
  // between them passed, and the kernels then touched the gap — a GPU page fault instead of a fall-back.)
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a.devicePointer) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  const char *lo = (const char *)a.devicePointer, *b0 = (const char *)base;
  if (!base || lo < b0 || bytes > size || (size_t)(lo - b0) > size - bytes) return nullptr;
  return a.devicePointer;
}

int for_each_shard(int n, int (*fn)(int g, void *arg), void *arg)
{
  if (n <= 1) return n == 1 ? fn(0, arg) : CCMP_OK;
  struct Res { int rc = CCMP_OK; char err[sizeof g_hip_err] = ""; };
  std::vector<Res> res((size_t)n);
  std::vector<std::thread> th;
  th.reserve((size_t)n);
  for (int g = 0; g < n; g++) {
    try {
      th.emplace_back([g, fn, arg, &res] {
        res[(size_t)g].rc = fn(g, arg);
        if (res[(size_t)g].rc != CCMP_OK) memcpy(res[(size_t)g].err, g_hip_err, sizeof g_hip_err); // thread_local: hand it to the caller
      });
    } catch (...) { // no thread to be had: this shard runs here
      res[(size_t)g].rc = fn(g, arg);
      if (res[(size_t)g].rc != CCMP_OK) memcpy(res[(size_t)g].err, g_hip_err, sizeof g_hip_err);
    }
  }
  for (auto &t : th) t.join();
  for (int g = 0; g < n; g++)
    if (res[(size_t)g].rc != CCMP_OK) {
      memcpy(g_hip_err, res[(size_t)g].err, sizeof g_hip_err);
      return res[(size_t)g].rc;
    }
  return CCMP_OK;
}
}  // namespace ccmp_host

namespace {

// Host buffers of the *_host entry points.  Up to kPinBytes the kernels work directly on a pinned, device-mapped host
// block (a single project(x) is then memcpy + launch + synchronize + memcpy: no staged pageable copies, ~3 driver
// calls fewer); larger batches go through device staging with asynchronous copies.
using ccmp_host::kPinBytes;
using ccmp_host::kPinData;
struct HostIO {
  ccmp_ctx *ctx;
  char *dev = nullptr;  // what the kernels get
  char *host = nullptr; // pinned alias (small calls) or nullptr
  struct Out { void *dst; size_t off, n; } outs[4];
  int n_outs = 0;
  // every *_host call starts with the completion-word machinery off: a flag left over from a call that returned early
  // (an argument error behind want_done = true) must never make a later launch arm, or a later finish() poll, a word
  // that its own kernel does not publish last
  explicit HostIO(ccmp_ctx *c) : ctx(c) { ctx->want_done = ctx->done_armed = false; }
  int begin(size_t bytes)
  {
    if (bytes <= kPinData) {
      if (!ctx->pin) {
        ccmp_host::quiesce(ctx);
        HIP_TRY(hipHostMalloc(&ctx->pin, kPinBytes, hipHostMallocMapped | hipHostMallocCoherent));
        memset(ctx->pin, 0, kPinBytes);
        hipError_t e = hipHostGetDevicePointer(&ctx->pin_dev, ctx->pin, 0);
        if (e != hipSuccess) { (void)hipHostFree(ctx->pin); ctx->pin = nullptr; return hip_fail(e, "hipHostGetDevicePointer"); }
      }
      host = (char *)ctx->pin;
      dev = (char *)ctx->pin_dev;
      return CCMP_OK;
    }
    int rc = ensure_stage(ctx, bytes);
    dev = (char *)ctx->stage;
    return rc;
  }
  int in(size_t off, const void *src, size_t n)
  {
    if (host) { memcpy(host + off, src, n); return CCMP_OK; }
    HIP_TRY(hipMemcpyAsync(dev + off, src, n, hipMemcpyHostToDevice, ctx->stream));
    return CCMP_OK;
  }
  int out(void *dst, size_t off, size_t n)
  {
    if (host) { outs[n_outs++] = Out{dst, off, n}; return CCMP_OK; }
    HIP_TRY(hipMemcpyAsync(dst, dev + off, n, hipMemcpyDeviceToHost, ctx->stream));
    return CCMP_OK;
  }
  // a failure behind the launch: earlier download copies of this call may still be writing the caller's buffers — the call
  // returns only when its stream is quiet
  int abandon(int rc)
  {
    ctx->done_armed = ctx->want_done = false;
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  int finish()
  {
    bool done = false;
    if (host && ctx->done_armed) {
      // the one-block latency kernel wrote its results into this pinned block and then published done_seq: poll the
      // word (a store from the GPU is visible here ~1 us after it retires) instead of waiting for the completion
      // signal of the stream; bounded — a kernel that never publishes is caught by the synchronise below
      const volatile unsigned int *flag = (const volatile unsigned int *)(host + kPinData);
      for (long spin = 0; spin < 4000000; spin++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ctx->done_seq) { done = true; break; }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    }
    ctx->done_armed = false;
    ctx->want_done = false;
    if (!done) HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_outs; i++) memcpy(outs[i].dst, host + outs[i].off, outs[i].n);
    return CCMP_OK;
  }
};

} // namespace

extern "C" {

// ---- host-pointer conveniences ----------------------------------------------------------------------
// A batch whose q_in and q_out the caller keeps in PAGE-LOCKED memory (the reference's entry is host memory: an Eigen::Ref
// over the OMPL state's values, src/base/jy_ProjectedStateSpace.cpp:10-15; a planner that batches keeps its states in a
// pinned arena) is not staged like pageable memory: the kernels write every projected row straight into the caller's q_out
// (and, with host_zero_copy = 2, read q_in from it) — no 29 MB download behind the last kernel, no second copy of the batch
// on the device.  Only the flags and iteration counts (3 bytes per sample) go through device staging.  kNotPinned: the
// buffers are not both page-locked (or not 16-byte aligned): the caller falls back to the staged path.
static const int kNotPinned = 1;
static int project_host_pinned(ccmp_ctx *ctx, const ccmp_problem *p, const double *q_in, double *q_out, uint8_t *ok, uint16_t *iters,
                               size_t B)
{
  const size_t qb = B * 14 * sizeof(double);
  const double *din = (const double *)ccmp_host::pinned_alias(q_in, qb);
  double *dout = q_out == q_in ? (double *)din : (double *)ccmp_host::pinned_alias(q_out, qb);
  if (!din || !dout || ((uintptr_t)din & 15) || ((uintptr_t)dout & 15)) return kNotPinned;
  const bool copy_in = ctx->host_zero_copy < 2;
  const size_t off_ok = copy_in ? ((qb + 255) & ~(size_t)255) : 0;
  const size_t off_it = (off_ok + B + 255) & ~(size_t)255;
  int rc = ensure_stage(ctx, off_it + B * sizeof(uint16_t));
  if (rc != CCMP_OK) return rc;
  char *stage = (char *)ctx->stage;
  ctx->want_done = ctx->done_armed = false;
  const double *src = din;
  if (copy_in) { // one asynchronous copy at the link's rate; the scout and the projector then read device memory
    HIP_TRY(hipMemcpyAsync(stage, q_in, qb, hipMemcpyHostToDevice, ctx->stream));
    src = (const double *)stage;
  }
  rc = ccmp_project_batch(ctx, p, src, dout, (uint8_t *)(stage + off_ok), (uint16_t *)(stage + off_it), B, ctx->stream);
  if (rc == CCMP_OK) {
    hipError_t e = hipMemcpyAsync(ok, stage + off_ok, B, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && iters) e = hipMemcpyAsync(iters, stage + off_it, B * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpyAsync(D2H flags)");
  }
  hipError_t e = hipStreamSynchronize(ctx->stream); // also on the error path: the caller's buffers must be quiet
  if (e != hipSuccess && rc == CCMP_OK) rc = hip_fail(e, "hipStreamSynchronize");
  return rc;
}

int ccmp_project_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q_in, double *q_out, uint8_t *ok,
                      uint16_t *iters, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q_in || !q_out || !ok) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  if (B == 1 && ctx->resident_on) { // the resident service kernel (opt-in): no launch on the call path
    const int rc = ccmp_host::resident_call(ctx, p, ccmp_host::ResidentCall{kResProject, q_in, q_out, nullptr, ok, iters});
    if (rc != ccmp_host::kResidentFallBack) return rc;
  }
  const size_t qb = B * 14 * sizeof(double);
  if (qb > kPinData && ctx->host_zero_copy) {
    const int done = project_host_pinned(ctx, p, q_in, q_out, ok, iters, B);
    if (done != kNotPinned) return done;
  }
  const size_t off_ok = (qb + 255) & ~(size_t)255;
  const size_t off_it = (off_ok + B + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_it + B * sizeof(uint16_t));
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, q_in, qb)) != CCMP_OK) return rc;
  ctx->want_done = (B == 1 && io.host != nullptr); // single state: poll the kernel's completion word (HostIO::finish)
  rc = ccmp_project_batch(ctx, p, (const double *)io.dev, (double *)io.dev, (uint8_t *)(io.dev + off_ok), (uint16_t *)(io.dev + off_it), B,
                          ctx->stream);
  if (rc != CCMP_OK) { ctx->want_done = ctx->done_armed = false; return rc; }
  if ((rc = io.out(q_out, 0, qb)) != CCMP_OK) return io.abandon(rc);
  if ((rc = io.out(ok, off_ok, B)) != CCMP_OK) return io.abandon(rc);
  if (iters && (rc = io.out(iters, off_it, B * sizeof(uint16_t))) != CCMP_OK) return io.abandon(rc);
  return io.finish();
}

int ccmp_function_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, double *f, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q || !f) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  if (B == 1 && ctx->resident_on) {
    const int rc = ccmp_host::resident_call(ctx, p, ccmp_host::ResidentCall{kResFunction, q, nullptr, f, nullptr, nullptr});
    if (rc != ccmp_host::kResidentFallBack) return rc;
  }
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_f = (qb + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_f + B * 2 * sizeof(double));
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, q, qb)) != CCMP_OK) return rc;
  ctx->want_done = (B == 1 && io.host != nullptr);
  rc = ccmp_function_batch(ctx, p, (const double *)io.dev, (double *)(io.dev + off_f), B, ctx->stream);
  if (rc != CCMP_OK) { ctx->want_done = ctx->done_armed = false; return rc; }
  if ((rc = io.out(f, off_f, B * 2 * sizeof(double))) != CCMP_OK) return io.abandon(rc);
  return io.finish();
}

int ccmp_is_satisfied_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q || !ok) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  if (B == 1 && ctx->resident_on) {
    const int rc = ccmp_host::resident_call(ctx, p, ccmp_host::ResidentCall{kResIsSatisfied, q, nullptr, nullptr, ok, nullptr});
    if (rc != ccmp_host::kResidentFallBack) return rc;
  }
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_ok = (qb + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_ok + B);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, q, qb)) != CCMP_OK) return rc;
  ctx->want_done = (B == 1 && io.host != nullptr);
  rc = ccmp_is_satisfied_batch(ctx, p, (const double *)io.dev, (uint8_t *)(io.dev + off_ok), B, ctx->stream);
  if (rc != CCMP_OK) { ctx->want_done = ctx->done_armed = false; return rc; }
  if ((rc = io.out(ok, off_ok, B)) != CCMP_OK) return io.abandon(rc);
  return io.finish();
}

int ccmp_joint_valid_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q || !ok) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  if (B == 1 && ctx->resident_on) {
    const int rc = ccmp_host::resident_call(ctx, p, ccmp_host::ResidentCall{kResJointValid, q, nullptr, nullptr, ok, nullptr});
    if (rc != ccmp_host::kResidentFallBack) return rc;
  }
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_ok = (qb + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_ok + B);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, q, qb)) != CCMP_OK) return rc;
  ctx->want_done = (B == 1 && io.host != nullptr);
  rc = ccmp_joint_valid_batch(ctx, p, (const double *)io.dev, (uint8_t *)(io.dev + off_ok), B, ctx->stream);
  if (rc != CCMP_OK) { ctx->want_done = ctx->done_armed = false; return rc; }
  if ((rc = io.out(ok, off_ok, B)) != CCMP_OK) return io.abandon(rc);
  return io.finish();
}

int ccmp_sample_project_host(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index, double *q_out,
                             uint8_t *ok, uint16_t *iters, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q_out || !ok) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_ok = (qb + 255) & ~(size_t)255;
  const size_t off_it = (off_ok + B + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_it + B * sizeof(uint16_t));
  if (rc != CCMP_OK) return rc;
  rc = ccmp_sample_project_batch(ctx, p, seed, first_index, (double *)io.dev, (uint8_t *)(io.dev + off_ok), (uint16_t *)(io.dev + off_it),
                                 nullptr, B, ctx->stream);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.out(q_out, 0, qb)) != CCMP_OK) return io.abandon(rc);
  if ((rc = io.out(ok, off_ok, B)) != CCMP_OK) return io.abandon(rc);
  if (iters && (rc = io.out(iters, off_it, B * sizeof(uint16_t))) != CCMP_OK) return io.abandon(rc);
  return io.finish();
}

int ccmp_sample_ref_project_host(ccmp_ctx *ctx, const ccmp_problem *p, int kind, uint64_t seed, uint64_t first_index,
                                 const double ref[14], double param, double *q_out, uint8_t *ok, uint16_t *iters, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!ref || !q_out || !ok || (kind != 0 && kind != 1)) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_ok = (qb + 255) & ~(size_t)255;
  const size_t off_it = (off_ok + B + 255) & ~(size_t)255;
  const size_t off_ref = (off_it + B * sizeof(uint16_t) + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_ref + 14 * sizeof(double));
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(off_ref, ref, 14 * sizeof(double))) != CCMP_OK) return rc;
  rc = (kind == 0 ? ccmp_sample_near_project_batch : ccmp_sample_gaussian_project_batch)(
      ctx, p, seed, first_index, (const double *)(io.dev + off_ref), 0, param, (double *)io.dev, (uint8_t *)(io.dev + off_ok),
      (uint16_t *)(io.dev + off_it), nullptr, B, ctx->stream);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.out(q_out, 0, qb)) != CCMP_OK) return io.abandon(rc);
  if ((rc = io.out(ok, off_ok, B)) != CCMP_OK) return io.abandon(rc);
  if (iters && (rc = io.out(iters, off_it, B * sizeof(uint16_t))) != CCMP_OK) return io.abandon(rc);
  return io.finish();
}

// proxy-geometry clearance on host states (ccmp_scene.cpp holds the scene and the device-pointer entry); a single state
// goes through the pinned block: memcpy, launch, synchronise, memcpy
int ccmp_clearance_host(ccmp_ctx *ctx, const ccmp_problem *p, const ccmp_scene *scene, const double *q, size_t B, double margin,
                        double *clearance, int32_t *pair, uint8_t *free_out)
{
  if (!ctx || !p || !scene) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q || !clearance) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_c = (qb + 255) & ~(size_t)255;
  const size_t off_p = (off_c + B * sizeof(double) + 255) & ~(size_t)255;
  const size_t off_f = (off_p + B * sizeof(int32_t) + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_f + B);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, q, qb)) != CCMP_OK) return rc;
  ctx->want_done = (B == 1 && io.host != nullptr);
  rc = ccmp_clearance_batch(ctx, p, scene, (const double *)io.dev, nullptr, B, margin, (double *)(io.dev + off_c),
                            (int32_t *)(io.dev + off_p), (uint8_t *)(io.dev + off_f), ctx->stream);
  if (rc != CCMP_OK) { ctx->want_done = ctx->done_armed = false; return rc; }
  if ((rc = io.out(clearance, off_c, B * sizeof(double))) != CCMP_OK) return io.abandon(rc);
  if (pair && (rc = io.out(pair, off_p, B * sizeof(int32_t))) != CCMP_OK) return io.abandon(rc);
  if (free_out && (rc = io.out(free_out, off_f, B)) != CCMP_OK) return io.abandon(rc);
  return io.finish();
}

static int geodesic_host_common(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                                double *states, int32_t *n_states, uint8_t *ok, const double *carry_in, double *carry_out,
                                int round_budget, int check_target)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (E == 0) return CCMP_OK;
  if (!from || !to || !states || !n_states || !ok || max_states < 1) return CCMP_EINVAL;
  if ((carry_in || carry_out || round_budget > 0) && max_states < 2) return CCMP_EINVAL; // as geodesic_common: before any buffer is touched
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  // ONE edge, as the unchanged planner asks for them (discreteGeodesic / checkMotion of a single pair, src/planner/stefanBiPRM.cpp:
  // 315-318,397-398): through the resident service kernel when the option is on — no launch on the call path
  if (E == 1 && ctx->resident_on && !carry_in && max_states <= kResMaxStates && p->delta > 0 && p->lambda > 0) {
    ccmp_host::ResidentCall call{kResGeodesic, from, nullptr, nullptr, ok, nullptr};
    call.to = to;
    call.max_states = max_states;
    call.round_budget = round_budget;
    call.check_target = check_target;
    call.states = states;
    call.n_states = n_states;
    call.carry_out = carry_out;
    const int rc = ccmp_host::resident_call(ctx, p, call);
    if (rc != ccmp_host::kResidentFallBack) return rc;
  }
  const size_t eb = E * 14 * sizeof(double);
  const size_t sb = E * (size_t)max_states * 14 * sizeof(double);
  const size_t cb = E * 2 * sizeof(double);
  const size_t off_to = (eb + 255) & ~(size_t)255;
  const size_t off_st = (off_to + eb + 255) & ~(size_t)255;
  const size_t off_n = (off_st + sb + 255) & ~(size_t)255;
  const size_t off_ok = (off_n + E * sizeof(int32_t) + 255) & ~(size_t)255;
  const size_t off_ci = (off_ok + E + 255) & ~(size_t)255;
  const size_t off_co = (off_ci + cb + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_co + cb);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, from, eb)) != CCMP_OK) return rc;
  if ((rc = io.in(off_to, to, eb)) != CCMP_OK) return rc;
  if (carry_in && (rc = io.in(off_ci, carry_in, cb)) != CCMP_OK) return rc;
  rc = ccmp_geodesic_batch_ex(ctx, p, (const double *)io.dev, (const double *)(io.dev + off_to), E, max_states, (double *)(io.dev + off_st),
                              (int32_t *)(io.dev + off_n), (uint8_t *)(io.dev + off_ok), nullptr,
                              carry_in ? (const double *)(io.dev + off_ci) : nullptr, carry_out ? (double *)(io.dev + off_co) : nullptr,
                              round_budget, check_target, ctx->stream);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.out(states, off_st, sb)) != CCMP_OK) return io.abandon(rc);
  if ((rc = io.out(n_states, off_n, E * sizeof(int32_t))) != CCMP_OK) return io.abandon(rc);
  if ((rc = io.out(ok, off_ok, E)) != CCMP_OK) return io.abandon(rc);
  if (carry_out && (rc = io.out(carry_out, off_co, cb)) != CCMP_OK) return io.abandon(rc);
  return io.finish();
}

int ccmp_geodesic_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                       double *states, int32_t *n_states, uint8_t *ok)
{
  return geodesic_host_common(ctx, p, from, to, E, max_states, states, n_states, ok, nullptr, nullptr, 0, 0);
}

int ccmp_check_motion_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                           double *states, int32_t *n_states, uint8_t *ok)
{
  return geodesic_host_common(ctx, p, from, to, E, max_states, states, n_states, ok, nullptr, nullptr, 0, 1);
}

int ccmp_geodesic_host_ex(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                          double *states, int32_t *n_states, uint8_t *ok, const double *carry_in, double *carry_out, int round_budget,
                          int check_target)
{
  if (carry_in && check_target) return CCMP_EINVAL;
  if (round_budget > 0 && !carry_out) return CCMP_EINVAL; // a suspended edge is useless without what its continuation needs
  return geodesic_host_common(ctx, p, from, to, E, max_states, states, n_states, ok, carry_in, carry_out, round_budget, check_target);
}

// One shard of ccmp_*_sharded_host, on its context's device and stream: upload (mode 0), project, download, wait.  Runs
// on its own thread when there are several (for_each_shard): a copy from pageable memory returns only when the data is
// staged and a copy into pageable memory only when it has arrived, so shards driven from ONE thread start and finish one
// after the other — 8 x 1.2 ms of stagger against a 16 ms kernel at 8 GPUs.
struct ShardedArgs {
  ccmp_ctx *const *ctxs;
  int n, mode;
  const ccmp_problem *p;
  const double *q_in;
  double *q_out;
  uint8_t *ok;
  uint16_t *iters;
  uint64_t seed, first_index;
  size_t B;
  std::chrono::steady_clock::time_point t0;
};

static int sharded_one(int g, void *arg)
{
  const ShardedArgs &A = *(const ShardedArgs *)arg;
  const size_t base = A.B / (size_t)A.n, rem = A.B % (size_t)A.n;
  const size_t lo = (size_t)g * base + ((size_t)g < rem ? (size_t)g : rem), nb = base + ((size_t)g < rem ? 1 : 0);
  ccmp_ctx *ctx = A.ctxs[g];
  ctx->shard_launch_ms = -1.0;
  if (nb == 0) return CCMP_OK;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t qb = nb * 14 * sizeof(double);
  const size_t off_ok = (qb + 255) & ~(size_t)255, off_it = (off_ok + nb + 255) & ~(size_t)255;
  int rc = ensure_stage(ctx, off_it + nb * sizeof(uint16_t));
  if (rc != CCMP_OK) return rc;
  if (!ctx->ev_shard && hipEventCreate(&ctx->ev_shard) != hipSuccess) ctx->ev_shard = nullptr; // timing is optional
  char *stage = (char *)ctx->stage;
  hipError_t e = hipSuccess;
  if (A.mode == 0) e = hipMemcpyAsync(stage, A.q_in + lo * 14, qb, hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(H2D shard)");
  // the upload is behind the host now (pageable source) or queued (page-locked source): from here the stream runs kernels
  if (ctx->ev_shard) (void)hipEventRecord(ctx->ev_shard, ctx->stream);
  ctx->shard_launch_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - A.t0).count();
  if (A.mode == 0)
    rc = ccmp_project_batch(ctx, A.p, (const double *)stage, (double *)stage, (uint8_t *)(stage + off_ok), (uint16_t *)(stage + off_it), nb,
                            ctx->stream);
  else
    rc = ccmp_sample_project_batch(ctx, A.p, A.seed, A.first_index + lo, (double *)stage, (uint8_t *)(stage + off_ok),
                                   (uint16_t *)(stage + off_it), nullptr, nb, ctx->stream);
  if (rc == CCMP_OK) {
    e = hipMemcpyAsync(A.q_out + lo * 14, stage, qb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(A.ok + lo, stage + off_ok, nb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && A.iters) e = hipMemcpyAsync(A.iters + lo, stage + off_it, nb * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpyAsync(D2H shard)");
  }
  e = hipStreamSynchronize(ctx->stream); // also on the error path: the caller's buffers must be quiet
  if (e != hipSuccess && rc == CCMP_OK) rc = hip_fail(e, "hipStreamSynchronize(shard)");
  return rc;
}

static int sharded_common(ccmp_ctx *const *ctxs, int n, const ccmp_problem *p, int mode, const double *q_in, double *q_out,
                          uint8_t *ok, uint16_t *iters, uint64_t seed, uint64_t first_index, size_t B)
{
  if (!ctxs || n < 1 || n > 64 || !p) return CCMP_EINVAL;
  for (int g = 0; g < n; g++) {
    if (!ctxs[g]) return CCMP_EINVAL;
    for (int h = 0; h < g; h++)
      if (ctxs[h] == ctxs[g]) return CCMP_EINVAL; // one context holds one shard's staging and queues
  }
  if (B == 0) return CCMP_OK;
  if (!q_out || !ok || (mode == 0 && !q_in)) return CCMP_EINVAL;
  ShardedArgs A{ctxs, n, mode, p, q_in, q_out, ok, iters, seed, first_index, B, std::chrono::steady_clock::now()};
  return ccmp_host::for_each_shard(n, sharded_one, &A);
}

int ccmp_sharded_host_last_timing(ccmp_ctx *const *ctxs, int n, double *launch_ms, double *start_ms)
{
  if (!ctxs || n < 1 || n > 64 || !launch_ms || !start_ms) return CCMP_EINVAL;
  for (int g = 0; g < n; g++) {
    if (!ctxs[g]) return CCMP_EINVAL;
    launch_ms[g] = ctxs[g]->shard_launch_ms;
    start_ms[g] = -1.0;
    if (ctxs[g]->device != ctxs[0]->device || !ctxs[g]->ev_shard || !ctxs[0]->ev_shard || ctxs[g]->shard_launch_ms < 0 ||
        ctxs[0]->shard_launch_ms < 0)
      continue;
    DeviceGuard guard(ctxs[0]->device);
    float ms = 0.0f;
    if (g == 0) start_ms[g] = 0.0;
    else if (hipEventElapsedTime(&ms, ctxs[0]->ev_shard, ctxs[g]->ev_shard) == hipSuccess) start_ms[g] = (double)ms;
    else (void)hipGetLastError();
  }
  return CCMP_OK;
}

int ccmp_project_sharded_host(ccmp_ctx *const *ctxs, int n, const ccmp_problem *p, const double *q_in, double *q_out,
                              uint8_t *ok, uint16_t *iters, size_t B)
{
  return sharded_common(ctxs, n, p, 0, q_in, q_out, ok, iters, 0, 0, B);
}

int ccmp_sample_project_sharded_host(ccmp_ctx *const *ctxs, int n, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                                     double *q_out, uint8_t *ok, uint16_t *iters, size_t B)
{
  return sharded_common(ctxs, n, p, 1, nullptr, q_out, ok, iters, seed, first_index, B);
}

} // extern "C"
