// ccmp_resident.cpp — host side of the opt-in resident service kernel (ccmp_resident.h: purpose, mailbox, and the five rules that
// keep a never-ending kernel from hanging anything).
#include "ccmp_resident.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <new>

#include "ccmp_ctx.h"
#include "ccmp_host.h"

using ccmp_host::DeviceGuard;
using ccmp_host::hip_fail;

extern "C" hipError_t ccmp_launch_resident(int stock, void *box_dev, unsigned long long last_tag, unsigned long long idle_ticks, hipStream_t st);

struct ccmp_resident {
  char *box = nullptr;      // pinned, device-mapped, coherent
  void *box_dev = nullptr;
  hipStream_t stream = nullptr;
  bool launched = false;    // a kernel was put on `stream` and has not been waited for
  bool abandoned = false;   // ... and did not get to run within the start bound: told to stop, not waited for (it leaves as soon as it runs)
  bool broken = false;      // a request was not answered within its bound: the service is not used again by this context
  double retry_after_ms = 0, backoff_ms = 0; // no new start before this time (steady clock); doubled by every start that gave up
  int stock = -1;           // which instantiation runs
  unsigned int tag = 0;     // sequence number of the last request
  unsigned int consts_seq = 0;
  bool have_problem = false;
  ccmp_problem problem;     // the problem whose constants the mailbox holds
  int stock_kernels = -1;   // ... under this setting of the option
};

namespace {

inline volatile unsigned long long *word(ccmp_resident *r, size_t byte_off) { return reinterpret_cast<volatile unsigned long long *>(r->box + byte_off); }

double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// a request line's tag: the request's sequence number (low half) and a checksum of the line's seven payload words (high half) — the
// device accepts a line only when the two belong together (ccmp_kernels_resident.hip)
unsigned long long line_tag(unsigned int seq, const volatile unsigned long long *w7)
{
  unsigned long long h = 0;
  for (int i = 0; i < 7; i++) {
    const unsigned long long v = w7[i];
    const int r = 7 * i + 1;
    h ^= (v << r) | (v >> (64 - r));
  }
  return (unsigned long long)seq | ((unsigned long long)(unsigned int)(h ^ (h >> 32)) << 32);
}
// payloads are in place: the five tags, last
void post(volatile unsigned long long *req, unsigned int seq)
{
  for (int line = 0; line < 5; line++) __atomic_store_n(&req[8 * line + 7], line_tag(seq, req + 8 * line), __ATOMIC_RELEASE);
}

// waits for the kernel on the service's stream to be gone (it has been told to stop, or stopped by itself)
void drain(ccmp_resident *r)
{
  if (!r->launched) return;
  (void)hipStreamSynchronize(r->stream);
  r->launched = false;
  r->abandoned = false;
}

// Has the kernel on the stream ended?  (kResExited is the last thing it writes.)
bool exited(ccmp_resident *r) { return __atomic_load_n(word(r, kResStateOff), __ATOMIC_ACQUIRE) == (unsigned long long)kResExited; }

// Every wait on the host is bounded (ccmp_resident.h rule 4).  A kernel that was ABANDONED — queued behind another context's
// resident kernel on a shared hardware queue, told to stop, never seen running — or one that stopped answering is not waited for
// by quiesce(): it holds no CU until it runs, and it leaves at once when it does; what the caller's hipFree / hipMalloc then waits
// for is at most the OTHER kernel's idle time, which is HIP's business, not a spin of ours.
void stop(ccmp_resident *r)
{
  if (!r || !r->launched) return;
  if (r->abandoned || r->broken) {
    if (exited(r)) drain(r); // it got to run meanwhile and is gone: the stream is idle, nothing to wait for
    return;
  }
  if (!exited(r)) {
    // a request whose command is "stop": payload first, the five tags last (ccmp_resident.h)
    const unsigned int tag = ++r->tag;
    volatile unsigned long long *req = word(r, kResReqOff);
    req[32] = (unsigned long long)kResStop;
    post(req, tag);
  }
  drain(r);
}

int start(ccmp_ctx *ctx, ccmp_resident *r, int stock)
{
  if (!r->box) {
    HIP_TRY(hipHostMalloc((void **)&r->box, kResBoxBytes, hipHostMallocMapped | hipHostMallocCoherent));
    memset(r->box, 0, kResBoxBytes);
    hipError_t e = hipHostGetDevicePointer(&r->box_dev, r->box, 0);
    if (e != hipSuccess) { (void)hipHostFree(r->box); r->box = nullptr; return hip_fail(e, "hipHostGetDevicePointer(resident mailbox)"); }
  }
  if (!r->stream) {
    // lowest priority: a stream of its own priority has a hardware queue of its own — nothing else is ever queued behind the kernel
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e == hipSuccess && least != greatest) e = hipStreamCreateWithPriority(&r->stream, hipStreamNonBlocking, least);
    else e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return hip_fail(e, "hipStreamCreate(resident)");
  }
  drain(r); // (a kernel that left by itself)
  __atomic_store_n(word(r, kResStateOff), (unsigned long long)kResStarting, __ATOMIC_RELEASE);
  const unsigned long long idle_ticks = (unsigned long long)ctx->resident_idle_ms * 100000ull; // wall_clock64: 100 MHz
  HIP_TRY(ccmp_launch_resident(stock, r->box_dev, r->tag, idle_ticks, r->stream));
  r->launched = true;
  r->stock = stock;
  // The kernel reports itself running with its first instructions (~20 us behind the launch).  If it does not within 5 ms — two
  // orders of magnitude above launch jitter — its stream is queued behind something that does not end soon: another context's
  // resident kernel on the same hardware queue (streams of one priority share a few), or this process's own persistent batch
  // kernels filling every CU.  THIS CALL then takes the launch path and the kernel is abandoned: told to stop (it leaves as soon
  // as it gets to run), not waited for; a later call starts the service again once that kernel is gone and a back-off has passed
  // (10 ms, doubled by every start that gave up, at most 1 s).  The option stays on; "resident_gave_up" counts the occasions.
  const double t0 = now_ms();
  while (__atomic_load_n(word(r, kResStateOff), __ATOMIC_ACQUIRE) == (unsigned long long)kResStarting) {
    if (now_ms() - t0 > 5.0) {
      const unsigned int tag = ++r->tag;
      volatile unsigned long long *req = word(r, kResReqOff);
      req[32] = (unsigned long long)kResStop;
      post(req, tag);
      r->abandoned = true;
      r->backoff_ms = r->backoff_ms > 0 ? (r->backoff_ms * 2 > 1000.0 ? 1000.0 : r->backoff_ms * 2) : 10.0;
      r->retry_after_ms = now_ms() + r->backoff_ms;
      ctx->resident_gave_up++;
      return ccmp_host::kResidentFallBack;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  r->backoff_ms = 0;
  return CCMP_OK;
}

// May the service be (re)started now?  Not while an abandoned kernel is still queued, not before the back-off has passed, never
// again once a request went unanswered.
bool may_start(ccmp_resident *r)
{
  if (r->broken) return false;
  if (r->abandoned) {
    if (!exited(r)) return false;
    drain(r); // gone: the stream is idle
  }
  return now_ms() >= r->retry_after_ms;
}

}  // namespace

namespace ccmp_host {

void quiesce(ccmp_ctx *ctx)
{
  if (ctx && ctx->resident) stop(ctx->resident);
}

void resident_destroy(ccmp_ctx *ctx)
{
  ccmp_resident *r = ctx ? ctx->resident : nullptr;
  if (!r) return;
  stop(r);
  if (r->launched) {
    // an abandoned (or unresponsive) kernel is still on the stream: it may yet run and read its mailbox.  Give it 100 ms to be gone;
    // otherwise the mailbox (64 KB of pinned memory) and the stream are LEFT to the process rather than freed under a kernel or
    // waited for without bound.
    const double t0 = now_ms();
    while (!exited(r) && now_ms() - t0 < 100.0) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    if (exited(r)) drain(r);
  }
  if (!r->launched) {
    if (r->stream) (void)hipStreamDestroy(r->stream);
    if (r->box) (void)hipHostFree(r->box);
  }
  delete r;
  ctx->resident = nullptr;
}

int resident_set(ccmp_ctx *ctx, long on)
{
  if (!ctx || (on != 0 && on != 1)) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  if (!on) {
    resident_destroy(ctx);
    ctx->resident_on = 0;
    return CCMP_OK;
  }
  if (!ctx->resident) {
    ctx->resident = new (std::nothrow) ccmp_resident();
    if (!ctx->resident) return CCMP_ENOMEM;
  }
  ctx->resident_on = 1; // started by the first single-state call: it brings the problem the kernel is instantiated for
  ctx->resident_gave_up = 0; // (turning the option on again is also how a caller asks for a fresh try at once)
  ctx->resident->backoff_ms = 0;
  ctx->resident->retry_after_ms = 0;
  return CCMP_OK;
}

int resident_call(ccmp_ctx *ctx, const ccmp_problem *p, const ResidentCall &call)
{
  ccmp_resident *r = ctx->resident;
  if (!ctx->resident_on || !r || !p || p->jacobian_mode != CCMP_JAC_FD) return kResidentFallBack; // reference arithmetic only
  if (r->broken) return kResidentFallBack;
  if (r->abandoned && !may_start(r)) return kResidentFallBack; // an abandoned kernel is still queued, or the back-off has not passed
  // a request whose worst case (max_iter rounds for each of max_states states, ~20 us each under load) exceeds what the answer is
  // waited for goes to the launch path instead of timing out spuriously
  const double worst_ms = 0.02 * (double)p->max_iter * (double)(call.cmd == kResGeodesic ? (call.max_states > 0 ? call.max_states : 1) : 1);
  if (worst_ms > 1500.0) return kResidentFallBack;
  // the problem in force: its kernel constants go into the mailbox when it changes (a planner sets its problem up once)
  if (!r->have_problem || r->stock_kernels != ctx->stock_kernels || memcmp(&r->problem, p, sizeof *p) != 0) {
    if (!(p->tol_pos > 0) || !(p->tol_rot > 0) || p->max_iter < 0 || p->max_iter > 65535) return CCMP_EINVAL;
    ccmp_consts K;
    make_consts(*p, K);
    if (!ctx->stock_kernels) K.stock = K.twin_arms = 0;
    if (r->launched && r->stock != (K.stock ? 1 : 0)) stop(r); // the other instantiation of the kernel
    static_assert(sizeof(ccmp_consts) <= kResStateOff, "the constants fit in front of the state word");
    if (!r->box) { // (first use: the mailbox comes with the first start)
      if (!may_start(r)) return kResidentFallBack;
      int rc = start(ctx, r, K.stock ? 1 : 0);
      if (rc != CCMP_OK) return rc; // (kResidentFallBack: the service could not get a queue of its own)
    }
    memcpy(r->box + kResConstsOff, &K, sizeof K);
    r->consts_seq++;
    r->problem = *p;
    r->have_problem = true;
    r->stock_kernels = ctx->stock_kernels;
    if (!r->launched) r->stock = K.stock ? 1 : 0;
  }
  // (re)start: never started, stopped by quiesce(), or left by itself after its idle time
  if (!r->launched || __atomic_load_n(word(r, kResStateOff), __ATOMIC_ACQUIRE) == (unsigned long long)kResExited) {
    if (!may_start(r)) return kResidentFallBack;
    int rc = start(ctx, r, r->stock);
    if (rc != CCMP_OK) return rc;
  }
  // ---- the request: payloads, then the five tags ---------------------------------------------------------------------------------
  const unsigned int tag = ++r->tag;
  volatile unsigned long long *req = word(r, kResReqOff);
  volatile unsigned long long *resp = word(r, kResRespOff);
  for (int i = 0; i < 7; i++) {
    unsigned long long a, b;
    memcpy(&a, call.x + i, 8);
    memcpy(&b, call.x + 7 + i, 8);
    req[i] = a;
    req[8 + i] = b;
  }
  req[32] = (unsigned long long)(unsigned int)call.cmd | ((unsigned long long)r->consts_seq << 32);
  if (call.cmd == kResGeodesic) {
    for (int i = 0; i < 7; i++) {
      unsigned long long a, b;
      memcpy(&a, call.to + i, 8);
      memcpy(&b, call.to + 7 + i, 8);
      req[16 + i] = a;
      req[24 + i] = b;
    }
    unsigned long long d, l;
    memcpy(&d, &p->delta, 8);
    memcpy(&l, &p->lambda, 8);
    req[33] = (unsigned long long)(unsigned int)call.max_states | ((unsigned long long)(unsigned int)call.round_budget << 32);
    req[34] = (unsigned long long)(unsigned int)call.check_target;
    req[35] = d;
    req[36] = l;
  }
  post(req, tag);
  // ---- the answer: bounded.  A kernel that left by itself between our look at its state and our request never answers: the state
  // word says so, and this one call takes the launch path (the next one starts the service again).
  bool done = false;
  const double t0 = now_ms();
  for (long spin = 0;; spin++) {
    if (__atomic_load_n(&resp[kResRespDone], __ATOMIC_ACQUIRE) == tag) { done = true; break; }
    if ((spin & 1023) == 1023) {
      if (__atomic_load_n(word(r, kResStateOff), __ATOMIC_ACQUIRE) == (unsigned long long)kResExited) {
        // one more look: the answer may have been written just before the kernel left
        if (__atomic_load_n(&resp[kResRespDone], __ATOMIC_ACQUIRE) == tag) { done = true; break; }
        drain(r);
        return kResidentFallBack;
      }
      if (now_ms() - t0 > 2000.0) break; // 250 Newton rounds are under a millisecond
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  if (!done) {
    // never post into a service that stopped answering: told to stop, marked unusable (later calls take the launch path; quiesce()
    // and destroy do not wait for it)
    snprintf(g_hip_err, sizeof g_hip_err, "resident service kernel: no answer within 2 s (state %llu); the service is not used again by this context",
             (unsigned long long)__atomic_load_n(word(r, kResStateOff), __ATOMIC_ACQUIRE));
    const unsigned int stop_tag = ++r->tag;
    req[32] = (unsigned long long)kResStop;
    post(req, stop_tag);
    r->broken = true;
    return CCMP_EHIP;
  }
  const unsigned long long flags = resp[kResRespFlags];
  if (call.cmd == kResGeodesic) {
    const unsigned long long nw = resp[kResRespN];
    int32_t n;
    memcpy(&n, &nw, 4);
    if (call.n_states) *call.n_states = n;
    if (call.ok) *call.ok = (uint8_t)(flags & 0xffull); // 0 / 1 / 2 (round budget spent)
    const int rows = n > call.max_states ? call.max_states : (n < 0 ? 0 : n);
    memcpy(call.states, r->box + kResStatesOff, (size_t)rows * 14 * sizeof(double));
    if (call.carry_out)
      for (int i = 0; i < 2; i++) { const unsigned long long v = resp[kResRespCarry + i]; memcpy(call.carry_out + i, &v, 8); }
    return CCMP_OK;
  }
  if (call.q_out)
    for (int i = 0; i < 14; i++) { const unsigned long long v = resp[kResRespQ + i]; memcpy(call.q_out + i, &v, 8); }
  if (call.f)
    for (int i = 0; i < 2; i++) { const unsigned long long v = resp[kResRespF + i]; memcpy(call.f + i, &v, 8); }
  if (call.ok) *call.ok = (uint8_t)(flags & 1ull);
  if (call.iters) *call.iters = (uint16_t)(flags >> 32);
  return CCMP_OK;
}

}  // namespace ccmp_host
