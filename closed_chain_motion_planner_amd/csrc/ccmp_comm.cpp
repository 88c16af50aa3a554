// ccmp_comm.cpp — one process, several GPUs, with the collective: ccmp_comm_* and ccmp_project_sharded /
// ccmp_sample_project_sharded of include/ccmp.h.  Host code only; librccl is opened at run time.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include <rccl/rccl.h> // types and prototypes only: the library itself is opened at run time (no link dependency)

#include "../../include/ccmp.h"
#include "ccmp_ctx.h"
#include "ccmp_resident.h"

using ccmp_host::DeviceGuard;
using ccmp_host::ensure_stage;
using ccmp_host::g_hip_err;
using ccmp_host::hip_fail;

extern "C" {
// ---- one process, several GPUs, RCCL all-gather of the valid states ---------------------------------------------------
// The reference's planner is ONE process (src/main.cpp); this is SURVEY.md §8b's ccmp_project_sharded: every GPU projects
// its contiguous shard, compacts its valid states into a fixed-capacity block (row 0 = count) and joins ONE ncclAllGather
// over xGMI; GPU 0 then holds every shard's valid states and hands them to the host tree in global sample order.
// librccl is opened at run time (dlopen): libccmp.so has no link-time dependency on it, and a process that already
// carries an RCCL (PyTorch's) shares that copy.
} // extern "C"

namespace {

struct RcclApi {
  void *handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

RcclApi &rccl()
{
  static RcclApi api = [] {
    RcclApi a;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (!a.handle) return a;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.handle, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
    a.GroupStart = (decltype(a.GroupStart))dlsym(a.handle, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.handle, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
    a.ok = a.CommInitAll && a.CommDestroy && a.AllGather && a.GroupStart && a.GroupEnd && a.GetErrorString;
    return a;
  }();
  return api;
}

int rccl_fail(ncclResult_t r, const char *what)
{
  snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
  return CCMP_ECOMM;
}

} // namespace

extern "C" {

struct ccmp_comm {
  int n = 0;
  std::vector<ccmp_ctx *> ctxs;
  std::vector<ncclComm_t> comms;
  std::vector<double *> send, recv; // per GPU: (cap + 1) x 14 and n x (cap + 1) x 14 doubles
  size_t cap = 0;                   // rows per block
  double *host_recv = nullptr;      // pinned staging of GPU 0's gathered blocks
  size_t host_cap = 0;
  // timing of the last sharded call, per GPU, on that GPU's stream: [start -> shard projected and compacted] and
  // [-> all-gather complete] (ccmp_comm_last_timing; what a first multi-GPU run is diagnosed by)
  std::vector<hipEvent_t> ev; // 3 per GPU
  std::vector<float> kernel_ms, gather_ms;
};

int ccmp_comm_create(ccmp_ctx *const *ctxs, int n, ccmp_comm **out)
{
  if (!ctxs || !out || n < 1 || n > 64) return CCMP_EINVAL;
  *out = nullptr;
  std::vector<int> devs(n);
  for (int g = 0; g < n; g++) {
    if (!ctxs[g]) return CCMP_EINVAL;
    devs[g] = ctxs[g]->device;
    for (int h = 0; h < g; h++)
      if (devs[h] == devs[g]) return CCMP_EINVAL; // RCCL wants one rank per device
  }
  if (!rccl().ok) {
    const char *why = rccl().handle ? nullptr : dlerror(); // dlerror() clears itself: read it once
    snprintf(g_hip_err, sizeof g_hip_err, "librccl.so could not be opened: %s", why ? why : "symbols missing");
    return CCMP_ECOMM;
  }
  ccmp_comm *c = new (std::nothrow) ccmp_comm();
  if (!c) return CCMP_ENOMEM;
  c->n = n;
  c->ctxs.assign(ctxs, ctxs + n);
  c->comms.assign(n, nullptr);
  c->send.assign(n, nullptr);
  c->recv.assign(n, nullptr);
  ncclResult_t r = rccl().CommInitAll(c->comms.data(), n, devs.data());
  if (r != ncclSuccess) { delete c; return rccl_fail(r, "ncclCommInitAll"); }
  c->ev.assign(3 * (size_t)n, nullptr);
  c->kernel_ms.assign(n, -1.0f);
  c->gather_ms.assign(n, -1.0f);
  for (int g = 0; g < n; g++) {
    DeviceGuard guard(devs[g]);
    for (int k = 0; k < 3; k++)
      if (hipEventCreate(&c->ev[3 * g + k]) != hipSuccess) c->ev[3 * g + k] = nullptr; // timing is optional: the call works without
  }
  *out = c;
  return CCMP_OK;
}

int ccmp_comm_last_timing(const ccmp_comm *c, double *kernel_ms, double *gather_ms)
{
  if (!c || !kernel_ms || !gather_ms) return CCMP_EINVAL;
  for (int g = 0; g < c->n; g++) { kernel_ms[g] = c->kernel_ms[g]; gather_ms[g] = c->gather_ms[g]; }
  return CCMP_OK;
}

void ccmp_comm_destroy(ccmp_comm *c)
{
  if (!c) return;
  for (int g = 0; g < c->n; g++) {
    DeviceGuard guard(c->ctxs[g]->device);
    ccmp_host::quiesce(c->ctxs[g]); // (hipFree waits for the whole device: a resident service kernel of the context must be gone first)
    (void)hipStreamSynchronize(c->ctxs[g]->stream);
    if (c->comms[g]) (void)rccl().CommDestroy(c->comms[g]);
    for (int k = 0; k < 3; k++)
      if (!c->ev.empty() && c->ev[3 * g + k]) (void)hipEventDestroy(c->ev[3 * g + k]);
    if (c->send[g]) (void)hipFree(c->send[g]);
    if (c->recv[g]) (void)hipFree(c->recv[g]);
  }
  if (c->host_recv) (void)hipHostFree(c->host_recv);
  delete c;
}

static int comm_ensure_blocks(ccmp_comm *c, size_t cap)
{
  if (c->cap >= cap && c->send[0]) return CCMP_OK;
  c->cap = 0; // nothing is usable until every block below exists (a failure half-way must not leave a stale capacity)
  const size_t block = (cap + 1) * 14 * sizeof(double);
  for (int g = 0; g < c->n; g++) {
    DeviceGuard guard(c->ctxs[g]->device);
    if (!guard.ok) return CCMP_ENODEV;
    ccmp_host::quiesce(c->ctxs[g]); // ccmp_resident.h rule (2): before every hipFree / hipMalloc of the library's own
    if (c->send[g]) (void)hipFree(c->send[g]);
    if (c->recv[g]) (void)hipFree(c->recv[g]);
    c->send[g] = c->recv[g] = nullptr;
    HIP_TRY(hipMalloc((void **)&c->send[g], block));
    HIP_TRY(hipMalloc((void **)&c->recv[g], block * (size_t)c->n));
  }
  if (c->host_recv) (void)hipHostFree(c->host_recv);
  c->host_recv = nullptr;
  HIP_TRY(hipHostMalloc((void **)&c->host_recv, block * (size_t)c->n, hipHostMallocDefault));
  c->cap = cap;
  return CCMP_OK;
}

// The three phases of a sharded call.  Phases 1 and 3 touch the caller's host buffers — uploads from and downloads into
// memory that is usually pageable, i.e. copies that block the thread that issues them — and therefore run on one
// short-lived thread per GPU (ccmp_host::for_each_shard): issued from one thread, GPU g would start g uploads late and
// hand its results back g downloads late (8-9 ms of stagger against a 16 ms kernel at 8 GPUs).
struct CommShard { size_t lo, nb, off_ok, off_it; };
struct CommArgs {
  ccmp_comm *c;
  const ccmp_problem *p;
  int mode;
  const double *q_in;
  uint64_t seed, first_index;
  size_t B;
  double *q_out;
  uint8_t *ok;
  uint16_t *iters;
  std::vector<CommShard> *sh;
  std::chrono::steady_clock::time_point t0;
};

// phase 1: upload (mode 0), project and compact one shard into its send block, all on its GPU's stream
static int comm_phase1(int g, void *arg)
{
  CommArgs &A = *(CommArgs *)arg;
  ccmp_comm *c = A.c;
  const int n = c->n;
  CommShard &S = (*A.sh)[(size_t)g];
  const size_t base = A.B / (size_t)n, rem = A.B % (size_t)n;
  S.lo = (size_t)g * base + ((size_t)g < rem ? (size_t)g : rem);
  S.nb = base + ((size_t)g < rem ? 1 : 0);
  ccmp_ctx *ctx = c->ctxs[g];
  ctx->shard_launch_ms = -1.0;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t nb = S.nb, qb = nb * 14 * sizeof(double);
  S.off_ok = (qb + 255) & ~(size_t)255;
  S.off_it = (S.off_ok + nb + 255) & ~(size_t)255;
  if (c->ev[3 * g]) (void)hipEventRecord(c->ev[3 * g], ctx->stream);
  hipError_t e = hipMemsetAsync(c->send[g], 0, 14 * sizeof(double), ctx->stream); // row 0: count 0 for an empty shard
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(send block)");
  if (nb == 0) return CCMP_OK;
  int rc = ensure_stage(ctx, S.off_it + nb * sizeof(uint16_t));
  if (rc != CCMP_OK) return rc;
  char *stage = (char *)ctx->stage;
  if (A.mode == 0) {
    e = hipMemcpyAsync(stage, A.q_in + S.lo * 14, qb, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(H2D shard)");
  }
  ctx->shard_launch_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - A.t0).count();
  if (A.mode == 0)
    rc = ccmp_project_batch(ctx, A.p, (const double *)stage, (double *)stage, (uint8_t *)(stage + S.off_ok), (uint16_t *)(stage + S.off_it), nb,
                            ctx->stream);
  else
    rc = ccmp_sample_project_batch(ctx, A.p, A.seed, A.first_index + S.lo, (double *)stage, (uint8_t *)(stage + S.off_ok),
                                   (uint16_t *)(stage + S.off_it), nullptr, nb, ctx->stream);
  if (rc != CCMP_OK) return rc;
  rc = ccmp_compact_valid_capped(ctx, (const double *)stage, (const uint8_t *)(stage + S.off_ok), nb, c->send[g] + 14, c->cap,
                                 (uint64_t *)c->send[g], ctx->stream);
  if (rc == CCMP_OK && c->ev[3 * g + 1]) (void)hipEventRecord(c->ev[3 * g + 1], ctx->stream);
  return rc;
}

// phase 3: GPU 0 returns the gathered blocks; every GPU returns its shard's full results if the caller wants them; wait
static int comm_phase3(int g, void *arg)
{
  CommArgs &A = *(CommArgs *)arg;
  ccmp_comm *c = A.c;
  const CommShard &S = (*A.sh)[(size_t)g];
  ccmp_ctx *ctx = c->ctxs[g];
  DeviceGuard guard(ctx->device);
  // without the device current the copies below would be issued against whichever device this shard thread happens to have:
  // report that — but still wait for the shard's stream (a stream handle can be synchronised from any device context), so the
  // caller's buffers are idle when the call returns
  int rc = guard.ok ? CCMP_OK : CCMP_ENODEV;
  hipError_t e = hipSuccess;
  if (rc == CCMP_OK && g == 0) {
    e = hipMemcpyAsync(c->host_recv, c->recv[0], (c->cap + 1) * 14 * sizeof(double) * (size_t)c->n, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpyAsync(D2H gathered blocks)");
  }
  if (rc == CCMP_OK && S.nb != 0 && (A.q_out || A.ok || A.iters)) {
    const char *stage = (const char *)ctx->stage;
    if (A.q_out) e = hipMemcpyAsync(A.q_out + S.lo * 14, stage, S.nb * 14 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && A.ok) e = hipMemcpyAsync(A.ok + S.lo, stage + S.off_ok, S.nb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && A.iters)
      e = hipMemcpyAsync(A.iters + S.lo, stage + S.off_it, S.nb * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpyAsync(D2H shard)");
  }
  e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess && rc == CCMP_OK) rc = hip_fail(e, "hipStreamSynchronize(shard)");
  return rc;
}

static int comm_sync_one(int g, void *arg)
{
  ccmp_comm *c = ((CommArgs *)arg)->c;
  DeviceGuard guard(c->ctxs[g]->device);
  const hipError_t e = hipStreamSynchronize(c->ctxs[g]->stream); // (attempted whether or not the device could be made current)
  if (!guard.ok) return CCMP_ENODEV;
  return e == hipSuccess ? CCMP_OK : hip_fail(e, "hipStreamSynchronize(shard, error path)");
}

static int comm_project_common(ccmp_comm *c, const ccmp_problem *p, int mode, const double *q_in, uint64_t seed, uint64_t first_index,
                               size_t B, double *q_out, uint8_t *ok, uint16_t *iters, size_t block_rows, double *valid_out,
                               size_t valid_capacity, uint64_t *counts, uint64_t *n_valid)
{
  if (!c || !p || !valid_out || !n_valid || block_rows < 1) return CCMP_EINVAL;
  if (mode == 0 && !q_in) return CCMP_EINVAL;
  *n_valid = 0;
  if (B == 0) return CCMP_OK;
  const int n = c->n;
  int rc = comm_ensure_blocks(c, block_rows);
  if (rc != CCMP_OK) return rc;
  const size_t cap = c->cap, block_doubles = (cap + 1) * 14;
  std::vector<CommShard> sh((size_t)n, CommShard{0, 0, 0, 0});
  CommArgs A{c, p, mode, q_in, seed, first_index, B, q_out, ok, iters, &sh, std::chrono::steady_clock::now()};
  rc = ccmp_host::for_each_shard(n, comm_phase1, &A);
  // phase 2: ONE all-gather of the fixed-capacity blocks over the n devices (grouped: one call per rank of this process)
  if (rc == CCMP_OK) {
    ncclResult_t r = rccl().GroupStart();
    for (int g = 0; g < n && r == ncclSuccess; g++)
      r = rccl().AllGather(c->send[g], c->recv[g], block_doubles, ncclDouble, c->comms[g], c->ctxs[g]->stream);
    ncclResult_t r2 = rccl().GroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) rc = rccl_fail(r != ncclSuccess ? r : r2, "ncclAllGather");
    for (int g = 0; g < n && rc == CCMP_OK; g++) {
      DeviceGuard guard(c->ctxs[g]->device);
      if (c->ev[3 * g + 2]) (void)hipEventRecord(c->ev[3 * g + 2], c->ctxs[g]->stream);
    }
  }
  if (rc == CCMP_OK) rc = ccmp_host::for_each_shard(n, comm_phase3, &A);
  else (void)ccmp_host::for_each_shard(n, comm_sync_one, &A); // the error path still waits for every stream
  if (rc != CCMP_OK) return rc;
  for (int g = 0; g < n; g++) {
    c->kernel_ms[g] = c->gather_ms[g] = -1.0f;
    if (!c->ev[3 * g] || !c->ev[3 * g + 1] || !c->ev[3 * g + 2]) continue;
    DeviceGuard guard(c->ctxs[g]->device);
    (void)hipEventElapsedTime(&c->kernel_ms[g], c->ev[3 * g], c->ev[3 * g + 1]);
    (void)hipEventElapsedTime(&c->gather_ms[g], c->ev[3 * g + 1], c->ev[3 * g + 2]);
  }
  // unpack in rank order = global sample order for contiguous shards
  size_t total = 0;
  bool overflow = false;
  for (int g = 0; g < n; g++) {
    const double *blk = c->host_recv + (size_t)g * block_doubles;
    uint64_t cnt;
    memcpy(&cnt, blk, sizeof cnt);
    if (counts) counts[g] = cnt;
    if (cnt > cap) { overflow = true; continue; }
    if (total + cnt <= valid_capacity) memcpy(valid_out + total * 14, blk + 14, (size_t)cnt * 14 * sizeof(double));
    total += (size_t)cnt;
  }
  *n_valid = total;
  if (overflow || total > valid_capacity) return CCMP_EOVERFLOW; // counts[] tells the caller how much room is needed
  return CCMP_OK;
}

int ccmp_project_sharded(ccmp_comm *comm, const ccmp_problem *p, const double *q_in, size_t B, double *q_out, uint8_t *ok,
                         uint16_t *iters, size_t block_rows, double *valid_out, size_t valid_capacity, uint64_t *counts,
                         uint64_t *n_valid)
{
  return comm_project_common(comm, p, 0, q_in, 0, 0, B, q_out, ok, iters, block_rows, valid_out, valid_capacity, counts, n_valid);
}

int ccmp_sample_project_sharded(ccmp_comm *comm, const ccmp_problem *p, uint64_t seed, uint64_t first_index, size_t B, double *q_out,
                                uint8_t *ok, uint16_t *iters, size_t block_rows, double *valid_out, size_t valid_capacity,
                                uint64_t *counts, uint64_t *n_valid)
{
  return comm_project_common(comm, p, 1, nullptr, seed, first_index, B, q_out, ok, iters, block_rows, valid_out, valid_capacity, counts,
                             n_valid);
}

} // extern "C"
