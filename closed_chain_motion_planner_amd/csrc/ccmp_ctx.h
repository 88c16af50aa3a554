/* ccmp_ctx.h — internals shared by the host translation units of libccmp (ccmp_api.cpp, ccmp_host_io.cpp, ccmp_comm.cpp); not part of
 * the public ABI. */
#ifndef CCMP_CTX_H
#define CCMP_CTX_H
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>

#include "../../include/ccmp.h"

namespace ccmp_host {

/* text of the last HIP / RCCL failure on this thread (ccmp_last_hip_error) */
extern thread_local char g_hip_err[256];

inline int hip_fail(hipError_t e, const char *what)
{
  snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
  return CCMP_EHIP;
}
#define HIP_TRY(call)                                            \
  do {                                                           \
    hipError_t e_ = (call);                                      \
    if (e_ != hipSuccess) return ccmp_host::hip_fail(e_, #call); \
  } while (0)

/* makes `dev` current for the lifetime of the guard and restores the previous device */
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev)
  {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard()
  {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace ccmp_host

// Scheduling constants.  Every value below was chosen from an interleaved A/B on the device; the tables behind them live in
// DESIGN_experiments.md §11 (they were comments here until round 5), the regimes they delimit are what ccmp_ctx_describe() prints,
// and tools/policy_check.py re-times each boundary on whatever box it runs on.
constexpr size_t kDefaultSmallBatch = 10240;       // at or below: the latency kernel alone
constexpr size_t kSplitWideMax = 24576;            // split launch: up to here two latency blocks per CU for the front (fd_split*)
constexpr size_t kGeoGroupHighCut = 20480;         // bulk extend calls: the low cut of the order is 40 rounds below this many edges, 48 from here on ...
constexpr size_t kGeoGroupLateHandoverFrom = 32768; // bulk extend calls: the group kernel hands over below 50 % occupancy up to here, below 80 % from here on (profiles/r05_bulk_handover_sweep.log)
constexpr size_t kGeoGroupHigherCut = 65536;        // ... and 56 from here on (profiles/r05_low_cut_sweep.log: 65 536 edges -1.7 %, 131 072 -2.1 % against 48)
constexpr int kGeoPoolDoubles = 40;                // = kGeoPoolEntry (ccmp_fd_common.h): one handed-over edge of the extend step's bulk form
constexpr int kAnalyticWords = 64 + 2;            // ctx->queue + 8: 64 ticket words of the analytic mode's lane-pair kernel, the hand-over pool's fill count, the latency kernel's ticket word
constexpr int kGeoGroupWords = 8 + kAnalyticWords + 1; // ctx->queue: first of the 8 words of the extend step's bulk form (behind the analytic kernel's)
constexpr size_t kDefaultLatencyOrderMin = 2049;   // latency kernel alone: FP32 scout order as soon as the blocks take tickets (more samples than blocks)
constexpr size_t kDefaultLptMinBatch = 16384;      // throughput kernel: the scout's order pays from here on
constexpr size_t kOccupancyHandoverBelow = 53248;  // below: the throughput kernel hands over by occupancy, from here on at once
constexpr int kOccupancyHandoverValue = 10 + 70;   // "handover_threshold" encoding of that rule: 10 + per cent of the group slots (70 %)
constexpr size_t kNoHandoverFrom = 131072;         // ordered batches of this size or more end on their shortest samples: no hand-over

// One execution context (include/ccmp.h).  The tuning members are reached by name through the option table of ccmp_policy.cpp
// (ccmp_ctx_set_option / ccmp_ctx_get_option / ccmp_ctx_option_info), which also holds their ranges and one-line meanings; their
// initialisers here ARE the built-in defaults the table reports.
struct ccmp_ctx {
  int device = 0;
  int num_cus = 0;
  hipStream_t stream = nullptr;        // internal stream of the *_host entry points
  hipStream_t side = nullptr;          // side stream of split launches (highest priority: a hardware queue of its own)
  hipEvent_t fork = nullptr, join = nullptr;
  unsigned long long *queue = nullptr; // work-queue words: 0-2, 6-7 projector kernels, 3-4 extend step, 5 scout, 8.. analytic kernels, kGeoGroupWords.. bulk extend
  double *pool = nullptr;              // straggler hand-over records (throughput kernel -> latency kernel)
  size_t pool_cap = 0;                 // in records
  void *lpt_buf = nullptr;             // pred (u16 x B) | hist (u32 x 1024) | order (u32 x B) | flags (u8 x B, bulk checkMotion)
  size_t lpt_cap = 0;                  // in samples
  double *geo_pool = nullptr;          // bulk extend hand-over: kGeoPoolDoubles per edge
  size_t geo_pool_cap = 0;
  void *geo_an = nullptr;              // the analytic-mode extend step's per-edge state (ccmp_kernels_fast.hip: geo_an_ws) + target flags
  size_t geo_an_cap = 0;               // in edges
  unsigned int *scan = nullptr;        // compaction block counts
  size_t scan_cap = 0;
  void *stage = nullptr;               // device staging of the *_host conveniences
  size_t stage_cap = 0;
  void *pin = nullptr;                 // pinned, device-mapped host block for small *_host calls (single states of the reference signature)
  void *pin_dev = nullptr;             // the same block as the kernels see it
  // completion word of single-state calls (last 64 bytes of the pinned block): the kernel publishes done_seq behind its results
  // and the host polls it instead of waiting for the stream's completion signal
  unsigned int done_seq = 0;
  bool done_armed = false;             // set by the launch path when the launch it made will publish done_seq
  bool want_done = false;              // set by the host entry point that is going to poll
  const unsigned int *order = nullptr; // experimental: externally supplied processing order
  double shard_launch_ms = -1.0;       // ccmp_*_sharded*: when this context's shard had its upload behind it (host clock, ms from entry) ...
  hipEvent_t ev_shard = nullptr;       // ... and the event recorded on its stream at that point
  // resident service kernel (opt-in, option "resident"; ccmp_resident.h)
  struct ccmp_resident *resident = nullptr;
  int resident_on = 0;
  int resident_gave_up = 0;            // how often a start of the service kernel gave up (it did not get to run within 5 ms: that ONE call took the launch path; ccmp_resident.cpp)

  // ---- tuning (option table: ccmp_policy.cpp) ----------------------------------------------------------------------------
  int waves_per_cu = 0;                // persistent wavefronts of the throughput kernels per CU (0 = 12)
  int wave_kernel = 1;                 // "hand_over": 0 throughput kernel only, 1 + latency kernel for stragglers, 2 latency kernel only
  size_t small_batch = kDefaultSmallBatch;
  int flat_kernel = 1;
  int stock_kernels = 1;
  int lpt = 1;
  size_t lpt_min_batch = kDefaultLptMinBatch;
  size_t latency_order_min = kDefaultLatencyOrderMin;
  int dump_threshold = -1;             // "handover_threshold"
  int pool_long_remaining = 24;
  int latency_blocks_per_cu = 8;
  int fd_split = 1;
  size_t fd_split_min = 0, fd_split_max = 90112;
  int fd_split_pred = -1, fd_split_front = -1, fd_split_group_cut = -1;
  long long fd_split_samples = -1;
  size_t analytic_small_batch = 8192;  // analytic mode: at or below, the sixteen-lanes-per-sample latency kernel alone
  int analytic_waves_per_cu = 12;      // ... persistent wavefronts of the lane-pair kernel per CU (142 registers, 13 KB of LDS: three per SIMD)
  int analytic_handover = 8;           // ... one of them hands over to the latency kernel once its tickets are gone and it holds at most this many samples (0: never)
  int scout_pairs = 1, scout_pair_blocks_per_cu = 1;
  size_t scout_pair_max_edges = 131072;
  int geodesic_blocks_per_cu = 4, geodesic_flavour = 0, geodesic_order = 2;
  size_t geodesic_order_min = 2049, geodesic_scout_min = 2049; // (4096 / 6144 until round 5: the scout's order pays as soon as the blocks take tickets — tools/policy_check.py)
  int geodesic_scout_rounds = 64;
  double geodesic_long_steps = 12.0;
  int geodesic_group = 1;
  size_t geodesic_group_min = 13312;  // (16384 until round 5: tools/policy_check.py found the bulk form 10-12 % ahead at 15872 edges; crossover at 13312, profiles/r05_bulk_crossover.log)
  int geodesic_group_pred = -1, geodesic_group_low_cut = -1, geodesic_group_heavy_permille = 100, geodesic_group_permille = 0;
  int geodesic_group_front_per_cu = 8, geodesic_group_waves_per_cu = 8, geodesic_group_handover_pct = -1;
  size_t clearance_per_state_max = 8192;
  int host_zero_copy = 2;
  int resident_idle_ms = 10;           // the resident service kernel leaves by itself after this long without a request
  int fail_after_fork = 0;             // debug build only (include/ccmp_debug.h): 1 / 2 = the split launches report a failure in front of / behind their side-stream part
  int debug_fail_calls = 0;            // debug build only: the next so many compute entry points return CCMP_EHIP before anything is launched
};

namespace ccmp_host {
/* the pinned, device-mapped block of the small *_host calls: its last 64 bytes hold the completion word */
constexpr size_t kPinBytes = 64 * 1024;
constexpr size_t kPinData = kPinBytes - 64;
/* device staging of the *_host conveniences, grown on demand */
int ensure_stage(ccmp_ctx *ctx, size_t bytes);
/* the device-visible alias of a caller's host range if all of it is page-locked and mapped (hipHostMalloc,
 * hipHostRegister), else nullptr; the calling thread's current device is the one that will use it */
void *pinned_alias(const void *host, size_t bytes);
/* runs fn(g) for g = 0..n-1, each on its own short-lived thread when n > 1 (one planner process driving several GPUs: a
 * copy from or to pageable memory blocks its caller, so shards issued from ONE thread start one after the other);
 * returns the first non-zero result and leaves that thread's error text in this thread's g_hip_err */
int for_each_shard(int n, int (*fn)(int g, void *arg), void *arg);
}  // namespace ccmp_host

#endif /* CCMP_CTX_H */
