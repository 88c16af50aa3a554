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

// scheduling defaults (round-2 sweeps with the faster throughput kernel, Wine_Bottle / stefan, in-process, ms):
//   B        latency kernel alone   throughput + hand-over at once   + hand-over below 80 % occupancy   scout + 80 %
//   8192     1.46 / 2.39            1.41 / 2.41                      1.59 / 2.64                        1.78 / 2.86
//   12288    1.96 / 3.14            1.91 / 3.14                      1.84 / 2.90                        1.93 / 3.06
//   16384    2.27 / 3.98            2.37 / 3.92                      2.06 / 3.39                        2.16 / 3.37
//   20480    3.21 / 4.80            3.03 / 4.77                      2.35 / 3.83                        2.34 / 3.68
//   28672    4.01 / 6.26            3.20 / 5.42                      3.02 / 4.93                        2.85 / 4.45
//   32768    4.24 / 7.10            3.44 / 5.69                      3.43 / 5.38                        3.09 / 4.88
// (from 49152 on the scout with immediate hand-over is best or equal; from 120000 on no hand-over at all)
// Up to here the latency kernel alone is quickest.  Round 4: it now runs in the FP32 scout's longest-predicted-first order from
// kDefaultLatencyOrderMin samples on (profiles/r04_latency_order_ab.log, mean of six seeds, index order | scout order, ms: Wine_Bottle
// 4096: 0.792 | 0.733, 8192: 1.252 | 1.063, 10240: 1.464 | 1.293; stefan 4096: 1.347 | 1.130, 8192: 2.034 | 1.650, 10240: 2.400 | 2.000;
// 3072: +0.5 % / -8 %), which moved its crossover with scout + throughput kernel + split launch upward
// (profiles/r04_small_batch_crossover.log, latency alone | split path, ms: Wine_Bottle 12288: 1.488 | 1.523, 14336: 1.694 | 1.608,
// 16384: 1.917 | 1.677; stefan 12288: 2.324 | 2.753, 14336: 2.663 | 2.887, 16384: 3.028 | 3.029, 20480: 3.706 | 3.302)
constexpr size_t kDefaultSmallBatch = 10240; // (14336 before the wide split launch: profiles/r04_hybrid_launch_sweep.log)
constexpr size_t kSplitWideMax = 24576;
constexpr size_t kGeoGroupHighCut = 20480;  // bulk extend calls: the low cut of the order is 40 rounds below this many edges, 48 from there on (ccmp_ctx: geodesic_group*)
constexpr int kGeoPoolDoubles = 40;  // = kGeoPoolEntry (ccmp_fd_common.h): one handed-over edge of the extend step's bulk form
constexpr int kGeoGroupWords = 8 + 64 + 4; // ctx->queue: first of the 8 words of the extend step's bulk form (behind the analytic kernels')     // split launch: up to here two latency blocks per CU (ccmp_ctx: fd_split*)
constexpr size_t kDefaultLatencyOrderMin = 3072;
constexpr size_t kDefaultLptMinBatch = 16384;  // from here on the scout pays (round 3: its predictions also sort the hand-over into two classes
                                               // and its sort lost 0.09 ms; Wine_Bottle / stefan, ms without | with: 16384: 1.99 | 1.92 / 3.16 | 3.15;
                                               // 20480: 2.37 | 2.16 / 3.60 | 3.43; 26624: 2.72 | 2.38 / 4.40 | 3.94; 14336: 1.85 | 1.87 / 2.91 | 2.98)
// smaller batches: keep the throughput kernel going while >= 70 % of its slots are busy.  Re-swept in round 4 with the split
// launch on (profiles/r04_handover_sweep_with_split.log; ms, 80 % below 40960 and at once above = the rule before | 70 %):
// Wine_Bottle 32768: 2.61 | 2.59   40960: 3.21 | 3.05   49152: 3.69 | 3.53   65536: 4.61 | 4.80 (stays "at once");
// stefan 16384 ... 32768: +-0.5 %   40960: 5.60 | 5.37   49152: 6.12 | 6.16   65536: 7.68 | 7.99
constexpr size_t kOccupancyHandoverBelow = 53248;
constexpr int kOccupancyHandoverValue = 80; // 10 + per cent

struct ccmp_ctx {
  int device = 0;
  int num_cus = 0;
  int waves_per_cu = 0;
  hipStream_t stream = nullptr;
  unsigned long long *queue = nullptr; // work-queue heads: words 0-2 projector kernels, 3-4 extend step, 5 scout, 8.. analytic kernels
  double *pool = nullptr;              // straggler hand-over records (group kernel -> wave kernel)
  size_t pool_cap = 0;                 // in records
  int wave_kernel = 1;                 // 0: group kernel only, 1: group + wave-per-sample (default), 2: wave only
  const unsigned int *order = nullptr; // experimental: externally supplied processing order
  int flat_kernel = 1;                 // latency work (small batches, hand-over): 1 = one-round 128-thread kernel, 0 = single-wave kernel
  int stock_kernels = 1;               // 0: always the general kernels, also for the stock Panda structure (tests, A/B)
  int lpt = 1;                         // 0: in-order; 1: FP32 scout + longest-predicted-first, hand-over kept; 2: same, no hand-over
  size_t lpt_min_batch = kDefaultLptMinBatch; // below this the scout costs more than the tail it removes
  void *lpt_buf = nullptr;             // pred (u16 x B) | hist (u32 x 1024) | order (u32 x B)
  size_t lpt_cap = 0;                  // in samples
  int analytic_cap = 96;               // analytic mode: samples past this many iterations go to the rows kernel (0 = never)
  size_t analytic_small_batch = 16384; // analytic mode: at or below, the rows kernel alone
  size_t analytic_handover_max = 131072; // analytic mode: hand-over for batches up to here (larger ones: one-lane kernel alone)
  int analytic_split = 1;                // analytic mode: scout order + six-lane kernel beside the one-lane kernel for large batches
  size_t analytic_split_min = 100000, analytic_split_max = 300000; // batch sizes of the split launch (sweep in ccmp_api.cpp)
  int analytic_split_front = 128;        // wavefronts of the six-lane kernel beside the one-lane kernel
  int analytic_split_cap = 160;          // split launch: mispredicted samples leave the one-lane kernel past this many iterations
  int analytic_split_pred = 90;          // predicted iterations from which a sample goes to the six-lane kernel
  hipStream_t side = nullptr;            // side stream of split launches
  hipEvent_t fork = nullptr, join = nullptr;
  int pool_long_remaining = 24;          // hand-over with scout predictions: samples with >= this many predicted iterations left go to the
                                         // front of the pool and are taken first by the latency kernel (0 = one class).  Wine_Bottle, ms,
                                         // one class | 16 | 24 | 32 | 48 | 64: 28672: 2.83 | 2.62 | 2.42 | 2.41 | 2.44 | 2.78; 32768: 3.18 |
                                         // 2.88 | 2.71 | 2.73 | 2.74 | 3.01; 40960: 3.44 | 3.31 | 3.28 | 3.27 | 3.38 | 3.47; >= 49152 and stefan: +-2 %
  // Split launch of mid-size reference-arithmetic batches (round 4): the front of the scout's descending order — samples
  // predicted >= fd_split_pred iterations, at most fd_split_front, one latency block each — runs on the side stream BESIDE the
  // throughput kernel from the start (which gives up fd_split_group_cut wavefronts per CU: a latency block needs two SIMDs
  // with a free register slot) instead of behind its hand-over.  Interleaved A/B, ms, off | on (profiles/r04_split_launch_ab2.log;
  // below 16384 "off" = the policy before: latency kernel alone / no scout):
  //   Wine_Bottle 10240: 1.638 | 1.418   12288: 1.789 | 1.502   16384: 1.92 | 1.686   24576: 2.29 | 2.066   32768: 2.655 | 2.607
  //               40960: 3.165 | 3.20    57344: 4.366 | 3.963   65536: 4.80 | 4.60    81920: 6.095 | 5.756   98304: 6.75 | 6.87
  //   stefan      10240: 2.397 | 2.60    12288: 2.724 | 2.74    16384: 3.05 | 3.01    32768: 4.65 | 4.67     65536: 7.47 | 7.66   81920: 9.37 | 9.27
  // (stefan / dumbbell: thousands of samples never converge — the front is full of them whatever it takes; neutral.)  A cut of
  // one wavefront per CU leaves the blocks nowhere to go (+5 % from 28672 on); 128 instead of 256 blocks: +1 %.
  // Late round 4: the front's blocks go on with the next-longest samples (fd_split_samples > blocks), and its shape follows the
  // batch (-1 = by the rule below; profiles/r04_hybrid_launch_sweep.log, ms, old shape | new):
  //   up to kSplitWideMax samples the throughput kernel fills half the chip or less and latency blocks fit beside it without
  //   displacing anything: 2 blocks per CU, samples predicted >= 40, up to 4 samples per CU, 3 wavefronts per CU left out —
  //   Wine_Bottle 12288: 1.443 (latency kernel alone) | 1.388   14336: 1.660 | 1.484   16384: 1.63 | 1.57   20480: 1.804 | 1.771
  //   stefan      12288: 2.290 | 2.106   14336: 2.635 | 2.293   16384: 2.98 | 2.47   20480: 3.226 | 3.147   24576: 3.56 | 3.54
  //   above: 1 block per CU, predicted >= 56, 2 wavefronts left out as before, up to 3 (from 40960: 4) samples per CU —
  //   Wine_Bottle 28672: 2.273 | 2.227   32768: 2.535 | 2.493   49152: 3.525 | 3.424   65536: 4.591 | 4.453   stefan 65536: 7.644 | 7.495
  int fd_split = 1;
  size_t fd_split_min = 0, fd_split_max = 90112; // (batches <= small_batch never get here)
  int fd_split_pred = -1;                // predicted iterations from which a sample goes to the front ...
  int fd_split_front = -1;               // ... on this many latency blocks ...
  long long fd_split_samples = -1;       // ... at most this many samples (0 = one per block); more: the blocks go on with the next-longest
  int fd_split_group_cut = -1;           // throughput wavefronts per CU the split launch leaves out
  // Bulk extend calls (round budget + scout order, geodesic_group_min edges or more): the edges predicted shorter than the cut run on
  // geodesic_group_kernel (ten edges per wavefront, ccmp_kernels_fd.hip: ~1 250 instead of ~2 700 wave-instructions per round and
  // edge, at twelve times the latency per round), the others on geodesic_flat_kernel blocks on the side stream, both from the start;
  // once the ticket queue is dry and the group kernel's live edges fill less than geodesic_group_handover_pct % of its slots it hands
  // them — in the middle of their projections — to latency blocks launched behind it.  tools/geo_group_ab.py, lists of 16 + 128
  // rounds, ms, latency kernel alone | hybrid (profiles/r04_bulk_extend_ab.log, calls 10-13):
  //   Wine_Bottle 16384: 1.36 | 1.29   24576: 1.97 | 1.60   32768: 2.57 | 1.91   65536: 5.00 | 3.49   131072: 9.86 | 6.69
  //   stefan      16384: 1.93 | 1.70   24576: 2.78 | 2.26   32768: 3.65 | 2.98   65536: 7.12 | 5.57   131072: 14.1 | 10.8
  //   12288 and below: slower (Wine_Bottle +5..+40 %).  Without the hand-over the group kernel's longest TRUE edge sets a floor of
  //   2-3 ms under a call and the hybrid only paid from 32768 edges (calls 1-9).
  int geodesic_group = 1;
  size_t geodesic_group_min = 16384;
  int geodesic_group_pred = -1;          // cut of the order: edges predicted this many rounds or more go to the latency blocks (-1: by the batch,
                                         // see geodesic_group_low_cut) ...
  int geodesic_group_low_cut = -1, geodesic_group_heavy_permille = 100; // pred = -1: cut at the scout's cap (64) if the edges beyond it carry
                                         // this share of the predicted work, else at the low cut (-1: 40 below kGeoGroupHighCut edges, 48 from there on)
  int geodesic_group_permille = 0;       // ... or, > 0: the largest cut <= geodesic_group_pred whose front carries this share of the predicted work
  int geodesic_group_front_per_cu = 8;   // latency blocks per CU for the front
  int geodesic_group_handover_pct = 50;  // hand the group kernel's live edges to latency blocks once the queue is dry and they fill less than this share of its slots (0 = never)
  double *geo_pool = nullptr;            // ... through this pool (kGeoPoolDoubles per edge)
  size_t geo_pool_cap = 0;
  int geodesic_group_waves_per_cu = 8;   // group kernel's wavefronts per CU at most (10 fit by LDS; the front's blocks need room)
  size_t latency_order_min = kDefaultLatencyOrderMin; // latency kernel alone (batches <= small_batch): FP32 scout order from this many samples on
  // FP32 scouts on LANE PAIRS (round 4, ccmp_kernels_scout.hip): the even lane takes arm 0, the odd lane arm 1 — half the chain
  // work per lane and round, and the scout's run time is its longest lane's.  Same predictions (equal to the one-lane scout's on
  // 84 % of samples, within 1 on 91 %; both equal the true count on 84 %).  profiles/r04_scout_pairs_ab.log, one lane | pairs, ms:
  // Wine_Bottle 4096: 0.698 | 0.656   8192: 1.064 | 1.024   16384: 1.681 | 1.638   32768: 2.580 | 2.524   65536: 4.60 | 4.61;
  // extend step 8192 edges: 0.88 | 0.84   16384: 1.411 | 1.360   65536: 5.00 | 4.96; stefan -1 ... -4 % / -3 ... -6 %.
  int scout_pairs = 1;
  int scout_pair_blocks_per_cu = 1;      // ... projector scout: while every sample gets its pair at once (128 x this x CUs samples = 32768)
  size_t scout_pair_max_edges = 131072;  // ... extend-step scout: up to this many edges
  int latency_blocks_per_cu = 8;         // persistent 128-thread blocks of the projector's latency kernel per CU (8 resident: 128 registers)
  int geodesic_blocks_per_cu = 4;        // ... of the extend step's latency flavour (ccmp_kernels_geo.hip: 256-register budget, 4 resident)
  int geodesic_flavour = 0;              // extend step build: 0 = by call shape (round budget and size), 1 = throughput, 2 = latency
  int geodesic_order = 2;                // extend step, batches beyond the resident blocks: 1 = far-apart edges first, 2 = FP32 scout order
  size_t geodesic_scout_min = 6144;      // ... the scout from this many edges on (below: the two-class order by distance)
  int geodesic_scout_rounds = 64;        // ... the scout stops an edge after this many Newton rounds (all such edges are "long");
                                         // 16384 near-neighbour edges, lists of 16, ms: index order 2.90, far-apart first 2.43,
                                         // scout capped at 32 / 48 / 64 / 96 rounds 2.40 / 2.54 / 2.18 / 2.25 (8192 edges: 1.20 / 1.14 / 0.98 at 64)
  size_t geodesic_order_min = 4096;      // ... from this many edges on (the ordering pass is one more launch)
  double geodesic_long_steps = 12.0;     // ... "long" = further than this many delta apart (median near-neighbour edge: 4)
  size_t clearance_per_state_max = 8192; // proxy clearance: up to here one block per state, above 64-state tiles
  int dump_threshold = -1;             // hand a wave's samples over once the queue is dry and <= this many groups are busy; -1 = auto
  size_t small_batch = kDefaultSmallBatch;    // at or below: latency kernel on everything
  unsigned int *scan = nullptr;        // compaction block counts
  size_t scan_cap = 0;
  // staging for the *_host conveniences
  void *stage = nullptr;
  size_t stage_cap = 0;
  void *pin = nullptr;     // pinned, device-mapped host block for small *_host calls (single states of the reference signature)
  void *pin_dev = nullptr; // the same block as the kernels see it
  // completion word of single-state calls (last 64 bytes of the pinned block): the latency kernel publishes done_seq
  // behind its results and the host polls it instead of waiting for the stream's completion signal
  unsigned int done_seq = 0;
  bool done_armed = false; // set by project_common when the launch it made will publish done_seq
  bool want_done = false;  // set by the host entry point that is going to poll
  // *_host calls on a caller's PAGE-LOCKED buffers (hipHostMalloc / hipHostRegister; found with hipPointerGetAttributes):
  // 0 = stage them like pageable memory, 1 = q_in is uploaded by one asynchronous copy and the kernels write q_out straight
  // into the caller's buffer, 2 = the kernels also read q_in from it (nothing is staged but the flags; default).  A projection
  // reads and writes each 112-byte row once: 3.7 GB/s at 16 M projections/s, a fraction of what the link carries.  C3 batch
  // through ccmp_project_host, ms (kernels alone 16.24): pageable 18.71 | pinned staged 18.04 | 1: 18.40 | 2: 17.50.
  int host_zero_copy = 2;
  // ccmp_*_sharded_host / ccmp_*_sharded: when this context's shard had its upload behind it (host clock, ms from the
  // call's entry) and the event recorded on its stream at that point (ccmp_sharded_host_last_timing)
  double shard_launch_ms = -1.0;
  hipEvent_t ev_shard = nullptr;
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
