// ccmp_policy.cpp — the tuning options of a context (one table: names, ranges, built-in defaults, meanings — behind
// ccmp_ctx_set_option / ccmp_ctx_get_option / ccmp_ctx_option_info) and the scheduling policy (which kernels a call of a given
// size runs on — behind the launches of ccmp_api.cpp and behind ccmp_ctx_describe).  No option and no plan changes a result bit.
#include "ccmp_policy.h"

#include <climits>
#include <cstdarg>
#include <cstring>

using ccmp_host::DeviceGuard;

namespace {

enum Kind { kInt, kSize, kLongLong, kDouble };
enum : unsigned { kNotZero = 1u, kDebug = 2u };

struct OptionDesc {
  const char *name;
  size_t offset;
  Kind kind;
  long lo, hi;
  unsigned flags; // kNotZero: 0 is not a value (lo is -1 = "by the call's size"); kDebug: not a tuning knob
  const char *doc;
};

#define OPT(name, member, kind, lo, hi, flags, doc) {name, offsetof(ccmp_ctx, member), kind, lo, hi, flags, doc}
// The order is the order of the table in include/ccmp.h (tests/test_host_cabi.py compares the two).
// Round 6 removed seventeen options whose sweeps had one setting winning everywhere (DESIGN_experiments.md §11, §13): their members
// of ccmp_ctx keep the winning value as a constant of the build.
const OptionDesc kOptions[] = {
    // projector, reference arithmetic
    OPT("hand_over", wave_kernel, kInt, 0, 2, 0, "0 = throughput kernel (10 samples per wavefront) only, 1 = that kernel until its queue drains, then the latency kernel on what is in flight, 2 = latency kernel only (= ccmp_ctx_set_schedule)"),
    OPT("small_batch", small_batch, kSize, 0, LONG_MAX, 0, "batches of at most this many samples run on the latency kernel alone (= ccmp_ctx_set_schedule)"),
    OPT("waves_per_cu", waves_per_cu, kInt, 0, 32, 0, "persistent wavefronts of the throughput kernels per CU, 0 = 12 (= ccmp_ctx_set_waves_per_cu)"),
    OPT("lpt", lpt, kInt, 0, 2, 0, "0 = index order, 1 = FP32 scout + longest-predicted-first, hand-over kept below 131072 samples, 2 = the same without hand-over (= ccmp_ctx_set_lpt)"),
    OPT("lpt_min_batch", lpt_min_batch, kSize, 0, LONG_MAX, 0, "the scout's order from this many samples on (= ccmp_ctx_set_lpt)"),
    OPT("latency_order_min", latency_order_min, kSize, 0, LONG_MAX, 0, "latency kernel alone: tickets through the scout's order from this many samples on"),
    OPT("flat_kernel", flat_kernel, kInt, 0, 1, 0, "latency work: 1 = one sample per 128-thread block, an iteration's evaluations in one round, 0 = one wavefront per sample"),
    OPT("stock_kernels", stock_kernels, kInt, 0, 1, 0, "1 = kernels that skip the exact zeros of the uncalibrated Panda when both arms carry them (same bits), 0 = always the general kernels"),
    OPT("handover_threshold", dump_threshold, kInt, -1, 110, 0, "-1 = automatic; 0..10: a wavefront hands over once the queue is dry and at most this many of its 10 groups are busy; 11..110: all hand over once the samples in flight fill less than (value - 10) % of the group slots"),
    OPT("fd_split", fd_split, kInt, 0, 1, 0, "1 = split launch: above small_batch, the predicted-longest samples run on latency blocks on a side stream beside the throughput kernel"),
    OPT("fd_split_min", fd_split_min, kSize, 0, LONG_MAX, 0, "split launch from this many samples ..."),
    OPT("fd_split_max", fd_split_max, kSize, 0, LONG_MAX, 0, "... up to this many"),
    OPT("fd_split_pred", fd_split_pred, kInt, -1, 1023, kNotZero, "predicted iterations from which a sample belongs to the front (-1: 40 up to 24576 samples, 56 above)"),
    OPT("fd_split_front", fd_split_front, kInt, -1, 4096, 0, "latency blocks of the front (-1: two per CU up to 24576 samples, one above; 0 = no split)"),
    OPT("fd_split_samples", fd_split_samples, kLongLong, -1, 0x7fffffffl, 0, "samples of the front at most (-1: four per CU up to 24576 samples, three to four above; 0 = one per block)"),
    OPT("fd_split_group_cut", fd_split_group_cut, kInt, -1, 8, 0, "throughput wavefronts per CU left out for the front's blocks (-1: 3 up to 24576 samples, 2 above)"),
    // projector, analytic mode
    OPT("analytic_small_batch", analytic_small_batch, kSize, 0, LONG_MAX, 0, "analytic mode: at or below, the sixteen-lanes-per-sample latency kernel alone"),
    OPT("analytic_waves_per_cu", analytic_waves_per_cu, kInt, 1, 12, 0, "analytic mode: persistent wavefronts of the lane-pair kernel per CU"),
    OPT("analytic_handover", analytic_handover, kInt, 0, 32, 0, "analytic mode: a wavefront of the lane-pair kernel whose tickets are gone hands over to the latency kernel once it holds at most this many samples (0 = never: one launch)"),
    // FP32 scouts
    OPT("scout_pairs", scout_pairs, kInt, 0, 1, 0, "1 = two lanes per sample / edge, one arm each, where lanes are plentiful (stock twin arms)"),
    // extend step
    OPT("geodesic_flavour", geodesic_flavour, kInt, 0, 2, 0, "two builds, same bits: 0 = throughput build for calls with a round budget beyond the latency build's blocks, latency build otherwise; 1 / 2 = always the throughput / latency build"),
    OPT("geodesic_order", geodesic_order, kInt, 0, 2, 0, "batches beyond the resident blocks: 0 = index order, 1 = far-apart edges first, 2 = FP32 scout order from geodesic_scout_min edges on"),
    OPT("geodesic_order_min", geodesic_order_min, kSize, 0, LONG_MAX, 0, "no ordering pass below this many edges"),
    OPT("geodesic_long_steps", geodesic_long_steps, kDouble, 0, LONG_MAX, 0, "order 1: edges further apart than this many delta count as long"),
    OPT("geodesic_scout_min", geodesic_scout_min, kSize, 0, LONG_MAX, 0, "the scout from this many edges on"),
    OPT("geodesic_scout_rounds", geodesic_scout_rounds, kInt, 1, 1023, 0, "the scout stops an edge after this many Newton rounds"),
    OPT("geodesic_group", geodesic_group, kInt, 0, 1, 0, "1 = bulk calls (round budget, scout order): short edges ten to a wavefront on the throughput layout, the front of the order on latency blocks beside them"),
    OPT("geodesic_group_min", geodesic_group_min, kSize, 0, LONG_MAX, 0, "... from this many edges"),
    OPT("geodesic_group_pred", geodesic_group_pred, kInt, -1, 1023, kNotZero, "... cut of the order in predicted rounds (-1: the scout's cap where the edges beyond it carry a tenth of the predicted work, else geodesic_group_low_cut)"),
    OPT("geodesic_group_low_cut", geodesic_group_low_cut, kInt, -1, 64, kNotZero, "... (-1: 40 below 20480 edges, 48 from there on, 56 from 65536)"),
    OPT("geodesic_group_permille", geodesic_group_permille, kInt, 0, 1000, 0, "... > 0: instead, the largest cut whose front carries this share of the predicted work"),
    OPT("geodesic_group_handover_pct", geodesic_group_handover_pct, kInt, -1, 100, 0, "... with the queue dry, every wavefront gives its edges to latency blocks once those in flight fill less than this share of the slots (0 = never; -1: 50 below 32768 edges, 80 from there on)"),
    // other
    OPT("clearance_per_state_max", clearance_per_state_max, kSize, 0, LONG_MAX, 0, "proxy clearance: one block per state up to this many states, 64-state tiles above"),
    OPT("host_zero_copy", host_zero_copy, kInt, 0, 2, 0, "*_host calls on page-locked caller buffers: 0 = staged, 1 = q_out written in place, 2 = q_in read in place too"),
    OPT("resident_idle_ms", resident_idle_ms, kInt, 1, 10000, 0, "the resident service kernel (option \"resident\") leaves by itself after this many milliseconds without a request"),
#ifdef CCMP_DEBUG_HOOKS // lib/libccmp_debug.so only (include/ccmp_debug.h)
    OPT("fail_after_fork", fail_after_fork, kInt, 0, 2, kDebug, "debug: the split launches report a failure in front of (1) / behind (2) their side-stream part"),
#endif
};
#undef OPT
constexpr int kNumOptions = (int)(sizeof kOptions / sizeof kOptions[0]);

const OptionDesc *find_option(const char *name)
{
  for (int i = 0; i < kNumOptions; i++)
    if (!strcmp(kOptions[i].name, name)) return &kOptions[i];
  return nullptr;
}

long read_option(const ccmp_ctx *ctx, const OptionDesc &o)
{
  const char *base = reinterpret_cast<const char *>(ctx) + o.offset;
  switch (o.kind) {
    case kInt: return *reinterpret_cast<const int *>(base);
    case kSize: {
      const size_t v = *reinterpret_cast<const size_t *>(base);
      return v > (size_t)LONG_MAX ? LONG_MAX : (long)v;
    }
    case kLongLong: return (long)*reinterpret_cast<const long long *>(base);
    case kDouble: return (long)*reinterpret_cast<const double *>(base);
  }
  return 0;
}

void write_option(ccmp_ctx *ctx, const OptionDesc &o, long v)
{
  char *base = reinterpret_cast<char *>(ctx) + o.offset;
  switch (o.kind) {
    case kInt: *reinterpret_cast<int *>(base) = (int)v; break;
    case kSize: *reinterpret_cast<size_t *>(base) = (size_t)v; break;
    case kLongLong: *reinterpret_cast<long long *>(base) = v; break;
    case kDouble: *reinterpret_cast<double *>(base) = (double)v; break;
  }
}

}  // namespace

namespace ccmp_host {

const ccmp_ctx &default_ctx()
{
  static const ccmp_ctx d = [] {
    ccmp_ctx c;
    c.num_cus = 256; // MI355X
    return c;
  }();
  return d;
}

static int projector_blocks(const ccmp_ctx *ctx, size_t B, int samples_per_wave, int default_wpc)
{
  const int wpc = ctx->waves_per_cu > 0 ? ctx->waves_per_cu : default_wpc;
  size_t want = (B + samples_per_wave - 1) / samples_per_wave;
  const size_t cap = (size_t)ctx->num_cus * (size_t)wpc;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

// shape of the split launch's front for a batch of B samples (an option that was set wins)
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

// Which kernels a reference-arithmetic batch of B samples runs on.
//   latency kernel alone            B <= small_batch (or hand_over 2): one sample per block, lowest latency per sample; from
//                                   latency_order_min samples on through the FP32 scout's order
//   throughput kernel (+ hand-over) otherwise: 10 samples per wavefront from a queue; when the queue runs dry the samples
//                                   still in flight go to the latency kernel
//   scout + longest-first order     from lpt_min_batch on (and wherever the split launch applies); from kNoHandoverFrom samples
//                                   on without hand-over
//   split launch                    small_batch < B <= fd_split_max: the front of the order on latency blocks beside it
FdPlan plan_fd_batch(const ccmp_ctx *ctx, size_t B, bool external_order)
{
  FdPlan pl;
  const int wpc = ctx->waves_per_cu > 0 ? ctx->waves_per_cu : 12;
  // latency kernels: the flat kernel runs 8 blocks of 128 threads per CU (16 waves), the one-wave kernel wpc waves
  const size_t lat_cap = ctx->flat_kernel ? (size_t)ctx->num_cus * (size_t)ctx->latency_blocks_per_cu : (size_t)ctx->num_cus * (size_t)wpc;
  const bool latency_only = ctx->wave_kernel == 2 || (ctx->wave_kernel == 1 && B <= ctx->small_batch);
  if (latency_only) {
    pl.latency_blocks = (int)(B < lat_cap ? B : lat_cap);
    pl.latency_static = ctx->flat_kernel && B <= lat_cap;
    // a batch of a few fills of the latency kernel's blocks ends on the serial chain of whichever long sample the index order
    // happened to start late; the scout's order starts them first
    pl.latency_order = ctx->flat_kernel && !pl.latency_static && ctx->lpt > 0 && !external_order && B >= ctx->latency_order_min && B < 0xffffffffull;
    return pl;
  }
  pl.group_blocks = projector_blocks(ctx, B, 10, 12);
  pl.handover = ctx->wave_kernel == 1;
  // the scout pays from lpt_min_batch samples on by its order alone, and earlier where the split launch uses its predictions
  // to start the longest samples on latency blocks at once
  const bool split_range = ctx->fd_split && ctx->flat_kernel && ctx->wave_kernel == 1 && B >= ctx->fd_split_min && B <= ctx->fd_split_max;
  pl.scout = !external_order && ctx->lpt > 0 && (B >= ctx->lpt_min_batch || split_range) && B < 0xffffffffull;
  // Large ordered batches end on their shortest samples, and the scout is accurate there: nothing worth handing over is left
  // (-3 % at 262144 Wine_Bottle without it).  An explicit threshold keeps the hand-over.
  if (pl.scout && (ctx->lpt == 2 || (ctx->dump_threshold < 0 && B >= kNoHandoverFrom))) pl.handover = false;
  // the samples still in flight go to the latency kernel: at once for large batches; for batches of about one fill of the
  // throughput kernel only when the samples in flight no longer fill 70 % of its group slots
  pl.dump_threshold = ctx->dump_threshold >= 0 ? ctx->dump_threshold : (B < kOccupancyHandoverBelow ? kOccupancyHandoverValue : 10);
  if (pl.handover) {
    const size_t in_flight = (size_t)pl.group_blocks * 10;
    pl.latency_blocks = (int)(in_flight < lat_cap ? in_flight : lat_cap);
  }
  pl.two_class_pool = pl.scout && pl.handover && ctx->flat_kernel && ctx->pool_long_remaining > 0;
  // Split launch.  Decided HERE, before anything of the call is launched: it needs the scout's predictions, the hand-over's
  // latency kernel, a front of at least one block, and at least one throughput wavefront per CU left beside it (a cut that
  // reaches waves_per_cu — both are options — once left a grid of 0 or fewer blocks behind a front that was already running).
  if (pl.scout && split_range && pl.handover) {
    const SplitShape sh = split_shape(ctx, B);
    if (sh.blocks > 0 && wpc - sh.cut >= 1) {
      pl.split = true;
      pl.shape = sh;
      const int room = ctx->num_cus * (wpc - sh.cut);
      if (pl.group_blocks > room) pl.group_blocks = room;
    }
  }
  return pl;
}

// Analytic mode (ccmp_kernels_fast.hip).  Throughput: one sample per lane pair, 32 per wavefront, three wavefronts per SIMD.  A
// wavefront alone on its SIMD pays ~4.6 cycles per instruction whatever it holds, so a Newton round there lasts 4 us and a call
// would end on wavefronts that hold one or two long samples each: once the tickets are gone a wavefront that holds at most
// analytic_handover samples hands them over (x, index, counters) to the sixteen-lanes-per-sample latency kernel launched behind
// (2.7 us per round, four samples per wavefront), which also takes small batches alone.  (Round 6 measured and dropped: a cap on
// the iterations a sample does in the lane-pair kernel, a second lane-pair generation in between — DESIGN_experiments.md §13.)
AnalyticPlan plan_analytic_batch(const ccmp_ctx *ctx, size_t B)
{
  AnalyticPlan pl;
  const size_t lat_resident = (size_t)ctx->num_cus * 8; // the latency kernel: two wavefronts per SIMD, four samples per wavefront
  auto lat_blocks = [&](size_t samples) {
    const size_t w = (samples + 3) / 4;
    return (int)(w < lat_resident ? (w < 1 ? 1 : w) : lat_resident); // (surplus wavefronts read the pool's fill count and exit at once)
  };
  if (B <= ctx->analytic_small_batch) { // the latency kernel alone
    pl.latency_blocks = lat_blocks(B);
    return pl;
  }
  const size_t resident = (size_t)ctx->num_cus * (size_t)ctx->analytic_waves_per_cu, want = (B + 31) / 32;
  pl.pair_blocks = (int)(want < resident ? want : resident);
  if (ctx->analytic_handover > 0) {
    pl.dump = ctx->analytic_handover;
    pl.pool_records = (size_t)pl.pair_blocks * (size_t)pl.dump;
    pl.latency_blocks = lat_blocks(pl.pool_records);
  }
  return pl;
}

// The extend step.  One 128-thread block per edge.  Up to the resident capacity every edge has its block at once and the
// hardware dispatcher is the queue.  Beyond it the blocks are persistent and take tickets from an atomic word, handed out
// through a long-edges-first order when the batch is large enough for the ordering pass to pay.  Two builds of the kernel: a
// call that bounds the rounds per edge is bound by the chip's turnover of Newton rounds and takes the throughput flavour
// (8 blocks per CU); a call that ends on one edge's serial chain — no round budget, or no more edges than the latency flavour
// has blocks — takes the latency flavour (4 blocks per CU, fewer instructions per round).  Bulk calls (round budget, scout
// order, geodesic_group_min edges or more): the short edges on geodesic_group_kernel, the front on latency blocks beside it.
GeoPlan plan_geodesic(const ccmp_ctx *ctx, size_t E, int round_budget, bool continuation)
{
  GeoPlan pl;
  const size_t lat_resident = (size_t)ctx->num_cus * (size_t)ctx->geodesic_blocks_per_cu;
  pl.latency_flavour = ctx->geodesic_flavour == 2 || (ctx->geodesic_flavour == 0 && (round_budget == 0 || E <= lat_resident));
  const size_t resident = pl.latency_flavour ? lat_resident : (size_t)ctx->num_cus * (size_t)ctx->latency_blocks_per_cu;
  pl.blocks = E;
  if (E > resident) {
    pl.blocks = resident;
    pl.queued = true;
    pl.ordered = ctx->geodesic_order && E >= ctx->geodesic_order_min && E < 0xffffffffull;
    pl.scouted = pl.ordered && ctx->geodesic_order == 2 && E >= ctx->geodesic_scout_min && !continuation;
    pl.scout_pairs = pl.scouted && ctx->scout_pairs && E <= ctx->scout_pair_max_edges;
  }
  pl.bulk = pl.scouted && round_budget > 0 && ctx->geodesic_group && E >= ctx->geodesic_group_min;
  if (pl.bulk) {
    pl.group_waves = (E + 9) / 10;
    const size_t cap = (size_t)ctx->num_cus * (size_t)ctx->geodesic_group_waves_per_cu;
    if (pl.group_waves > cap) pl.group_waves = cap;
    pl.front_blocks = ctx->num_cus * (ctx->geodesic_group_front_per_cu > 0 ? ctx->geodesic_group_front_per_cu : 8);
    pl.low_cut = ctx->geodesic_group_low_cut > 0 ? ctx->geodesic_group_low_cut : (E < kGeoGroupHighCut ? 40 : (E < kGeoGroupHigherCut ? 48 : 56));
    pl.default_cut = ctx->geodesic_group_permille <= 0 && ctx->geodesic_group_pred <= 0;
    pl.handover_pct = ctx->geodesic_group_handover_pct >= 0 ? ctx->geodesic_group_handover_pct : (E < kGeoGroupLateHandoverFrom ? 50 : 80);
    if (pl.handover_pct > 0) {
      const size_t lat = (size_t)ctx->num_cus * (size_t)ctx->latency_blocks_per_cu;
      pl.drain_blocks = (int)(pl.group_waves * 10 < lat ? pl.group_waves * 10 : lat);
    }
  }
  return pl;
}

}  // namespace ccmp_host

namespace {

struct Line {
  char *buf;
  size_t cap, len = 0;
  void add(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
  {
    va_list ap;
    va_start(ap, fmt);
    char tmp[512];
    const int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (n <= 0) return;
    for (int i = 0; i < n && i < (int)sizeof tmp - 1; i++, len++)
      if (buf && len + 1 < cap) buf[len] = tmp[i];
    if (buf && cap) buf[len + 1 < cap ? len : cap - 1] = 0;
  }
};

}  // namespace

extern "C" {

int ccmp_ctx_option_info(int index, const char **name, long *dflt, long *lo, long *hi, const char **doc)
{
  if (index < 0 || index >= kNumOptions) return CCMP_EINVAL;
  const OptionDesc &o = kOptions[index];
  if (name) *name = o.name;
  if (dflt) *dflt = read_option(&ccmp_host::default_ctx(), o);
  if (lo) *lo = o.lo;
  if (hi) *hi = o.hi;
  if (doc) *doc = o.doc;
  return CCMP_OK;
}

int ccmp_ctx_get_option(const ccmp_ctx *ctx, const char *name, long *value)
{
  if (!name || !value) return CCMP_EINVAL;
  // read-only facts of a live context
  if (!strcmp(name, "num_cus")) { *value = (ctx ? ctx : &ccmp_host::default_ctx())->num_cus; return CCMP_OK; }
  if (!strcmp(name, "side_stream_busy")) { // 1 = the context's side stream still holds unfinished work (tests of the fork / join paths)
    if (!ctx) return CCMP_EINVAL;
    DeviceGuard guard(ctx->device);
    *value = hipStreamQuery(ctx->side) == hipSuccess ? 0 : 1;
    return CCMP_OK;
  }
  if (!strcmp(name, "resident")) { *value = ctx ? ctx->resident_on : 0; return CCMP_OK; }
  if (!strcmp(name, "resident_gave_up")) { *value = ctx ? ctx->resident_gave_up : 0; return CCMP_OK; } // how often a start gave up (that call took the launch path; the option stays on)
  const OptionDesc *o = find_option(name);
  if (!o) return CCMP_EINVAL;
  *value = read_option(ctx ? ctx : &ccmp_host::default_ctx(), *o);
  return CCMP_OK;
}

int ccmp_policy_set_option(ccmp_ctx *ctx, const char *name, long value) // (behind ccmp_ctx_set_option: ccmp_api.cpp handles "resident")
{
  if (!ctx || !name) return CCMP_EINVAL;
  const OptionDesc *o = find_option(name);
  if (!o) return CCMP_EINVAL;
  if (value < o->lo || value > o->hi) return CCMP_EINVAL;
  if ((o->flags & kNotZero) && value == 0) return CCMP_EINVAL;
  write_option(ctx, *o, value);
  return CCMP_OK;
}

int ccmp_ctx_describe(const ccmp_ctx *ctx_in, int call_kind, size_t n, char *buf, size_t cap)
{
  using namespace ccmp_host;
  const ccmp_ctx *ctx = ctx_in ? ctx_in : &default_ctx();
  Line L{buf, cap};
  if (buf && cap) buf[0] = 0;
  switch (call_kind) {
    case CCMP_CALL_PROJECT:
    case CCMP_CALL_SAMPLE_PROJECT: {
      const FdPlan pl = plan_fd_batch(ctx, n, ctx->order != nullptr);
      L.add("%s B=%zu: ", call_kind == CCMP_CALL_PROJECT ? "project" : "sample_project", n);
      if (pl.group_blocks == 0) {
        L.add("latency kernel alone (%s, %d blocks%s)", ctx->flat_kernel ? "project_fd_flat_kernel" : "project_fd_wave_kernel", pl.latency_blocks,
              pl.latency_static ? ", one per sample" : ", ticket queue");
        if (pl.latency_order) L.add(" in the FP32 scout's longest-predicted-first order");
        L.add(" [small_batch=%zu latency_order_min=%zu]", ctx->small_batch, ctx->latency_order_min);
        break;
      }
      if (pl.scout) L.add("FP32 scout + counting sort -> longest-predicted-first; ");
      if (pl.split)
        L.add("split launch: front = samples predicted >= %d iterations (<= %u) on %d project_fd_flat_kernel blocks on the side stream, beside ",
              pl.shape.pred, pl.shape.samples, pl.shape.blocks);
      L.add("project_fd_kernel x %d wavefronts", pl.group_blocks);
      if (pl.split) L.add(" (%d per CU left out)", pl.shape.cut);
      if (pl.handover) {
        if (pl.dump_threshold > 10) L.add("; hand-over below %d %% occupancy", pl.dump_threshold - 10);
        else L.add("; hand-over per wavefront at <= %d busy groups", pl.dump_threshold);
        L.add(" to %d %s blocks", pl.latency_blocks, ctx->flat_kernel ? "project_fd_flat_kernel" : "project_fd_wave_kernel");
        if (pl.two_class_pool) L.add(" in two classes (>= %d predicted iterations left first)", ctx->pool_long_remaining);
      } else L.add("; no hand-over");
      L.add(" [small_batch=%zu lpt_min_batch=%zu fd_split=%d:%zu..%zu wide<=%zu occupancy_rule<%zu no_handover>=%zu]", ctx->small_batch,
            ctx->lpt_min_batch, ctx->fd_split, ctx->fd_split_min, ctx->fd_split_max, kSplitWideMax, kOccupancyHandoverBelow, kNoHandoverFrom);
      break;
    }
    case CCMP_CALL_PROJECT_ANALYTIC: {
      const AnalyticPlan pl = plan_analytic_batch(ctx, n);
      L.add("project (analytic mode) B=%zu: ", n);
      if (pl.pair_blocks > 0) {
        L.add("project_pair_kernel (one sample per lane pair) x %d wavefronts", pl.pair_blocks);
        if (pl.latency_blocks > 0)
          L.add(", each handing over once the tickets are gone and it holds <= %d samples, then project_row16_kernel (sixteen lanes per sample) x %d wavefronts",
                pl.dump, pl.latency_blocks);
      } else L.add("project_row16_kernel (sixteen lanes per sample) alone x %d wavefronts", pl.latency_blocks);
      L.add(" [analytic_small_batch=%zu analytic_waves_per_cu=%d analytic_handover=%d]", ctx->analytic_small_batch, ctx->analytic_waves_per_cu, ctx->analytic_handover);
      break;
    }
    case CCMP_CALL_GEODESIC:
    case CCMP_CALL_GEODESIC_BUDGET: {
      const GeoPlan pl = plan_geodesic(ctx, n, call_kind == CCMP_CALL_GEODESIC_BUDGET ? 128 : 0, false);
      L.add("geodesic%s E=%zu: ", call_kind == CCMP_CALL_GEODESIC_BUDGET ? " (round budget)" : "", n);
      if (pl.ordered) L.add(pl.scouted ? "FP32 scout%s + counting sort -> longest-predicted-first; " : "far-apart edges first%s; ", pl.scout_pairs ? " on lane pairs" : "");
      if (pl.bulk) {
        L.add("bulk form: front (predicted >= 64 rounds if those edges carry %d permille of the work, else >= %d) on %d geodesic_flat_kernel blocks on the side stream, "
              "beside geodesic_group_kernel x %zu wavefronts",
              ctx->geodesic_group_heavy_permille, pl.low_cut, pl.front_blocks, pl.group_waves);
        if (ctx->geodesic_group_pred > 0) L.add(" (cut fixed at %d)", ctx->geodesic_group_pred);
        if (pl.handover_pct > 0)
          L.add("; hand-over below %d %% occupancy to %d blocks behind it", pl.handover_pct, pl.drain_blocks);
      } else {
        L.add("%s x %zu blocks%s", pl.latency_flavour ? "geodesic_flat_kernel_lat" : "geodesic_flat_kernel", pl.blocks, pl.queued ? ", ticket queue" : ", one per edge");
      }
      L.add(" [geodesic_order_min=%zu geodesic_scout_min=%zu geodesic_group_min=%zu high_cut_from=%zu higher_cut_from=%zu late_handover_from=%zu]", ctx->geodesic_order_min, ctx->geodesic_scout_min,
            ctx->geodesic_group_min, kGeoGroupHighCut, kGeoGroupHigherCut, kGeoGroupLateHandoverFrom);
      break;
    }
    default: return CCMP_EINVAL;
  }
  // without a context the plan is the built-in policy on an ASSUMED device: say so (block counts and the thresholds that mark
  // "as soon as the resident blocks take tickets" follow the CU count)
  if (!ctx_in) L.add(" {no context: a %d-CU device assumed}", ctx->num_cus);
  return (int)L.len; // the length the whole line needs (snprintf's convention); the buffer holds what fitted, NUL-terminated
}

}  // extern "C"
