/* ccmp_policy.h — which kernels a call runs on.  Pure functions of the context's settings and the call's size: ccmp_api.cpp
 * launches what they return and ccmp_ctx_describe() prints it, so the description cannot drift from the launch sequence.
 * Nothing here changes a result bit. */
#ifndef CCMP_POLICY_H
#define CCMP_POLICY_H
#include "ccmp_ctx.h"

namespace ccmp_host {

/* shape of the split launch's front (reference arithmetic, mid-size batches) */
struct SplitShape {
  int pred = 0;             // predicted iterations from which a sample belongs to the front
  int blocks = 0;           // latency blocks of the front
  int cut = 0;              // throughput wavefronts per CU the launch leaves out for them
  unsigned int samples = 0; // samples of the front at most
};

/* a reference-arithmetic projector batch */
struct FdPlan {
  int group_blocks = 0;        // persistent wavefronts of the throughput kernel; 0 = latency kernel alone
  bool handover = false;       // throughput kernel dumps its last samples to the pool, the latency kernel finishes them
  bool scout = false;          // FP32 scout pass + descending counting sort -> processing order
  bool latency_order = false;  // latency kernel alone, tickets through the scout's order
  int dump_threshold = 10;     // a wave hands over once the queue is dry and at most this many of its 10 groups are busy (> 10: occupancy rule)
  int latency_blocks = 0;      // grid of the latency kernel (direct launch or hand-over)
  bool latency_static = false; // one block per sample, static striding: no queue word to reset
  bool two_class_pool = false; // hand-over in two classes by the scout's remaining prediction
  bool split = false;          // split launch: the front on latency blocks on the side stream, beside the throughput kernel
  SplitShape shape;
};
FdPlan plan_fd_batch(const ccmp_ctx *ctx, size_t B, bool external_order);

/* an analytic-mode projector batch */
struct AnalyticPlan {
  int pair_blocks = 0;      // wavefronts of project_pair_kernel (one sample per lane pair); 0: the latency kernel alone
  int latency_blocks = 0;   // wavefronts of project_row16_kernel (sixteen lanes per sample) behind it, or alone; 0: none
  int dump = 0;             // a wavefront of the lane-pair kernel hands over once its tickets are gone and it holds at most this many samples
  size_t pool_records = 0;  // capacity that hand-over needs
};
AnalyticPlan plan_analytic_batch(const ccmp_ctx *ctx, size_t B);

/* an extend-step call */
struct GeoPlan {
  bool latency_flavour = false; // which build of geodesic_flat_kernel
  size_t blocks = 0;            // its grid
  bool queued = false;          // persistent blocks + ticket queue (more edges than resident blocks)
  bool ordered = false;         // an ordering pass runs ...
  bool scouted = false;         // ... and it is the FP32 scout (else: far-apart edges first)
  bool scout_pairs = false;
  bool bulk = false;            // short edges on geodesic_group_kernel, the front on latency blocks beside it
  size_t group_waves = 0;       // grid of geodesic_group_kernel
  int front_blocks = 0;         // grid of the front's launch
  int low_cut = 0;              // cut of the order where the edges beyond the scout's cap carry little work
  bool default_cut = false;     // the cut is the default rule's (one of two by what the batch looks like), decided in the sort's kernel
  int handover_pct = 0;         // occupancy below which the group kernel gives up everything (0 = never)
  int drain_blocks = 0;         // grid of the launch behind the group kernel
};
GeoPlan plan_geodesic(const ccmp_ctx *ctx, size_t E, int round_budget, bool continuation);

/* a context-shaped default for calls that have none at hand (ccmp_ctx_get_option / ccmp_ctx_describe with ctx == NULL):
 * the built-in settings on a 256-CU device */
const ccmp_ctx &default_ctx();

}  // namespace ccmp_host
#endif /* CCMP_POLICY_H */
