/* ccmp_resident.h — the opt-in resident service kernel of a context (option "resident"): layout of its mailbox, shared by the
 * device side (ccmp_kernels_resident.hip) and the host side (ccmp_resident.cpp).  Not part of the public ABI.
 *
 * What it is for.  The unchanged planner calls the constraint one state at a time — project(State*), isSatisfied(State*)
 * (src/base/jy_ProjectedStateSpace.cpp:13,20,27,65; src/planner/stefanBiPRM.cpp:397-398) — and a single-state call is a kernel
 * launch plus a completion poll: ~17 us of fixed cost around 38-105 us of Newton rounds, 70 % of an isSatisfied.  With the option
 * on, ONE persistent 128-thread block waits on a mailbox in pinned, device-mapped host memory: the host writes the state and a
 * tag, the block — already running — finds it with its next read over the link, computes with the latency kernel's Newton
 * routine (ccmp_flat_newton.h: the same bits), writes the result and the tag back.  No launch on the call path.  The same for ONE
 * edge of discreteGeodesic / checkMotion (stefanBiPRM.cpp:315-318,397-398 ask for them one pair at a time): the per-edge body of
 * geodesic_flat_kernel, included (ccmp_geo_edge_body.inc); the edge's states land in the mailbox.
 *
 * What keeps it from hanging anything.  A kernel that never ends blocks whatever waits for the device — hipDeviceSynchronize
 * (torch.cuda.synchronize), hipFree, anything queued behind it on its hardware queue.  So (1) it ends by itself after idle_ticks
 * without a request (default 10 ms; the next single-state call starts it again: one launch), (2) the library stops it — mailbox
 * command, then a stream wait — before every hipFree / hipMalloc / device-wide synchronise of its own (ccmp_host::quiesce) and in
 * ccmp_ctx_destroy, (3) it is never started under stream capture (only the synchronous *_host entry points start it), (4) every
 * wait on the host is bounded: no answer within the bound = CCMP_EHIP, and (5) while it is stopped the calls take the launch path.
 * Its stream has the LOWEST priority: streams of one priority share a few hardware queues and a stream queued behind a resident
 * kernel would wait for its idle exit; a priority of its own is a queue of its own (DESIGN_experiments.md §10.8). */
#ifndef CCMP_RESIDENT_H
#define CCMP_RESIDENT_H
#include <stddef.h>
#include <stdint.h>

/* commands (low 32 bits of word 0 of request line E) */
enum { kResNone = 0, kResProject = 1, kResFunction = 2, kResIsSatisfied = 3, kResJointValid = 4, kResStop = 5, kResGeodesic = 6 };
/* states (kResStateOff) */
enum { kResStarting = 0, kResRunning = 1, kResExited = 2 };

/* The mailbox: one pinned, device-mapped, coherent host allocation.
 *   [0, 2304)            ccmp_consts of the problem in force (the host rewrites it when the problem changes and bumps consts_seq)
 *   [2304, 2312)         state word, written by the device
 *   [4096, 4096 + 320)   request, five 64-byte lines; a line = 7 payload words + its tag (the request's sequence number) LAST:
 *                          line A: x[0..6] | tag      line B: x[7..13] | tag          (the state; `from` of an edge)
 *                          line C: to[0..6] | tag     line D: to[7..13] | tag         (an edge's target)
 *                          line E: (cmd | consts_seq << 32), (max_states | round_budget << 32), check_target, delta, lambda, 2 spare | tag
 *                        a tag = the request's sequence number (low half) | a checksum of ITS line's seven payload words (high
 *                        half).  The host fills the payloads, then the five tags (x86 stores become visible in program order); the
 *                        device acts when the five sequence numbers agree, differ from the last request it served, and every line's
 *                        checksum matches the payload it read — a line read in pieces while the host was writing is polled again
 *   [4608, 4608 + 192)   response: q_out[14], f[2], (ok | iters << 32), n_states, newton_iters, carry[2], 2 spare, done tag LAST
 *                        (behind a system fence)
 *   [8192, 8192 + 7168)  the states of an edge: kResMaxStates x 14 doubles */
constexpr size_t kResConstsOff = 0, kResStateOff = 2304, kResReqOff = 4096, kResRespOff = 4608, kResStatesOff = 8192, kResBoxBytes = 16384;
constexpr int kResMaxStates = 64;
constexpr int kResReqWords = 40;  /* words the polling lanes read */
constexpr int kResRespQ = 0, kResRespF = 14, kResRespFlags = 16, kResRespN = 17, kResRespIts = 18, kResRespCarry = 19, kResRespDone = 23; /* word indices in the response */

#include "../../include/ccmp.h"

struct ccmp_ctx;
namespace ccmp_host {
struct ResidentCall {
  int cmd;
  const double *x;  /* 14: the state; `from` of an edge */
  double *q_out;    /* 14, project */
  double *f;        /* 2, function */
  uint8_t *ok;      /* project / isSatisfied / jointValid / geodesic */
  uint16_t *iters;  /* project (nullable) */
  /* kResGeodesic — one edge of discreteGeodesic / checkMotion (ccmp_geodesic_host_ex with E == 1, no carry_in) */
  const double *to = nullptr; /* 14 */
  int max_states = 0, round_budget = 0, check_target = 0;
  double *states = nullptr;   /* [max_states][14]: the rows the traversal listed are written, the others left alone */
  int32_t *n_states = nullptr;
  double *carry_out = nullptr; /* 2, nullable */
};
constexpr int kResidentFallBack = 1; /* the service is off, stopped or cannot serve this problem right now: take the launch path */
/* one single-state call through the service; CCMP_OK, kResidentFallBack, or an error */
int resident_call(ccmp_ctx *ctx, const ccmp_problem *p, const ResidentCall &call);
/* stops the service and waits for its stream (before hipFree / hipMalloc / device-wide synchronisation; idempotent, cheap when
 * nothing runs); the next single-state call starts it again if the option is still on */
void quiesce(ccmp_ctx *ctx);
/* is `ctx` a context ccmp_ctx_create made and ccmp_ctx_destroy has not yet taken?  (objects that remember their context — scenes — ask before
 * touching it: a scene may outlive the context it was created on) */
bool context_alive(const ccmp_ctx *ctx);
/* option "resident": 1 = on (started lazily by the first single-state *_host call), 0 = stop and release */
int resident_set(ccmp_ctx *ctx, long on);
/* ccmp_ctx_destroy */
void resident_destroy(ccmp_ctx *ctx);
}  // namespace ccmp_host
#endif /* CCMP_RESIDENT_H */
