/* ccmp.h — C ABI of libccmp: batched closed-chain constraint projector for AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE hot path of jkw0701/closed_chain_motion_planner: the Newton
 * retraction KinematicChainConstraint::project() and its sample/extend callers.  The reference has
 * no FFI layer (the path sits behind OMPL's C++ virtuals); every entry point below therefore names
 * the reference C++ member it replaces (paths relative to the reference checkout).  The C++ adapter
 * that keeps the OMPL surface source-compatible is include/ccmp_ompl_adapter.hpp; the binding a
 * maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no exceptions cross the ABI; every function returns CCMP_OK (0) or a negative code
 *   - all pointers are caller-owned; *_batch calls take DEVICE pointers and are asynchronous on
 *     the caller's HIP stream (NULL = the HIP default stream, as in every HIP API); *_host calls
 *     take host pointers, run on the ctx's own stream and are synchronous; a *_batch call may be recorded into a HIP
 *     graph by stream capture and replayed (tests/test_gpu_usage_modes.py) — after one eager call at that size, which grows
 *     the context's workspaces
 *   - state layout everywhere: row-major double q[B][14]; first 7 = the alphabetically first arm
 *     name, next 7 = the second (std::map order, src/base/constraints/ConstrainedPlanningCommon.cpp:89-91)
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     CCMP_ENODEV
 */
#ifndef CCMP_H
#define CCMP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCMP_VERSION 600

enum {
  CCMP_OK = 0,
  CCMP_EINVAL = -1, /* bad argument (also: non-positive tolerance, ConstraintFunction.h:106-108) */
  CCMP_EHIP = -2,   /* a HIP runtime call failed; see ccmp_last_hip_error()                      */
  CCMP_EIO = -3,    /* cannot read the YAML file                                                 */
  CCMP_EPARSE = -4, /* YAML key missing or malformed                                             */
  CCMP_ENODEV = -5, /* no HIP device / device index out of range                                 */
  CCMP_ENOMEM = -6,
  CCMP_ECOMM = -7,    /* RCCL call failed or librccl.so could not be opened; see ccmp_last_hip_error()  */
  CCMP_EOVERFLOW = -8 /* more valid states than the gather blocks / the output buffer hold: retry larger */
};

enum { CCMP_JAC_FD = 0, CCMP_JAC_ANALYTIC = 1 };

/* Everything the kernels need besides q.  POD; matrices row-major.  Built by
 * ccmp_problem_from_yaml / ccmp_problem_init, or filled by hand. */
typedef struct ccmp_problem {
  double axis[2][7][3];   /* joint axes in the parent body frame; src/kinematics/panda_rbdl.cpp:117-122 */
  double offset[2][7][3]; /* joint origin relative to the parent joint origin; panda_rbdl.cpp:128-130   */
  double ee[2][3];        /* rot_ee*(0,0,0.107); panda_rbdl.cpp:124-126                                 */
  double R_tool[2][9];    /* rot_ee*Rz(-pi/4); panda_rbdl.cpp:31                                        */
  double base_R[2][9];    /* t_wb per arm; src/kinematics/grasping_point.cpp:11-20                      */
  double base_p[2][3];
  double init_R[9];       /* init_chain_; ConstraintFunction.h:39                                       */
  double init_p[3];
  double lb[7];           /* ConstraintFunction.h:27-28 == KinematicChain.h:75-100                      */
  double ub[7];
  double joint_eps;       /* 0.001; ConstraintFunction.h:45                                             */
  double tol_pos;         /* tolerance1_ = 1e-3; ConstrainedPlanningCommon.cpp:120                      */
  double tol_rot;         /* tolerance2_ = 5e-3; ConstrainedPlanningCommon.cpp:121                      */
  double step;            /* 0.30; ConstraintFunction.h:71                                              */
  double delta;           /* 0.25; ConstrainedPlanningCommon.cpp:118                                    */
  double lambda;          /* 2.0;  ConstrainedPlanningCommon.cpp:119                                    */
  double start_joint[14]; /* config/<obj>.yaml: start_joint                                             */
  double obj_start_R[9];  /* t_wo_start; grasping_point.cpp:38-43                                       */
  double obj_start_p[3];
  double obj_goal_R[9];   /* t_wo_goal; grasping_point.cpp:45-50                                        */
  double obj_goal_p[3];
  double t_o7_R[2][9];    /* t_o7 per arm; ConstrainedPlanningCommon.cpp:110-111                        */
  double t_o7_p[2][3];
  int32_t max_iter;       /* 250; ConstraintFunction.h:26 (setMaxIterations(1000) never reaches it)     */
  int32_t jacobian_mode;  /* CCMP_JAC_FD (reference arithmetic, default) or CCMP_JAC_ANALYTIC           */
  int32_t arm_index[2];   /* 0 panda_left, 1 panda_right, 2 panda_top; grasping_point.cpp:59-63         */
} ccmp_problem;

/* ---- problem set-up (host, once per planning problem) ------------------------------------------ */
/* replaces grasping_point::loadConfig (grasping_point.cpp:34-65) + ConstrainedProblem::_setEnvironment
 * + setConstrainedOptions (ConstrainedPlanningCommon.cpp:85-132): reads the reference's YAML
 * unchanged (keys start_joint, t_wo_{start,goal}_{pos,quat}, arm1/arm2.{name,index}). */
int ccmp_problem_from_yaml(const char *yaml_path, ccmp_problem *out);
/* the same from explicit values; obj_* may be NULL (identity pose) */
int ccmp_problem_init(ccmp_problem *out, const char *arm1_name, int arm1_index, const char *arm2_name,
                      int arm2_index, const double start_joint[14], const double obj_start_pos[3],
                      const double obj_start_quat_xyzw[4], const double obj_goal_pos[3],
                      const double obj_goal_quat_xyzw[4]);
/* KinematicChainConstraint::setArmModels (ConstraintFunction.h:122-126) on a problem that is already set up — the
 * reference calls it after loadConfig (ConstrainedPlanningCommon.cpp:126): arm selection and base frames change, init_chain_
 * and t_o7 are recomputed from the stored start_joint and object pose; tolerances, delta/lambda, calibration (kept per
 * slot), mode and object poses stay.  The arms are stored IN THE ORDER GIVEN (arm1 = the first seven joints), as the
 * reference's setArmModels does; std::map (alphabetical) order is what ConstrainedProblem::_setEnvironment hands it
 * (ConstrainedPlanningCommon.cpp:89-91) and what ccmp_problem_init / ccmp_problem_from_yaml establish here.  Base frames come
 * from the index through the table of src/kinematics/grasping_point.cpp:11-20; ccmp_set_base_frame overrides one. */
int ccmp_set_arms(ccmp_problem *p, const char *arm1_name, int arm1_index, const char *arm2_name, int arm2_index);
/* ArmModel::t_wb of the arm in `arm_slot` (0 / 1), row-major rotation + translation — the frame
 * KinematicChainConstraint::function multiplies with (ConstraintFunction.h:89-90; panda_model.h:15; filled at
 * ConstrainedPlanningCommon.cpp:98).  The adapter's setArmModels passes what the ArmModel carries, so an edit of
 * grasping_point.cpp:11-20 reaches the GPU without touching this library.  init_chain_ / t_o7 are recomputed.
 * CCMP_EINVAL unless R is a finite rotation (orthonormal to 1e-9, determinant > 0) and pos is finite. */
int ccmp_set_base_frame(ccmp_problem *p, int arm_slot, const double R[9], const double pos[3]);
/* KinematicChainConstraint::setInitialPosition (ConstraintFunction.h:31-40) and the t_o7 of
 * ConstrainedPlanningCommon.cpp:110-111 */
int ccmp_set_start(ccmp_problem *p, const double q0[14]);
/* KinematicChainConstraint::setTolerance (ConstraintFunction.h:104-112): CCMP_EINVAL where the
 * reference throws ompl::Exception */
int ccmp_set_tolerance(ccmp_problem *p, double tolerance1, double tolerance2);
/* PandaModel::initModel(dh) with calibration offsets (panda_rbdl.cpp:80-148; columns a,d,theta,alpha) */
int ccmp_set_calibration(ccmp_problem *p, int arm_slot, const double dh_offsets[7][4]);

/* ---- execution context -------------------------------------------------------------------------- */
/* A context owns one device, one internal stream (used by the *_host entry points) and the work queues and
 * workspaces of the projector kernels.  Launches made through ONE context must be ordered: issue them on one
 * stream, or synchronise between streams — two projector launches of the same context running at once would
 * share queue words.  The same holds for EVERY launch-making call of a context, not only ccmp_project_* /
 * ccmp_sample_*: ccmp_geodesic_* and ccmp_check_motion_* keep their ticket and order counters in the context's queue words
 * and take their FP32 scout's workspace (predictions, histogram, processing order) from the same context-owned buffer the
 * projector's scout and two-class hand-over use — a buffer that grows (is freed and reallocated) on the first call at a
 * larger size — and the *_host / *_sharded* entry points use the context's staging buffer.  Overlapping any two of them on
 * different streams of ONE context is a data race; use one context per stream.  Calls on one context from several
 * threads must be serialised by the caller (the
 * reference serialises its constraint calls with graphMutex_, src/planner/stefanBiPRM.cpp:280,383,449); use one
 * context per thread or per stream for concurrency.  The problem description is passed by value into every
 * launch: it may be changed between calls without synchronising. */
typedef struct ccmp_ctx ccmp_ctx;
int ccmp_ctx_create(int device, ccmp_ctx **out);
void ccmp_ctx_destroy(ccmp_ctx *ctx);
/* persistent waves per CU for the projector kernels (0 = built-in default); = option "waves_per_cu" */
int ccmp_ctx_set_waves_per_cu(ccmp_ctx *ctx, int waves_per_cu);
/* projector scheduling (= options "hand_over", "small_batch"): hand_over 0 = throughput kernel (10 samples per wavefront) only,
 * 1 = that kernel until the sample queue drains, then the latency kernel on the samples still in flight (default), 2 = latency
 * kernel only; batches of at most small_batch samples always use the latency kernel (CCMP_DEFAULT = the built-in default of the
 * table below).  Results are bit-identical under every setting. */
#define CCMP_DEFAULT ((size_t)-1)
int ccmp_ctx_set_schedule(ccmp_ctx *ctx, int hand_over, size_t small_batch);
/* longest-predicted-first scheduling of large reference-arithmetic batches (= options "lpt", "lpt_min_batch"): mode 0 = index
 * order; 1 = an FP32 analytic-Jacobian scout pass predicts each sample's iteration count and the batch is processed longest
 * first, straggler hand-over kept below 131072 samples and dropped from there on (default); 2 = the same without hand-over at any size.
 * Used for batches of at least min_batch samples (CCMP_DEFAULT = the built-in default of the table below). */
int ccmp_ctx_set_lpt(ccmp_ctx *ctx, int mode, size_t min_batch);
/* Tuning options, by name.  NONE of them changes a result bit: they choose which kernels a call runs on and where the regimes
 * meet.  ccmp_ctx_set_option: CCMP_EINVAL for an unknown name or a value outside the range.  ccmp_ctx_get_option: the value in
 * force; with ctx == NULL the built-in default (no device needed); besides the table it answers "num_cus", "resident",
 * "resident_gave_up" and "side_stream_busy" (1 while the context's side stream still holds unfinished work).  ccmp_ctx_option_info enumerates the table
 * (index 0 .. until CCMP_EINVAL; any out-pointer may be NULL).  The table below is GENERATED from the library's
 * (tools/gen_option_docs.py) and compared with it by the CPU test suite, so a default stated here is the default in the code.
 * "resident" (0 / 1, default 0; not in the table: it starts and stops something) — see "resident service kernel" below. */
int ccmp_ctx_set_option(ccmp_ctx *ctx, const char *name, long value);
int ccmp_ctx_get_option(const ccmp_ctx *ctx, const char *name, long *value);
int ccmp_ctx_option_info(int index, const char **name, long *dflt, long *lo, long *hi, const char **doc);
/* BEGIN OPTION TABLE (generated from csrc/ccmp_policy.cpp: python tools/gen_option_docs.py --write)
 *   name                            default   range         meaning
 *   "hand_over"                     1         0..2          0 = throughput kernel (10 samples per wavefront) only, 1 = that kernel until its queue
 *                                                           drains, then the latency kernel on what is in flight, 2 = latency kernel only (=
 *                                                           ccmp_ctx_set_schedule)
 *   "small_batch"                   10240     0..max        batches of at most this many samples run on the latency kernel alone (=
 *                                                           ccmp_ctx_set_schedule)
 *   "waves_per_cu"                  0         0..32         persistent wavefronts of the throughput kernels per CU, 0 = 12 (=
 *                                                           ccmp_ctx_set_waves_per_cu)
 *   "lpt"                           1         0..2          0 = index order, 1 = FP32 scout + longest-predicted-first, hand-over kept below 131072
 *                                                           samples, 2 = the same without hand-over (= ccmp_ctx_set_lpt)
 *   "lpt_min_batch"                 16384     0..max        the scout's order from this many samples on (= ccmp_ctx_set_lpt)
 *   "latency_order_min"             2049      0..max        latency kernel alone: tickets through the scout's order from this many samples on
 *   "flat_kernel"                   1         0..1          latency work: 1 = one sample per 128-thread block, an iteration's evaluations in one
 *                                                           round, 0 = one wavefront per sample
 *   "stock_kernels"                 1         0..1          1 = kernels that skip the exact zeros of the uncalibrated Panda when both arms carry them
 *                                                           (same bits), 0 = always the general kernels
 *   "handover_threshold"            -1        -1..110       -1 = automatic; 0..10: a wavefront hands over once the queue is dry and at most this many
 *                                                           of its 10 groups are busy; 11..110: all hand over once the samples in flight fill less
 *                                                           than (value - 10) % of the group slots
 *   "fd_split"                      1         0..1          1 = split launch: above small_batch, the predicted-longest samples run on latency blocks
 *                                                           on a side stream beside the throughput kernel
 *   "fd_split_min"                  0         0..max        split launch from this many samples ...
 *   "fd_split_max"                  90112     0..max        ... up to this many
 *   "fd_split_pred"                 -1        -1..1023      predicted iterations from which a sample belongs to the front (-1: 40 up to 24576
 *                                                           samples, 56 above)
 *   "fd_split_front"                -1        -1..4096      latency blocks of the front (-1: two per CU up to 24576 samples, one above; 0 = no split)
 *   "fd_split_samples"              -1        -1..2147483647 samples of the front at most (-1: four per CU up to 24576 samples, three to four above;
 *                                                            0 = one per block)
 *   "fd_split_group_cut"            -1        -1..8         throughput wavefronts per CU left out for the front's blocks (-1: 3 up to 24576 samples,
 *                                                           2 above)
 *   "analytic_small_batch"          8192      0..max        analytic mode: at or below, the sixteen-lanes-per-sample latency kernel alone
 *   "analytic_waves_per_cu"         12        1..12         analytic mode: persistent wavefronts of the lane-pair kernel per CU
 *   "analytic_handover"             8         0..32         analytic mode: a wavefront of the lane-pair kernel whose tickets are gone hands over to
 *                                                           the latency kernel once it holds at most this many samples (0 = never: one launch)
 *   "scout_pairs"                   1         0..1          1 = two lanes per sample / edge, one arm each, where lanes are plentiful (stock twin
 *                                                           arms)
 *   "geodesic_flavour"              0         0..2          two builds, same bits: 0 = throughput build for calls with a round budget beyond the
 *                                                           latency build's blocks, latency build otherwise; 1 / 2 = always the throughput / latency
 *                                                           build
 *   "geodesic_order"                2         0..2          batches beyond the resident blocks: 0 = index order, 1 = far-apart edges first, 2 = FP32
 *                                                           scout order from geodesic_scout_min edges on
 *   "geodesic_order_min"            2049      0..max        no ordering pass below this many edges
 *   "geodesic_long_steps"           12        0..max        order 1: edges further apart than this many delta count as long
 *   "geodesic_scout_min"            2049      0..max        the scout from this many edges on
 *   "geodesic_scout_rounds"         64        1..1023       the scout stops an edge after this many Newton rounds
 *   "geodesic_group"                1         0..1          1 = bulk calls (round budget, scout order): short edges ten to a wavefront on the
 *                                                           throughput layout, the front of the order on latency blocks beside them
 *   "geodesic_group_min"            13312     0..max        ... from this many edges
 *   "geodesic_group_pred"           -1        -1..1023      ... cut of the order in predicted rounds (-1: the scout's cap where the edges beyond it
 *                                                           carry a tenth of the predicted work, else geodesic_group_low_cut)
 *   "geodesic_group_low_cut"        -1        -1..64        ... (-1: 40 below 20480 edges, 48 from there on, 56 from 65536)
 *   "geodesic_group_permille"       0         0..1000       ... > 0: instead, the largest cut whose front carries this share of the predicted work
 *   "geodesic_group_handover_pct"   -1        -1..100       ... with the queue dry, every wavefront gives its edges to latency blocks once those in
 *                                                           flight fill less than this share of the slots (0 = never; -1: 50 below 32768 edges, 80
 *                                                           from there on)
 *   "clearance_per_state_max"       8192      0..max        proxy clearance: one block per state up to this many states, 64-state tiles above
 *   "host_zero_copy"                2         0..2          *_host calls on page-locked caller buffers: 0 = staged, 1 = q_out written in place, 2 =
 *                                                           q_in read in place too
 *   "resident_idle_ms"              10        1..10000      the resident service kernel (option "resident") leaves by itself after this many
 *                                                           milliseconds without a request
 * END OPTION TABLE */
/* Resident service kernel (option "resident", 0 / 1, default 0): with it on, the one-at-a-time calls the unchanged planner makes —
 * ccmp_project_host, ccmp_function_host, ccmp_is_satisfied_host, ccmp_joint_valid_host with B == 1, and ccmp_geodesic_host /
 * ccmp_geodesic_host_ex / ccmp_check_motion_host with E == 1 (no carry_in, max_states <= 64), reference arithmetic: the reference's
 * project(State*) / isSatisfied(State*) / discreteGeodesic and checkMotion of one pair, src/base/jy_ProjectedStateSpace.cpp:13,20,27,
 * 32-96, src/planner/stefanBiPRM.cpp:315-318,397-398 — are served by ONE persistent 128-thread block that waits on a mailbox in pinned host memory, instead of a kernel launch
 * and a completion poll each: the same bits, ~10-13 us less per call.  Started by the first such call, never under stream capture.
 * What a kernel that stays on the device asks of the caller: the LIBRARY stops it before every hipFree / hipMalloc / device-wide
 * synchronise of its own and in ccmp_ctx_destroy (and takes the launch path while it is stopped), so every entry point of this
 * header can be mixed with resident calls; the APPLICATION's own hipDeviceSynchronize / hipFree waits until the service leaves by
 * itself, after "resident_idle_ms" (default 10) without a request — bounded, never for ever; the next call starts it again.  Its
 * stream has the lowest priority (a hardware queue of its own); should the kernel not get to run within 5 ms of a start — its queue
 * is shared with something that does not end soon, e.g. another context's service — THAT CALL takes the launch path, the kernel is
 * told to stop and abandoned (never waited for), and a later call starts the service again behind a back-off (10 ms, doubling up
 * to 1 s; "resident_gave_up" counts the occasions; setting "resident" to 1 again clears both).  Every host-side wait is bounded:
 * a request whose worst case exceeds 1.5 s goes to the launch path, CCMP_EHIP after 2 s without an answer (the service is then not
 * used again by the context). */
/* What the policy does with a call: writes ONE line into buf (NUL-terminated, truncated to cap) naming the kernels a call of
 * kind call_kind over n samples / edges runs on under the context's present settings, and the thresholds that delimit that
 * regime — computed by the same functions the launches use (csrc/ccmp_policy.cpp), so it cannot disagree with them.  ctx == NULL:
 * the built-in policy on a 256-CU device.  Returns the length of the whole line (snprintf's convention) or CCMP_EINVAL. */
enum {
  CCMP_CALL_PROJECT = 0,          /* ccmp_project_batch, reference arithmetic          */
  CCMP_CALL_SAMPLE_PROJECT = 1,   /* ccmp_sample_project_batch                         */
  CCMP_CALL_PROJECT_ANALYTIC = 2, /* ccmp_project_batch with CCMP_JAC_ANALYTIC         */
  CCMP_CALL_GEODESIC = 3,         /* ccmp_geodesic_batch / _ex without a round budget  */
  CCMP_CALL_GEODESIC_BUDGET = 4   /* ccmp_geodesic_batch_ex with round_budget > 0      */
};
int ccmp_ctx_describe(const ccmp_ctx *ctx, int call_kind, size_t n, char *buf, size_t cap);
int ccmp_ctx_device(const ccmp_ctx *ctx);
int ccmp_ctx_num_cus(const ccmp_ctx *ctx);

/* ---- the hot path: device pointers, asynchronous on `hip_stream` -------------------------------- */
/* KinematicChainConstraint::function (ConstraintFunction.h:84-102): f[i] = (|dp|, angle) */
int ccmp_function_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, double *f, size_t B,
                        void *hip_stream);
/* KinematicChainConstraint::project (ConstraintFunction.h:57-82): q_out[i] = final iterate whether
 * or not ok[i]; ok[i] = the reference's return value; iters[i] (nullable) = Newton updates done.
 * q_in == q_out is allowed (in place, as the reference).  q_in and q_out must be 16-byte aligned (any row of a
 * hipMalloc'ed [B][14] array is): CCMP_EINVAL otherwise. */
int ccmp_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q_in, double *q_out,
                       uint8_t *ok, uint16_t *iters, size_t B, void *hip_stream);
/* KinematicChainConstraint::isSatisfied (ConstraintFunction.h:114-120) */
int ccmp_is_satisfied_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok,
                            size_t B, void *hip_stream);
/* KinematicChainConstraint::jointValid (ConstraintFunction.h:43-55) */
int ccmp_joint_valid_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B,
                           void *hip_stream);
/* jy_ProjectedStateSampler::sampleUniform (src/base/jy_ProjectedStateSpace.cpp:10-15): ambient
 * uniform sample (counter-based: sample i, dim j uses splitmix64(seed ^ ((first_index+i)*14+j))) ->
 * project (result kept in ok[], the reference ignores it) -> enforceBounds (KinematicChain.h:118-130).
 * q_ambient (nullable) receives the un-projected samples. */
int ccmp_sample_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                              double *q_out, uint8_t *ok, uint16_t *iters, double *q_ambient, size_t B,
                              void *hip_stream);
/* jy_ProjectedStateSampler::sampleUniformNear (jy_ProjectedStateSpace.cpp:17-22): per dimension
 * uniform in [max(low, near-d), min(high, near+d)], then project, then enforceBounds.  `near` holds one
 * state per sample (near_stride 14) or one shared state (near_stride 0). */
int ccmp_sample_near_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                                   const double *near, int near_stride, double distance, double *q_out, uint8_t *ok,
                                   uint16_t *iters, double *q_ambient, size_t B, void *hip_stream);
/* jy_ProjectedStateSampler::sampleGaussian (jy_ProjectedStateSpace.cpp:24-29): mean + stdDev*N(0,1) per
 * dimension (Box-Muller on counter-based uniforms), clamped to the bounds, then project, then enforceBounds */
int ccmp_sample_gaussian_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                                       const double *mean, int mean_stride, double std_dev, double *q_out, uint8_t *ok,
                                       uint16_t *iters, double *q_ambient, size_t B, void *hip_stream);
/* IKTask::compute_t_wo (src/base/constraints/ik_task.cpp:10-14): object pose t_wb*FK(q)*t_o7^-1 from the
 * left arm's 7 joints (q + i*q_stride); t_wo[i] = rotation (9, row-major) then translation (3) */
int ccmp_compute_t_wo_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, int q_stride, double *t_wo, size_t B,
                            void *hip_stream);
/* jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:32-96), E edges at once:
 * states[e][0..n_states[e]) (capacity max_states each) = `from`, then every accepted state; ok[e] = the reference's
 * return value.  An edge whose list would need more than max_states entries stops there and reports n_states[e] =
 * max_states + 1 (ok[e] = 0): repeat it with a larger buffer before anything is concluded from it (the adapter and the
 * Python mirror do) — a creeping edge (observed: 952 accepted states, each a hair closer to the target) must not hold
 * a whole launch, and a cut list must never look complete.  Runs as the reference does with interpolate == true; for
 * interpolate == false the host truncates at the first state its StateValidityChecker rejects.
 * With jacobian_mode = CCMP_JAC_ANALYTIC the traversal is a step loop around the batched analytic projector (at most max_states steps of
 * three launches each, no host synchronisation): the same lists, counts, flags and carries as the analytic mode's CPU restatement, bit for
 * bit; newton_iters must be given, a round budget (ccmp_geodesic_batch_ex) is not enforced in that mode (ok is never 2), the resident
 * service does not serve it. */
int ccmp_geodesic_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                        double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters, void *hip_stream);
/* The same, resumable.  carry_out (nullable, [E][2]) receives what a continuation needs: the running length and the
 * bound lambda * dist(from, to).  An edge can stop short of its end in two ways, and either way a later call with
 * from[e] = its last stored state, the same to[e] and carry_in[e] = that carry_out[e] goes on where it stopped (its list
 * starts with that state again: drop it when joining the lists); first call + continuations give the states, flags and
 * Newton counts of ONE uninterrupted traversal, bit for bit:
 *   - its list is full: n_states[e] = max_states + 1, ok[e] = 0, last stored state = states[e][max_states - 1];
 *   - round_budget > 0 and the edge has spent that many Newton rounds (Jacobian evaluations) in this call: it stops between
 *     two states, ok[e] = 2, n_states[e] = the states stored so far (last stored state = states[e][n_states[e] - 1]).
 * A call thereby bounds the serial work it spends on any one edge — one edge in 16 384 near-neighbour edges creeps (952
 * states, 12 379 Newton rounds) and two dozen need more than 128 rounds; without a bound a launch lasts as long as its
 * longest edge.  round_budget = 0: no bound (ok is 0 / 1 only).  check_target as in ccmp_check_motion_batch (not together
 * with carry_in: the target was tested by the call being continued).  A resumable call (carry_in, carry_out or round_budget
 * given) needs max_states >= 2 — a one-entry list holds `from` only and its continuation would start from `from` again:
 * CCMP_EINVAL otherwise. */
int ccmp_geodesic_batch_ex(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                           double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters, const double *carry_in,
                           double *carry_out, int round_budget, int check_target, void *hip_stream);
/* OMPL ConstrainedMotionValidator::checkMotion as the reference's planner calls it (src/planner/stefanBiPRM.cpp:397-398,
 * 463-464; jy_MotionValidator, jy_ProjectedStateSpace.h:57-69): isSatisfied(to) && discreteGeodesic(from, to) in ONE
 * launch — same outputs as ccmp_geodesic_batch, except that an edge whose target fails isSatisfied reports ok = 0 and
 * only `from` (n_states = 1) without being traversed */
int ccmp_check_motion_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                            double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters, void *hip_stream);
/* the ambient sampler alone (RealVectorStateSampler::sampleUniform over KinematicChain.h:75-100) */
int ccmp_ambient_uniform_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                               double *q_out, size_t B, void *hip_stream);
/* KinematicChainSpace::enforceBounds (KinematicChain.h:118-130), in place */
int ccmp_enforce_bounds_batch(ccmp_ctx *ctx, double *q, size_t B, void *hip_stream);
/* stable compaction of the rows with ok != 0 (what the host tree consumes); *count_dev is a device
 * uint64 */
int ccmp_compact_valid(ccmp_ctx *ctx, const double *q, const uint8_t *ok, size_t B, double *q_valid,
                       uint64_t *count_dev, void *hip_stream);
/* the same into a buffer of `capacity` rows (fixed-size send buffers of the all-gather): valid rows past the capacity
 * are dropped, *count_dev still reports how many there were — a count above the capacity tells the consumer */
int ccmp_compact_valid_capped(ccmp_ctx *ctx, const double *q, const uint8_t *ok, size_t B, double *q_valid,
                              size_t capacity, uint64_t *count_dev, void *hip_stream);

/* ---- host-pointer conveniences (H2D, kernel, D2H; synchronous) ----------------------------------
 * ccmp_project_host recognises a caller's PAGE-LOCKED q_in / q_out (hipHostMalloc, hipHostRegister; a batch larger than 64 KB,
 * both 16-byte aligned): the kernels then write the projected rows straight into q_out instead of staging and downloading
 * them (option "host_zero_copy": 2 = that and q_in is read in place as well, default; 1 = q_in is uploaded by one copy first;
 * 0 = staged like pageable memory). */
int ccmp_project_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q_in, double *q_out, uint8_t *ok,
                      uint16_t *iters, size_t B);
int ccmp_function_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, double *f, size_t B);
int ccmp_is_satisfied_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B);
int ccmp_joint_valid_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B);
int ccmp_sample_project_host(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index, double *q_out,
                             uint8_t *ok, uint16_t *iters, size_t B);
/* sampleUniformNear (kind 0, param = distance) / sampleGaussian (kind 1, param = stdDev) x B around ONE host reference
 * state (jy_ProjectedStateSpace.cpp:17-29): ambient draw, project, enforceBounds */
int ccmp_sample_ref_project_host(ccmp_ctx *ctx, const ccmp_problem *p, int kind, uint64_t seed, uint64_t first_index,
                                 const double ref[14], double param, double *q_out, uint8_t *ok, uint16_t *iters, size_t B);
/* states: [E][max_states][14], n_states: [E], ok: [E] (host buffers) */
int ccmp_geodesic_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                       double *states, int32_t *n_states, uint8_t *ok);
int ccmp_check_motion_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                           double *states, int32_t *n_states, uint8_t *ok);
int ccmp_geodesic_host_ex(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                          double *states, int32_t *n_states, uint8_t *ok, const double *carry_in, double *carry_out, int round_budget,
                          int check_target);

/* ---- one process, several GPUs (the reference's planner is a single process) ------------------------- */
/* Contiguous shards of the batch go to the n contexts (1 <= n <= 64, one per device; the same device may appear twice),
 * each shard is uploaded, projected and downloaded on its context's own stream, all concurrently; returns
 * when every shard is back.  No collective is needed: every GPU returns its shard straight to the host
 * tree.  Results are bit-identical to a single-GPU call.  (With the collective: ccmp_project_sharded below; the
 * one-process-per-GPU form lives above the ABI: closed_chain_motion_planner_amd/distributed.py.) */
int ccmp_project_sharded_host(ccmp_ctx *const *ctxs, int n, const ccmp_problem *p, const double *q_in, double *q_out,
                              uint8_t *ok, uint16_t *iters, size_t B);
/* sampleUniform x B across the contexts; sample i is a function of (seed, first_index + i) only */
int ccmp_sample_project_sharded_host(ccmp_ctx *const *ctxs, int n, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                                     double *q_out, uint8_t *ok, uint16_t *iters, size_t B);
/* Per context (n values each) of the last ccmp_*_sharded_host / ccmp_*_sharded call: launch_ms = milliseconds from the
 * call's entry to the moment the host had that shard's upload behind it and issued its kernels; start_ms = the same moment
 * on the GPU's timeline relative to ctxs[0]'s (an event on each context's stream; -1 for a context on another device than
 * ctxs[0], where no common timeline exists).  Every shard is driven by its own short-lived host thread, so both stay
 * within a fraction of a millisecond of each other for any number of GPUs; a serial upload would show as values growing
 * with the shard index (a pageable 29 MB shard: ~1.2 ms each). */
int ccmp_sharded_host_last_timing(ccmp_ctx *const *ctxs, int n, double *launch_ms, double *start_ms);

/* ---- one process, several GPUs, with the collective (ccmp_project_sharded) ------------------------- */
/* A communicator over the devices of n contexts (one context per device, 1 <= n <= 64): ncclCommInitAll in this
 * process — RCCL over xGMI inside a node; librccl.so is opened at run time, libccmp.so does not link against it. */
typedef struct ccmp_comm ccmp_comm;
int ccmp_comm_create(ccmp_ctx *const *ctxs, int n, ccmp_comm **out);
void ccmp_comm_destroy(ccmp_comm *comm);
/* per GPU (n values each), measured on that GPU's stream during the last ccmp_*_sharded call: milliseconds from the start
 * of its shard to the end of its projection + compaction, and from there to the completion of the all-gather (-1: no
 * call yet).  What a multi-GPU run is diagnosed by: stragglers show in kernel_ms, a slow collective in gather_ms. */
int ccmp_comm_last_timing(const ccmp_comm *comm, double *kernel_ms, double *gather_ms);
/* project(x) x B (host buffers), contiguous shards over the communicator's GPUs; every GPU compacts its VALID states
 * into a block of block_rows rows (+ one leading row that carries its count) and ONE ncclAllGather brings all blocks
 * to every GPU; GPU 0 hands them to the host: valid_out[0..*n_valid) (capacity valid_capacity rows) = the valid
 * states of all shards in global sample order, counts[g] (nullable, n entries) = valid states of shard g.  q_out / ok /
 * iters (all nullable) additionally receive the full per-sample results, each GPU returning its shard directly.
 * CCMP_EOVERFLOW if a shard holds more valid states than block_rows or the total exceeds valid_capacity (counts and
 * *n_valid are filled: retry with room).  Bit-identical to a single-GPU call. */
int ccmp_project_sharded(ccmp_comm *comm, const ccmp_problem *p, const double *q_in, size_t B, double *q_out, uint8_t *ok,
                         uint16_t *iters, size_t block_rows, double *valid_out, size_t valid_capacity, uint64_t *counts,
                         uint64_t *n_valid);
/* sampleUniform x B the same way; sample i is a function of (seed, first_index + i) only */
int ccmp_sample_project_sharded(ccmp_comm *comm, const ccmp_problem *p, uint64_t seed, uint64_t first_index, size_t B,
                                double *q_out, uint8_t *ok, uint16_t *iters, size_t block_rows, double *valid_out,
                                size_t valid_capacity, uint64_t *counts, uint64_t *n_valid);

/* ---- proxy-geometry clearance: a pre-filter in FRONT of the MoveIt validity test ----------------------------------- */
/* KinematicChainValidityChecker::isValid (src/kinematics/KinematicChain.cpp:94-123) sets both arms' joints, updates the
 * robot state and asks MoveIt's PlanningScene::checkCollision under an AllowedCollisionMatrix.  MoveIt, FCL and the
 * robot's URDF meshes stay on the host (out of scope to accelerate); what runs here is the cheap test a
 * planner puts in front of it ("per-arm collision pre-filter (capsule/sphere proxies) ahead of MoveIt
 * isValidImpl"): spheres rigidly attached to link frames of the two arms (or to the world), static oriented boxes (the
 * reference's "sub_table", KinematicChain.cpp:25-30), a 32 x 32 allowed-pair matrix between user-chosen groups (the
 * reference's acm_, KinematicChain.cpp:9,86-91), and for every state the smallest signed distance over all pairs that
 * are not allowed.  With proxies INSCRIBED in the real geometry a negative clearance proves a collision (the state can
 * be dropped without asking MoveIt); with CIRCUMSCRIBED proxies a positive clearance proves freedom.  The proxies are
 * the caller's: the reference ships no link geometry (robot_description comes from the ROS parameter server), so this
 * entry point cannot be — and is not — compared with MoveIt; it is bit-exact against oracle/ccmp_oracle.c:orc_clearance,
 * whose frames are the projector's own forward kinematics.
 *
 * Frames: CCMP_FRAME_WORLD, or CCMP_FRAME(arm, k) with arm 0 / 1 in state order and
 *   k = 0..6  the RBDL body of joint k (panda_link1..7): origin at the joint, axes parallel to the arm base at q = 0
 *             (panda_rbdl.cpp:117-147: the bodies are chained by pure translations)
 *   k = 7     the hand frame PandaModel::getTransform returns (panda_rbdl.cpp:24-42): 0.107 m beyond joint 6, turned -45 deg
 *   k = 8     the arm's base (panda_link0), t_wb of grasping_point.cpp:11-20 */
#define CCMP_FRAME_WORLD (-1)
#define CCMP_FRAME(arm, k) ((arm) * 9 + (k))
#define CCMP_MAX_SPHERES 64
#define CCMP_MAX_BOXES 8
typedef struct ccmp_sphere {
  int32_t frame; /* CCMP_FRAME_WORLD or CCMP_FRAME(arm, k) */
  int32_t group; /* 0..31: row / column of the allowed-pair matrix */
  double c[3];   /* centre in that frame, metres */
  double r;      /* radius, >= 0 */
} ccmp_sphere;
typedef struct ccmp_box { /* static, oriented, in the world frame */
  int32_t group;
  int32_t reserved;
  double c[3];    /* centre */
  double R[9];    /* row-major rotation, box axes -> world */
  double half[3]; /* half extents along the box axes */
} ccmp_box;
typedef struct ccmp_scene ccmp_scene;
/* allowed[g] bit h set = pairs between groups g and h are never tested (either direction counts; NULL = nothing allowed).
 * Never tested either: two proxies on the same frame, and two proxies that are both static (world or an arm's base).
 * Pairs are numbered sphere i < sphere j in the caller's order first, then (sphere i, box b). */
int ccmp_scene_create(ccmp_ctx *ctx, const ccmp_sphere *spheres, int n_spheres, const ccmp_box *boxes, int n_boxes,
                      const uint32_t allowed[32], ccmp_scene **out);
void ccmp_scene_destroy(ccmp_scene *scene);
int ccmp_scene_num_pairs(const ccmp_scene *scene);
/* clearance[i] = min over tested pairs of (distance between centres - r_i - r_j), or (distance from the sphere centre to the
 * box, 0 inside) - r_i; +inf when nothing is tested; NaN for a state with a non-finite joint value.  pair[i] (nullable)
 * = i_sphere | (j << 8) of the first pair in the numbering above that attains it, j = 64 + box index for a box, -1 when
 * there is none.  free_out[i] (nullable) = (ok_in == NULL || ok_in[i]) && clearance[i] > margin — it can go straight into
 * ccmp_compact_valid behind a projection.  Device pointers, asynchronous on hip_stream. */
int ccmp_clearance_batch(ccmp_ctx *ctx, const ccmp_problem *p, const ccmp_scene *scene, const double *q, const uint8_t *ok_in,
                         size_t B, double margin, double *clearance, int32_t *pair, uint8_t *free_out, void *hip_stream);
/* the same on host buffers (a single state: what a StateValidityChecker wrapper calls before MoveIt) */
int ccmp_clearance_host(ccmp_ctx *ctx, const ccmp_problem *p, const ccmp_scene *scene, const double *q, size_t B, double margin,
                        double *clearance, int32_t *pair, uint8_t *free_out);

/* ---- diagnostics ---------------------------------------------------------------------------------- */
/* (test and tool hooks — the device probe of ccmp_detmath.h, an externally supplied processing order, the scout's predictions, fault
 * injection — are not part of this library: include/ccmp_debug.h, lib/libccmp_debug.so) */
const char *ccmp_strerror(int code);
const char *ccmp_last_hip_error(void);
int ccmp_version(void);
size_t ccmp_problem_sizeof(void);

#ifdef __cplusplus
}
#endif
#endif /* CCMP_H */
