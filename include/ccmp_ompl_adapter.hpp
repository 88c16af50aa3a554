// ccmp_ompl_adapter.hpp — header-only C++ host side above the C ABI (include/ccmp.h).
//
// Part 1 (always available, no third-party headers): ccmp::Projector, an RAII owner of a
// ccmp_ctx + ccmp_problem with the reference's method names on raw double[14] buffers;
// ccmp::SampleBuffer / ccmp::RefSampleBuffer, refill-on-empty batches of GPU-projected samples;
// ccmp::discreteGeodesic; ccmp::ShardedProjector (one process, several GPUs, RCCL all-gather of the valid states);
// ccmp::ProxyScene (sphere / box pre-filter ahead of the MoveIt validity test); and
// the dump formats of the reference's planner run (ccmp::printAsMatrix, ccmp::printGraphML, ccmp::printGraphviz).
//
// Part 2 (compiled only with CCMP_WITH_OMPL defined, i.e. inside the reference's catkin workspace where OMPL and Eigen
// exist): drop-in replacements that keep the reference's class names and virtual signatures, so
// src/planner/stefanBiPRM.cpp, src/base/constraints/ConstrainedPlanningCommon.cpp and src/main.cpp compile unchanged
// against them.  They replace the class bodies of TWO headers and the members of ONE source file of the reference:
//   class KinematicChainConstraint : public ompl::base::Constraint, typedef ChainConstraintPtr
//       (include/closed_chain_motion_planner/base/constraints/ConstraintFunction.h:21-140)
//   typedef jy_ProjectedStateSpacePtr, class jy_ProjectedStateSampler : public ompl::base::WrapperStateSampler,
//   class jy_ProjectedStateSpace : public ompl::base::ConstrainedStateSpace,
//   class jy_MotionValidator : public ompl::base::ConstrainedMotionValidator
//       (include/closed_chain_motion_planner/base/jy_ProjectedStateSpace.h:15-69; members: src/base/jy_ProjectedStateSpace.cpp:5-96,
//        which therefore leaves the build)
// The replacement headers and the CMake edit are in include/reference_overlay/ (INTEGRATION.md section 2).
// Neither OMPL nor Eigen is installed in the build image of this repository: part 2 is compiled against an interface mock
// (tests/cpp/mock_ompl) — two translation units including both replacement headers, linked into one program — and run on
// the GPU box by tests/test_cpp_adapter.py; part 1 is compiled and run there as plain C++14.
#ifndef CCMP_OMPL_ADAPTER_HPP
#define CCMP_OMPL_ADAPTER_HPP

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <limits>
#include <mutex>
#include <ostream>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ccmp.h"

namespace ccmp {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &what) : std::runtime_error(what + ": " + ccmp_strerror(c)), code(c) {}
};
inline void check(int rc, const char *what)
{
  if (rc != CCMP_OK) throw Error(rc, what);
}

// SplitMix64 (the generator behind the kernels' counter-based samplers) and a process-wide source of sampler seeds:
// every sampler owns its own stream, as every OMPL StateSampler owns its own ompl::RNG.  CCMP_SEED in the
// environment makes runs reproducible (the counterpart of ompl::RNG::setSeed).
inline uint64_t splitmix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t next_sampler_seed()
{
  static const uint64_t process_seed = [] {
    if (const char *env = std::getenv("CCMP_SEED")) return (uint64_t)std::strtoull(env, nullptr, 0);
    std::random_device rd;
    return ((uint64_t)rd() << 32) ^ (uint64_t)rd();
  }();
  static std::atomic<uint64_t> counter{0};
  return splitmix64(process_seed + counter.fetch_add(1, std::memory_order_relaxed));
}

// Owner of one execution context and one problem description.  The drop-in KinematicChainConstraint is touched from
// more than one thread in the reference (the jy_GoalLazySamples sampling thread; checkForSolution beside
// constructRoadmap, src/planner/stefanBiPRM.cpp:848) and one ccmp_ctx holds per-call state (pinned I/O block, queue
// words, stream): every call that uses the context takes the Projector's mutex for its duration — negligible next to
// a ~100 us launch.  The problem is passed by value into every launch and may be changed between calls.
class Projector {
public:
  explicit Projector(int device = 0) { check(ccmp_ctx_create(device, &ctx_), "ccmp_ctx_create"); std::memset(&problem_, 0, sizeof problem_); }
  Projector(const std::string &yaml_path, int device = 0) : Projector(device) { loadConfig(yaml_path); }
  ~Projector() { ccmp_ctx_destroy(ctx_); }
  Projector(const Projector &) = delete;
  Projector &operator=(const Projector &) = delete;

  // grasping_point::loadConfig + ConstrainedProblem set-up (src/kinematics/grasping_point.cpp:34-65,
  // src/base/constraints/ConstrainedPlanningCommon.cpp:85-132)
  void loadConfig(const std::string &yaml_path)
  {
    check(ccmp_problem_from_yaml(yaml_path.c_str(), &problem_), "ccmp_problem_from_yaml");
    configured_ = true;
  }
  // KinematicChainConstraint::setArmModels (ConstraintFunction.h:122-126): arm1 = the first seven joints, arm2 the other
  // seven — the order given, as the reference stores them (sorting by name is ConstrainedProblem::_setEnvironment's
  // business, ConstrainedPlanningCommon.cpp:89-91).  On a configured problem (the reference calls it after loadConfig,
  // ConstrainedPlanningCommon.cpp:126) only the arm / base-frame fields change: object poses, tolerances, delta / lambda,
  // calibration and mode are kept, init_chain_ and t_o7 are recomputed.
  void setArmModels(const std::string &name1, int index1, const std::string &name2, int index2)
  {
    if (!configured_) {
      const double q0[14] = {0};
      check(ccmp_problem_init(&problem_, name1.c_str(), index1, name2.c_str(), index2, q0, nullptr, nullptr, nullptr, nullptr), "ccmp_problem_init");
      configured_ = true;
    }
    check(ccmp_set_arms(&problem_, name1.c_str(), index1, name2.c_str(), index2), "ccmp_set_arms");
  }
  // ArmModel::t_wb of the arm in `slot` (panda_model.h:15; row-major rotation, translation): what function() multiplies
  // with (ConstraintFunction.h:89-90).  After setArmModels.
  void setBaseFrame(int slot, const double *R9, const double *p3) { check(ccmp_set_base_frame(&problem_, slot, R9, p3), "ccmp_set_base_frame"); }
  void setInitialPosition(const double *init_joint14) { check(ccmp_set_start(&problem_, init_joint14), "ccmp_set_start"); }
  // throws where the reference throws ompl::Exception (ConstraintFunction.h:106-108)
  void setTolerance(double tolerance1, double tolerance2) { check(ccmp_set_tolerance(&problem_, tolerance1, tolerance2), "setTolerance: tolerance must be positive"); }
  void setJacobianMode(int mode) { problem_.jacobian_mode = mode; }
  // The unchanged planner calls the constraint one state at a time (project(State*), isSatisfied(State*):
  // src/base/jy_ProjectedStateSpace.cpp:13,20,27,65, src/planner/stefanBiPRM.cpp:397-398), and each such call is a kernel launch
  // plus a completion poll.  setResident(true) turns on the context's resident service kernel (include/ccmp.h, option "resident"):
  // the same calls, the same bits, no launch on the call path (project(x) ~10 us less, isSatisfied 21 -> 9 us).  Opt-in because a
  // kernel that stays on the device makes a device-wide synchronise of the APPLICATION's own (hipDeviceSynchronize, hipFree) wait
  // until the service has idled out ("resident_idle_ms", default 10 ms); the library's own calls stop it first.
  void setResident(bool on)
  {
    std::lock_guard<std::mutex> hold(mu_);
    check(ccmp_ctx_set_option(ctx_, "resident", on ? 1 : 0), "ccmp_ctx_set_option(resident)");
  }
  // what the scheduling policy does with a batched call of n samples / edges (CCMP_CALL_*), in one line — ccmp_ctx_describe
  std::string describe(int call_kind, size_t n) const
  {
    char buf[2048];
    const int len = ccmp_ctx_describe(ctx_, call_kind, n, buf, sizeof buf);
    return len < 0 ? std::string() : std::string(buf);
  }

  // bool KinematicChainConstraint::project(Eigen::Ref<VectorXd> x) const — in place
  bool project(double *x14) const
  {
    uint8_t ok = 0;
    std::lock_guard<std::mutex> hold(mu_);
    check(ccmp_project_host(ctx_, &problem_, x14, x14, &ok, nullptr, 1), "ccmp_project_host");
    return ok != 0;
  }
  void function(const double *x14, double *out2) const
  {
    std::lock_guard<std::mutex> hold(mu_);
    check(ccmp_function_host(ctx_, &problem_, x14, out2, 1), "ccmp_function_host");
  }
  bool isSatisfied(const double *x14) const
  {
    uint8_t ok = 0;
    std::lock_guard<std::mutex> hold(mu_);
    check(ccmp_is_satisfied_host(ctx_, &problem_, x14, &ok, 1), "ccmp_is_satisfied_host");
    return ok != 0;
  }
  bool jointValid(const double *x14) const
  {
    uint8_t ok = 0;
    std::lock_guard<std::mutex> hold(mu_);
    check(ccmp_joint_valid_host(ctx_, &problem_, x14, &ok, 1), "ccmp_joint_valid_host");
    return ok != 0;
  }
  // batches on host buffers (q row-major [B][14])
  void projectBatch(const double *q_in, double *q_out, uint8_t *ok, uint16_t *iters, size_t B) const
  {
    std::lock_guard<std::mutex> hold(mu_);
    check(ccmp_project_host(ctx_, &problem_, q_in, q_out, ok, iters, B), "ccmp_project_host");
  }
  unsigned getCoDimension() const { return 2; }
  unsigned getAmbientDimension() const { return 14; }
  const ccmp_problem &problem() const { return problem_; }
  ccmp_problem &problem() { return problem_; }
  ccmp_ctx *ctx() const { return ctx_; }
  // for callers that use ctx() directly (the buffers and discreteGeodesic below): hold this across the call
  std::mutex &mutex() const { return mu_; }

private:
  mutable std::mutex mu_;
  ccmp_ctx *ctx_ = nullptr;
  ccmp_problem problem_;
  bool configured_ = false;
};

// jy_ProjectedStateSampler::sampleUniform served from a buffer that ONE launch of `batch` fused
// sample -> project -> enforceBounds refills when empty (src/base/jy_ProjectedStateSpace.cpp:10-15).
// The stream of samples depends on (seed, running index) only, not on `batch`.
class SampleBuffer {
public:
  SampleBuffer(const Projector &proj, uint64_t seed, size_t batch = 4096) : proj_(proj), seed_(seed), batch_(batch) {}
  // writes 14 doubles; returns project()'s result for that sample (the reference ignores it)
  bool next(double *state14)
  {
    if (pos_ >= filled_) refill();
    std::memcpy(state14, &buf_[14 * pos_], 14 * sizeof(double));
    return ok_[pos_++] != 0;
  }

private:
  void refill()
  {
    filled_ = 0;  // a refill that throws leaves the buffer EMPTY (the next call tries again with the same indices), never half-valid
    pos_ = 0;
    buf_.resize(batch_ * 14);
    ok_.resize(batch_);
    std::lock_guard<std::mutex> hold(proj_.mutex());
    check(ccmp_sample_project_host(proj_.ctx(), &proj_.problem(), seed_, next_index_, buf_.data(), ok_.data(), nullptr, batch_),
          "ccmp_sample_project_host");
    next_index_ += batch_;
    filled_ = batch_;
  }
  const Projector &proj_;
  uint64_t seed_, next_index_ = 0;
  size_t batch_, pos_ = 0, filled_ = 0;
  std::vector<double> buf_;
  std::vector<uint8_t> ok_;
};

// jy_ProjectedStateSampler::sampleUniformNear / sampleGaussian (src/base/jy_ProjectedStateSpace.cpp:17-29) through the
// batched counter-based kernels.  The reference state may change from call to call, so nothing can be drawn ahead of
// the call; but `lookahead` samples around one state cost what one does (a 128-thread block per sample on an
// otherwise idle GPU), so a call with a new (state, parameter) draws `lookahead` samples in one launch and the
// following calls with the same arguments pop from that buffer.  Every refill takes fresh indices of the sampler's
// stream: no sample is handed out twice, whatever the call pattern.
class RefSampleBuffer {
public:
  enum Kind { Near = 0, Gaussian = 1 };
  RefSampleBuffer(const Projector &proj, uint64_t seed, Kind kind, size_t lookahead = 32)
    : proj_(proj), seed_(seed ^ (kind == Near ? 0x4E454152ull : 0x47415553ull)), kind_(kind), lookahead_(lookahead)
  {
  }
  // writes 14 doubles: a sample around ref14 (Near: within `param` per dimension; Gaussian: std dev `param`),
  // projected and wrapped by enforceBounds; returns project()'s result
  bool next(double *state14, const double *ref14, double param)
  {
    if (pos_ >= filled_ || param != param_ || std::memcmp(ref_, ref14, sizeof ref_) != 0) refill(ref14, param);
    std::memcpy(state14, &buf_[14 * pos_], 14 * sizeof(double));
    return ok_[pos_++] != 0;
  }

private:
  void refill(const double *ref14, double param)
  {
    filled_ = 0;  // a refill that throws leaves the buffer EMPTY, never filled with samples around another state
    pos_ = 0;
    std::memcpy(ref_, ref14, sizeof ref_);
    param_ = param;
    buf_.resize(lookahead_ * 14);
    ok_.resize(lookahead_);
    std::lock_guard<std::mutex> hold(proj_.mutex());
    check(ccmp_sample_ref_project_host(proj_.ctx(), &proj_.problem(), (int)kind_, seed_, next_index_, ref_, param_, buf_.data(), ok_.data(),
                                       nullptr, lookahead_),
          "ccmp_sample_ref_project_host");
    next_index_ += lookahead_;
    filled_ = lookahead_;
    pos_ = 0;
  }
  const Projector &proj_;
  uint64_t seed_, next_index_ = 0;
  Kind kind_;
  size_t lookahead_, pos_ = 0, filled_ = 0;
  double ref_[14] = {0}, param_ = 0;
  std::vector<double> buf_;
  std::vector<uint8_t> ok_;
};

// jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:32-96) for E edges in ONE launch on raw
// buffers (from / to: E x 14, row-major): growTree tries its (up to five) nearest neighbours one after the other
// (src/planner/stefanBiPRM.cpp:307-351, each a discreteGeodesic(neighbour, new vertex, false, &states)); handed over
// together they cost the longest edge's serial chain instead of the sum.  `valid` is the host StateValidityChecker; it is
// consulted only when !interpolate, and it sees the same states in the same order as in the reference's loop — edge by
// edge, state by state, an edge stopping at its first rejected state, where the reference's loop breaks.  reached[e] is
// the bool of edge e, (*geodesics)[e] its list.
// The traversal on the GPU lists max_states states per call (edges of the reference's roadmaps have 3-7; the work spent
// beyond a state the checker rejects is bounded by this).  An edge that needs more is CONTINUED from its last stored
// state (ccmp_geodesic_host_ex: the states of one uninterrupted traversal, bit for bit) — after the checker has passed
// what is there, so that a long or creeping edge the checker cuts early costs nothing more.  A cut list never reaches
// the caller as if it were complete.
// check_target: ConstrainedMotionValidator::checkMotion in one launch — isSatisfied(to) is tested first and an edge whose
// target fails it returns false with only `from` in the list (src/planner/stefanBiPRM.cpp:397-398).
// delta / lambda > 0 override the problem's values for this call (the space's setDelta / setLambda) through a per-call
// copy of the problem: nothing shared is written, whichever thread calls.
template <class ValidFn>
inline void discreteGeodesicBatch(const Projector &proj, const double *from, const double *to, size_t E, bool interpolate, ValidFn valid,
                                  std::vector<std::vector<std::vector<double>>> *geodesics, std::vector<char> *reached, int max_states = 64,
                                  bool check_target = false, double delta = -1.0, double lambda = -1.0)
{
  if (reached) reached->assign(E, 0);
  if (geodesics) geodesics->assign(E, {});
  if (E == 0) return;
  if (max_states < 2) max_states = 2; // a continuation starts from a stored state other than `from`
  // A large batch takes the shape the library is fastest with: a first pass with lists of 16 states and at
  // most 128 Newton rounds per edge — one launch that no creeping edge can hold — and the few edges that stop short
  // (list full: n = cap + 1; rounds spent: ok = 2) are continued below, which gives the states of one uninterrupted
  // traversal bit for bit.  growTree's handful of neighbours goes through unchanged.
  const bool big = E >= 1024;
  const int first_cap = big && max_states > 16 ? 16 : max_states;
  const int first_budget = big ? 128 : 0;
  ccmp_problem pb;
  std::vector<double> states(E * (size_t)first_cap * 14), carry(E * 2);
  std::vector<int32_t> n(E);
  std::vector<uint8_t> ok(E);
  {
    std::lock_guard<std::mutex> hold(proj.mutex());
    pb = proj.problem();
    if (delta > 0) pb.delta = delta;
    if (lambda > 0) pb.lambda = lambda;
    check(ccmp_geodesic_host_ex(proj.ctx(), &pb, from, to, E, first_cap, states.data(), n.data(), ok.data(), nullptr, carry.data(),
                                first_budget, check_target ? 1 : 0),
          "ccmp_geodesic_host_ex");
  }
  std::vector<double> more((size_t)max_states * 14);
  for (size_t e = 0; e < E; ++e) {
    std::vector<std::vector<double>> list;
    const double *st = &states[e * (size_t)first_cap * 14];
    const double *to_e = to + 14 * e;
    double cr[2] = {carry[2 * e], carry[2 * e + 1]};
    int32_t ne = n[e];
    uint8_t oke = ok[e];
    bool good = false, cut = false;
    std::vector<double> last(14);
    int first = 0; // a continuation's row 0 repeats the state it started from
    int cap = first_cap;
    for (;;) {
      const int have = ne > cap ? cap : ne;
      for (int k = first; k < have && !cut; ++k) {
        const double *row = st + (size_t)k * 14;
        if (!interpolate && !(list.empty() && k == 0) && !valid(row)) {
          // the reference's loop breaks here, before `dist` is updated: the answer is about the last accepted state
          double d = 0;
          for (int i = 0; i < 14; ++i) { const double df = last[i] - to_e[i]; d += df * df; }
          good = std::sqrt(d) <= pb.delta;
          cut = true;
          break;
        }
        last.assign(row, row + 14);
        list.emplace_back(row, row + 14);
      }
      if (cut) break;
      if (ne <= cap && oke != 2) { good = oke != 0; break; }
      // ne == cap + 1: the list was full and the traversal stopped there; ok == 2: it had spent the call's Newton rounds —
      // either way go on from its last state
      double cr_out[2];
      {
        std::lock_guard<std::mutex> hold(proj.mutex());
        check(ccmp_geodesic_host_ex(proj.ctx(), &pb, last.data(), to_e, 1, max_states, more.data(), &ne, &oke, cr, cr_out, 0, 0),
              "ccmp_geodesic_host_ex(continue)");
      }
      cr[0] = cr_out[0];
      cr[1] = cr_out[1];
      st = more.data();
      cap = max_states;
      first = 1;
    }
    if (reached) (*reached)[e] = good ? 1 : 0;
    if (geodesics) (*geodesics)[e] = std::move(list);
  }
}

// one edge
template <class ValidFn>
inline bool discreteGeodesic(const Projector &proj, const double *from14, const double *to14, bool interpolate, ValidFn valid,
                             std::vector<std::vector<double>> *geodesic, int max_states = 64, bool check_target = false,
                             double delta = -1.0, double lambda = -1.0)
{
  std::vector<std::vector<std::vector<double>>> lists;
  std::vector<char> reached;
  discreteGeodesicBatch(proj, from14, to14, 1, interpolate, valid, geodesic ? &lists : nullptr, &reached, max_states, check_target, delta,
                        lambda);
  if (geodesic) *geodesic = std::move(lists[0]);
  return reached[0] != 0;
}

// One planner process, several GPUs (the reference's shape: src/main.cpp is one process): one context per device and an
// RCCL communicator over them.  sampleProjectSharded / projectSharded spread a batch over the GPUs in contiguous shards,
// every GPU compacts its valid states into a fixed-capacity block and ONE all-gather brings them together; the valid
// states come back in global sample order — what the host tree consumes.  Results are bit-identical to one GPU.
class ShardedProjector {
public:
  // devices: one entry per GPU (RCCL wants distinct devices)
  ShardedProjector(const std::string &yaml_path, const std::vector<int> &devices)
  {
    check(ccmp_problem_from_yaml(yaml_path.c_str(), &problem_), "ccmp_problem_from_yaml");
    try {
      for (int d : devices) {
        ccmp_ctx *c = nullptr;
        check(ccmp_ctx_create(d, &c), "ccmp_ctx_create");
        ctxs_.push_back(c);
      }
      check(ccmp_comm_create(ctxs_.data(), (int)ctxs_.size(), &comm_), "ccmp_comm_create");
    } catch (...) {
      release();
      throw;
    }
  }
  ~ShardedProjector() { release(); }
  ShardedProjector(const ShardedProjector &) = delete;
  ShardedProjector &operator=(const ShardedProjector &) = delete;
  ccmp_problem &problem() { return problem_; }
  size_t devices() const { return ctxs_.size(); }

  // jy_ProjectedStateSampler::sampleUniform x B over the GPUs: returns the valid states (row-major [n][14], global sample
  // order); counts (optional) receives the valid states per shard.  A shard holding more valid states than the block
  // (block_fraction of the shard; ~22 % of uniform samples are valid) is retried once with full-size blocks.
  std::vector<double> sampleProjectSharded(uint64_t seed, uint64_t first_index, size_t B, std::vector<uint64_t> *counts = nullptr,
                                           double block_fraction = 0.5)
  {
    return run(nullptr, seed, first_index, B, counts, block_fraction);
  }
  // KinematicChainConstraint::project x B (host buffer q_in [B][14]) over the GPUs: the valid projected states
  std::vector<double> projectSharded(const double *q_in, size_t B, std::vector<uint64_t> *counts = nullptr, double block_fraction = 0.5)
  {
    return run(q_in, 0, 0, B, counts, block_fraction);
  }

private:
  std::vector<double> run(const double *q_in, uint64_t seed, uint64_t first, size_t B, std::vector<uint64_t> *counts, double fraction)
  {
    std::lock_guard<std::mutex> hold(mu_);
    const size_t n = ctxs_.size(), shard = (B + n - 1) / n;
    std::vector<uint64_t> cnt(n, 0);
    for (int attempt = 0; attempt < 2; ++attempt) {
      const size_t rows = attempt == 0 ? std::max<size_t>(1, (size_t)(shard * fraction)) : std::max<size_t>(1, shard);
      std::vector<double> valid(n * rows * 14);
      uint64_t nv = 0;
      const int rc = q_in ? ccmp_project_sharded(comm_, &problem_, q_in, B, nullptr, nullptr, nullptr, rows, valid.data(), n * rows, cnt.data(), &nv)
                          : ccmp_sample_project_sharded(comm_, &problem_, seed, first, B, nullptr, nullptr, nullptr, rows, valid.data(), n * rows,
                                                        cnt.data(), &nv);
      if (rc == CCMP_EOVERFLOW && attempt == 0) continue;
      check(rc, "ccmp_project_sharded");
      valid.resize((size_t)nv * 14);
      if (counts) *counts = cnt;
      return valid;
    }
    return {};
  }
  void release()
  {
    if (comm_) ccmp_comm_destroy(comm_);
    comm_ = nullptr;
    for (ccmp_ctx *c : ctxs_) ccmp_ctx_destroy(c);
    ctxs_.clear();
  }
  std::vector<ccmp_ctx *> ctxs_;
  ccmp_comm *comm_ = nullptr;
  ccmp_problem problem_;
  std::mutex mu_;
};

// ---- proxy-geometry pre-filter in front of KinematicChainValidityChecker::isValid ---------------------------------
// (src/kinematics/KinematicChain.cpp:94-123: MoveIt's checkCollision under acm_).  Spheres on link frames, static
// boxes, an allowed-pair matrix (include/ccmp.h: ccmp_scene_create); clearance() is the smallest signed distance over
// the tested pairs.  With spheres inscribed in the links a negative clearance proves a collision, so the state can be
// refused without asking MoveIt; everything else still goes to MoveIt (PrefilteredValidityChecker in part 2).
// The scene borrows the Projector (context, problem, mutex): keep the Projector alive for as long as the scene is used.
class ProxyScene {
public:
  ProxyScene(const Projector &proj, const std::vector<ccmp_sphere> &spheres, const std::vector<ccmp_box> &boxes,
             const uint32_t *allowed32 = nullptr)
    : proj_(proj)
  {
    check(ccmp_scene_create(proj.ctx(), spheres.data(), (int)spheres.size(), boxes.data(), (int)boxes.size(), allowed32, &scene_),
          "ccmp_scene_create");
  }
  ~ProxyScene() { ccmp_scene_destroy(scene_); }
  ProxyScene(const ProxyScene &) = delete;
  ProxyScene &operator=(const ProxyScene &) = delete;
  int numPairs() const { return ccmp_scene_num_pairs(scene_); }
  // one state; pair (nullable) = i | j << 8 of the closest pair (j >= 64: box j - 64), -1 if nothing is tested
  double clearance(const double *x14, int32_t *pair = nullptr) const
  {
    double clr = 0.0;
    std::lock_guard<std::mutex> hold(proj_.mutex());
    check(ccmp_clearance_host(proj_.ctx(), &proj_.problem(), scene_, x14, 1, 0.0, &clr, pair, nullptr), "ccmp_clearance_host");
    return clr;
  }
  // B host states: clearance[B], free[B] = clearance > margin (both caller-owned; pair nullable)
  void clearanceBatch(const double *q, size_t B, double margin, double *clearance, int32_t *pair, uint8_t *free_out) const
  {
    std::lock_guard<std::mutex> hold(proj_.mutex());
    check(ccmp_clearance_host(proj_.ctx(), &proj_.problem(), scene_, q, B, margin, clearance, pair, free_out), "ccmp_clearance_host");
  }
  const ccmp_scene *handle() const { return scene_; }

  // AllowedCollisionMatrix::setEntry(g, h, true) on a 32-word matrix
  static void allow(uint32_t *allowed32, int g, int h)
  {
    allowed32[g] |= 1u << h;
    allowed32[h] |= 1u << g;
  }
  // the reference constructor's "sub_table" (KinematicChain.cpp:25-30: addBox(dim (0.65, 1.0, 0.2), pose (0.65, 0, 1.1)))
  static ccmp_box subTable(int group)
  {
    ccmp_box b;
    std::memset(&b, 0, sizeof b);
    b.group = group;
    b.c[0] = 0.65; b.c[1] = 0.0; b.c[2] = 1.1;
    b.R[0] = b.R[4] = b.R[8] = 1.0;
    b.half[0] = 0.325; b.half[1] = 0.5; b.half[2] = 0.1;
    return b;
  }

private:
  const Projector &proj_;
  ccmp_scene *scene_ = nullptr;
};

// ---- dump formats of the reference's planner run ----------------------------------------------------------------------
// PathGeometric::printAsMatrix as ConstrainedProblem::solveOnce writes `<obj>_path.txt`
// (src/base/constraints/ConstrainedPlanningCommon.cpp:219-222) and scripts/execute_path.py:65-87 / visualize_path.py
// parse it: the reals of one state per line in default stream format, each followed by a space, and one empty line
// after the last state.
inline void printAsMatrix(std::ostream &out, const double *states, size_t n_states, size_t dim = 14)
{
  for (size_t i = 0; i < n_states; ++i) {
    for (size_t j = 0; j < dim; ++j) out << states[i * dim + j] << ' ';
    out << '\n';
  }
  out << '\n';
  out.flush();
}

// PlannerData::printGraphML as ConstrainedProblem::dumpGraph writes `<obj>_node_info.graphml`
// (include/closed_chain_motion_planner/base/constraints/ConstrainedPlanningCommon.h:73-87): node data = the reals of
// the milestone joined by commas, directed edges in insertion order with their weight (nullptr = 1 each).
inline void printGraphML(std::ostream &out, const double *nodes, size_t n_nodes, const std::pair<unsigned, unsigned> *edges,
                         size_t n_edges, const double *weights = nullptr, size_t dim = 14)
{
  out << "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n"
         "<graphml xmlns=\"http://graphml.graphdrawing.org/xmlns\" xmlns:xsi=\"http://www.w3.org/2001/XMLSchema-instance\" "
         "xsi:schemaLocation=\"http://graphml.graphdrawing.org/xmlns http://graphml.graphdrawing.org/xmlns/1.0/graphml.xsd\">\n"
         "  <key id=\"key0\" for=\"node\" attr.name=\"coords\" attr.type=\"string\" />\n"
         "  <key id=\"key1\" for=\"edge\" attr.name=\"weight\" attr.type=\"double\" />\n"
         "  <graph id=\"G\" edgedefault=\"directed\" parse.nodeids=\"free\" parse.edgeids=\"canonical\" parse.order=\"nodesfirst\">\n";
  for (size_t i = 0; i < n_nodes; ++i) {
    out << "    <node id=\"n" << i << "\">\n      <data key=\"key0\">";
    for (size_t j = 0; j < dim; ++j) out << (j ? "," : "") << nodes[i * dim + j];
    out << "</data>\n    </node>\n";
  }
  for (size_t k = 0; k < n_edges; ++k)
    out << "    <edge id=\"e" << k << "\" source=\"n" << edges[k].first << "\" target=\"n" << edges[k].second
        << "\">\n      <data key=\"key1\">" << (weights ? weights[k] : 1.0) << "</data>\n    </edge>\n";
  out << "  </graph>\n</graphml>\n";
  out.flush();
}

// PlannerData::printGraphviz as dumpGraph writes `<obj>_graph_info.dot`
inline void printGraphviz(std::ostream &out, size_t n_nodes, const std::pair<unsigned, unsigned> *edges, size_t n_edges)
{
  out << "digraph G {\n";
  for (size_t i = 0; i < n_nodes; ++i) out << i << ";\n";
  for (size_t k = 0; k < n_edges; ++k) out << edges[k].first << "->" << edges[k].second << " ;\n";
  out << "}\n";
  out.flush();
}

}  // namespace ccmp

#ifdef CCMP_WITH_OMPL
// ---------------------------------------------------------------------------------------------------
// Part 2: the reference's classes, same names and signatures, backed by ccmp::Projector.
// ---------------------------------------------------------------------------------------------------
#include <Eigen/Dense>
#include <ompl/base/Constraint.h>
#include <ompl/base/StateSampler.h>
#include <ompl/base/spaces/constraint/ConstrainedStateSpace.h>
#include <ompl/util/Exception.h>

#include <closed_chain_motion_planner/kinematics/panda_model.h>  // ArmModelPtr {name, index, ...}

class KinematicChainConstraint : public ompl::base::Constraint {
public:
  explicit KinematicChainConstraint(unsigned int links, int device = 0) : ompl::base::Constraint(links, 2), impl_(new ccmp::Projector(device)) {}
  // the State* overloads of the base class stay visible next to the overrides below (the samplers and
  // discreteGeodesic call project(State*) / isSatisfied(const State*), src/base/jy_ProjectedStateSpace.cpp:13,20,27,65)
  using ompl::base::Constraint::project;
  using ompl::base::Constraint::isSatisfied;

  // ConstraintFunction.h:122-126: the two ArmModels in the order given; the base frame of each is the t_wb the ArmModel
  // carries (panda_model.h:15, filled from config->t_wb[index] at ConstrainedPlanningCommon.cpp:98) — not a table of this
  // library's — so function() multiplies with the very frame the reference's does (ConstraintFunction.h:89-90)
  void setArmModels(const ArmModelPtr &arm1, const ArmModelPtr &arm2)
  {
    impl_->setArmModels(arm1->name, arm1->index, arm2->name, arm2->index);
    const ArmModelPtr arms[2] = {arm1, arm2};
    for (int a = 0; a < 2; ++a) {
      const Eigen::Isometry3d &t_wb = arms[a]->t_wb;
      double R[9], p[3];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R[3 * r + c] = t_wb.linear()(r, c);
        p[r] = t_wb.translation()(r);
      }
      impl_->setBaseFrame(a, R, p);
    }
  }
  void setInitialPosition(const Eigen::Ref<const Eigen::VectorXd> init_joint)
  {
    Eigen::VectorXd q = init_joint;  // contiguous copy
    impl_->setInitialPosition(q.data());
  }
  // opt-in (not part of the reference's class): the context's resident service kernel for the one-state-at-a-time calls below
  // — same results, ~10 us less per call; see ccmp::Projector::setResident for what it asks of the application
  void setResident(bool on) { impl_->setResident(on); }
  void setTolerance(const double tolerance1, const double tolerance2)
  {
    try { impl_->setTolerance(tolerance1, tolerance2); }
    catch (const ccmp::Error &) {
      throw ompl::Exception("ompl::base::Constraint::setProjectionTolerance(): tolerance must be positive.");
    }
  }
  // The reference's overrides return plain bool / void and are called from the planner's main thread AND its solution-checker
  // thread (src/planner/stefanBiPRM.cpp:848-849): nothing may be thrown out of them — a transient HIP error would otherwise
  // std::terminate the planner.  A failed call keeps the reference's contract for "no": project / isSatisfied / jointValid return
  // false and leave the state as it was, function writes NaN (so that isSatisfied(f) is false); the error stays readable through
  // lastError() / lastErrorMessage() until clearError().  setTolerance above remains the only thrower, as in the reference
  // (ConstraintFunction.h:104-108).
  bool project(Eigen::Ref<Eigen::VectorXd> x) const override
  {
    double buf[14];
    for (int i = 0; i < 14; ++i) buf[i] = x[i];
    bool ok = false;
    if (!guarded([&] { ok = impl_->project(buf); })) return false;  // x untouched
    for (int i = 0; i < 14; ++i) x[i] = buf[i];
    return ok;
  }
  void function(const Eigen::Ref<const Eigen::VectorXd> &x, Eigen::Ref<Eigen::VectorXd> out) const override
  {
    double buf[14], f[2] = {std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN()};
    for (int i = 0; i < 14; ++i) buf[i] = x[i];
    guarded([&] { impl_->function(buf, f); });
    out[0] = f[0];
    out[1] = f[1];
  }
  bool isSatisfied(const Eigen::Ref<const Eigen::VectorXd> &x) const override
  {
    double buf[14];
    for (int i = 0; i < 14; ++i) buf[i] = x[i];
    bool ok = false;
    return guarded([&] { ok = impl_->isSatisfied(buf); }) && ok;
  }
  bool jointValid(const Eigen::Ref<const Eigen::VectorXd> &q) const
  {
    double buf[14];
    for (int i = 0; i < 14; ++i) buf[i] = q[i];
    bool ok = false;
    return guarded([&] { ok = impl_->jointValid(buf); }) && ok;
  }
  // 0 (CCMP_OK) or the code of the first error since the last clearError() (CCMP_EHIP, CCMP_ENODEV, ...) — sticky, thread-safe
  int lastError() const
  {
    std::lock_guard<std::mutex> hold(err_mu_);
    return err_code_;
  }
  std::string lastErrorMessage() const
  {
    std::lock_guard<std::mutex> hold(err_mu_);
    return err_what_;
  }
  void clearError() const
  {
    std::lock_guard<std::mutex> hold(err_mu_);
    err_code_ = CCMP_OK;
    err_what_.clear();
  }
  // runs `body`; an exception of the library (or any other) is recorded, never passed on: false = it failed.  Used by the space and
  // the samplers below for their own GPU calls too.
  template <class F>
  bool guarded(F &&body) const noexcept
  {
    try {
      body();
      return true;
    } catch (const ccmp::Error &e) {
      record(e.code, e.what());
    } catch (const std::exception &e) {
      record(CCMP_EHIP, e.what());
    } catch (...) {
      record(CCMP_EHIP, "unknown exception");
    }
    return false;
  }
  ccmp::Projector &impl() const { return *impl_; }

private:
  void record(int code, const char *what) const noexcept
  {
    try {
      std::lock_guard<std::mutex> hold(err_mu_);
      if (err_code_ == CCMP_OK) { err_code_ = code; err_what_ = what; }
    } catch (...) {
    }
  }
  std::shared_ptr<ccmp::Projector> impl_;
  mutable std::mutex err_mu_;
  mutable int err_code_ = CCMP_OK;
  mutable std::string err_what_;
};
typedef std::shared_ptr<KinematicChainConstraint> ChainConstraintPtr;

// jy_ProjectedStateSampler (include/closed_chain_motion_planner/base/jy_ProjectedStateSpace.h:18-29): same
// name, same overrides; all three draw from the GPU's counter-based samplers through refill-on-empty buffers
// (sampleUniform: `batch` samples per launch; Near / Gaussian: a look-ahead around the current reference state).
// Every sampler has its own stream: the space hands out seed = splitmix64(space seed + running sampler number),
// as every OMPL sampler owns an independently seeded ompl::RNG.
class jy_ProjectedStateSpace;
typedef std::shared_ptr<jy_ProjectedStateSpace> jy_ProjectedStateSpacePtr;

class jy_ProjectedStateSampler : public ompl::base::WrapperStateSampler {
public:
  jy_ProjectedStateSampler(const jy_ProjectedStateSpace *space, ompl::base::StateSamplerPtr sampler);
  jy_ProjectedStateSampler(const jy_ProjectedStateSpace *space, ompl::base::StateSamplerPtr sampler, uint64_t seed);
  void sampleUniform(ompl::base::State *state) override
  {
    auto &&x = *state->as<ompl::base::ConstrainedStateSpace::StateType>();
    double buf[14];
    if (!constraint_->guarded([&] { buffer_.next(buf); })) return;  // already projected and wrapped by enforceBounds on the GPU; a failed refill leaves the state as it was (lastError() of the constraint)
    for (int i = 0; i < 14; ++i) x[i] = buf[i];
  }
  void sampleUniformNear(ompl::base::State *state, const ompl::base::State *near, const double distance) override
  {
    sampleAround(near_, state, near, distance);
  }
  void sampleGaussian(ompl::base::State *state, const ompl::base::State *mean, const double stdDev) override
  {
    sampleAround(gauss_, state, mean, stdDev);
  }

protected:
  void sampleAround(ccmp::RefSampleBuffer &buf, ompl::base::State *state, const ompl::base::State *ref, double param)
  {
    const auto &r = *ref->as<ompl::base::ConstrainedStateSpace::StateType>();
    auto &&x = *state->as<ompl::base::ConstrainedStateSpace::StateType>();
    double rb[14], out[14];
    for (int i = 0; i < 14; ++i) rb[i] = r[i];
    if (!constraint_->guarded([&] { buf.next(out, rb, param); })) return;  // ambient draw, project (result ignored, as the reference does) and enforceBounds on the GPU
    for (int i = 0; i < 14; ++i) x[i] = out[i];
  }
  const std::shared_ptr<KinematicChainConstraint> constraint_;
  ccmp::SampleBuffer buffer_;
  ccmp::RefSampleBuffer near_, gauss_;
};

// jy_ProjectedStateSpace (include/closed_chain_motion_planner/base/jy_ProjectedStateSpace.h:31-54): same name,
// same overrides.  discreteGeodesic runs the traversal on the GPU and applies the space information's
// StateValidityChecker on the host exactly where the reference consults it (src/base/jy_ProjectedStateSpace.cpp:
// 65-68); checkMotion of OMPL's ConstrainedMotionValidator (isSatisfied(s2) && discreteGeodesic(s1, s2),
// src/planner/stefanBiPRM.cpp:397-398,463-464) therefore needs no change.
class jy_ProjectedStateSpace : public ompl::base::ConstrainedStateSpace {
public:
  jy_ProjectedStateSpace(const ompl::base::StateSpacePtr &ambientSpace, const ompl::base::ConstraintPtr &constraint)
    : ompl::base::ConstrainedStateSpace(ambientSpace, constraint), chain_(std::dynamic_pointer_cast<KinematicChainConstraint>(constraint)),
      seed_(ccmp::next_sampler_seed())
  {
    setName("Projected" + space_->getName());
  }
  // the seed of the next sampler of this space: splitmix64(space seed + n++)
  uint64_t nextSamplerSeed() const { return ccmp::splitmix64(seed_ + samplers_.fetch_add(1, std::memory_order_relaxed)); }
  ~jy_ProjectedStateSpace() override = default;
  ompl::base::StateSamplerPtr allocDefaultStateSampler() const override
  {
    return std::make_shared<jy_ProjectedStateSampler>(this, space_->allocDefaultStateSampler(), nextSamplerSeed());
  }
  ompl::base::StateSamplerPtr allocStateSampler() const override
  {
    return std::make_shared<jy_ProjectedStateSampler>(this, space_->allocStateSampler(), nextSamplerSeed());
  }
  bool discreteGeodesic(const ompl::base::State *from, const ompl::base::State *to, bool interpolate = false,
                        std::vector<ompl::base::State *> *geodesic = nullptr) const override
  {
    return traverse(from, to, interpolate, geodesic, false);
  }
  // checkMotion's two tests — isSatisfied(s2) && discreteGeodesic(s1, s2, false) — in one GPU launch
  bool checkMotion(const ompl::base::State *s1, const ompl::base::State *s2) const { return traverse(s1, s2, false, nullptr, true); }

  // growTree's neighbour loop (src/planner/stefanBiPRM.cpp:307-351) in one launch: discreteGeodesic(from[e], to, interpolate,
  // &(*geodesics)[e]) for every e; reached[e] is the bool the reference's call returns.  The StateValidityChecker is asked
  // about the same states in the same order as by that loop.
  void discreteGeodesics(const std::vector<const ompl::base::State *> &from, const ompl::base::State *to, bool interpolate,
                         std::vector<std::vector<ompl::base::State *>> *geodesics, std::vector<char> *reached) const
  {
    const size_t E = from.size();
    std::vector<double> a(E * 14), b(E * 14);
    const auto &tb = *to->as<StateType>();
    for (size_t e = 0; e < E; ++e) {
      const auto &fa = *from[e]->as<StateType>();
      for (int i = 0; i < 14; ++i) { a[14 * e + i] = fa[i]; b[14 * e + i] = tb[i]; }
    }
    ccmp::Projector &proj = chain_->impl();
    auto &&svc = si_->getStateValidityChecker();
    std::vector<std::vector<std::vector<double>>> lists;
    ompl::base::State *scratch = allocState();
    const bool done = chain_->guarded([&] {
      ccmp::discreteGeodesicBatch(proj, a.data(), b.data(), E, interpolate,
                                  [&](const double *q) {
                                    auto &x = *scratch->as<StateType>();
                                    for (int i = 0; i < 14; ++i) x[i] = q[i];
                                    return svc->isValid(scratch);
                                  },
                                  geodesics ? &lists : nullptr, reached, 64, false, delta_, lambda_); // setDelta / setLambda of the base class
    });
    freeState(scratch);
    if (!done) {  // the GPU call failed (KinematicChainConstraint::lastError()): no edge was extended
      if (geodesics) geodesics->assign(E, {});
      if (reached) reached->assign(E, 0);
      return;
    }
    if (geodesics) {
      geodesics->assign(E, {});
      for (size_t e = 0; e < E; ++e)
        for (const auto &st : lists[e]) {
          ompl::base::State *s = allocState();
          auto &x = *s->as<StateType>();
          for (int i = 0; i < 14; ++i) x[i] = st[i];
          (*geodesics)[e].push_back(s);
        }
    }
  }

private:
  bool traverse(const ompl::base::State *from, const ompl::base::State *to, bool interpolate, std::vector<ompl::base::State *> *geodesic,
                bool check_target) const
  {
    double a[14], b[14];
    const auto &fa = *from->as<StateType>();
    const auto &tb = *to->as<StateType>();
    for (int i = 0; i < 14; ++i) { a[i] = fa[i]; b[i] = tb[i]; }
    ccmp::Projector &proj = chain_->impl();
    auto &&svc = si_->getStateValidityChecker();
    std::vector<std::vector<double>> states;
    ompl::base::State *scratch = allocState();
    bool ok = false;
    const bool done = chain_->guarded([&] {
      ok = ccmp::discreteGeodesic(proj, a, b, interpolate,
                                  [&](const double *q) {
                                    auto &x = *scratch->as<StateType>();
                                    for (int i = 0; i < 14; ++i) x[i] = q[i];
                                    return svc->isValid(scratch);
                                  },
                                  geodesic ? &states : nullptr, 64, check_target, delta_,
                                  lambda_);  // setDelta / setLambda of the base class stay the source of truth
    });
    freeState(scratch);
    if (!done) {  // the GPU call failed (KinematicChainConstraint::lastError()): "not reached", no states
      if (geodesic) geodesic->clear();
      return false;
    }
    if (geodesic) {
      geodesic->clear();
      for (const auto &st : states) {
        ompl::base::State *s = allocState();
        auto &x = *s->as<StateType>();
        for (int i = 0; i < 14; ++i) x[i] = st[i];
        geodesic->push_back(s);
      }
    }
    return ok;
  }

  std::shared_ptr<KinematicChainConstraint> chain_;
  uint64_t seed_;
  mutable std::atomic<uint64_t> samplers_{0};
};

// jy_MotionValidator (include/closed_chain_motion_planner/base/jy_ProjectedStateSpace.h:57-69): the same result —
// isSatisfied(s2) && discreteGeodesic(s1, s2, false) — from one launch (jy_ProjectedStateSpace::checkMotion above); a
// space of another type falls back to the reference's two calls.
class jy_MotionValidator : public ompl::base::ConstrainedMotionValidator {
public:
  jy_MotionValidator(const ompl::base::SpaceInformationPtr &si) : ompl::base::ConstrainedMotionValidator(si) {}
  bool checkMotion(const ompl::base::State *s1, const ompl::base::State *s2) const override
  {
    if (const auto *space = dynamic_cast<const jy_ProjectedStateSpace *>(&ss_)) return space->checkMotion(s1, s2);
    return ss_.getConstraint()->isSatisfied(s2) && ss_.discreteGeodesic(s1, s2, false);
  }
};

inline jy_ProjectedStateSampler::jy_ProjectedStateSampler(const jy_ProjectedStateSpace *space, ompl::base::StateSamplerPtr sampler, uint64_t seed)
  : ompl::base::WrapperStateSampler(space, std::move(sampler)),
    constraint_(std::dynamic_pointer_cast<KinematicChainConstraint>(space->getConstraint())),
    buffer_(constraint_->impl(), seed),
    near_(constraint_->impl(), seed, ccmp::RefSampleBuffer::Near),
    gauss_(constraint_->impl(), seed, ccmp::RefSampleBuffer::Gaussian)
{
}
// the reference's two-argument constructor (jy_ProjectedStateSpace.h:21): its own stream, like every other sampler
inline jy_ProjectedStateSampler::jy_ProjectedStateSampler(const jy_ProjectedStateSpace *space, ompl::base::StateSamplerPtr sampler)
  : jy_ProjectedStateSampler(space, std::move(sampler), space->nextSamplerSeed())
{
}

// A StateValidityChecker that asks the proxy scene first and the exact checker (the reference's
// KinematicChainValidityChecker, i.e. MoveIt) only for states the proxies do not already refuse:
//   si->setStateValidityChecker(std::make_shared<PrefilteredValidityChecker>(si, scene, valid_checker_));
// `reject_below` = 0 with inscribed proxies; the answer for every state MoveIt is asked about is MoveIt's.
class PrefilteredValidityChecker : public ompl::base::StateValidityChecker {
public:
  PrefilteredValidityChecker(const ompl::base::SpaceInformationPtr &si, std::shared_ptr<ccmp::ProxyScene> scene,
                             ompl::base::StateValidityCheckerPtr exact, double reject_below = 0.0)
    : ompl::base::StateValidityChecker(si), scene_(std::move(scene)), exact_(std::move(exact)), reject_below_(reject_below)
  {
  }
  bool isValid(const ompl::base::State *state) const override
  {
    const auto &q = *state->as<ompl::base::ConstrainedStateSpace::StateType>();  // an Eigen::Map over the 14 joints, as in
    const double clr = scene_->clearance(&q[0]);                                 // KinematicChain.cpp:94-99
    if (!(clr > reject_below_)) { rejected_++; return false; }
    asked_++;
    return exact_ ? exact_->isValid(state) : true;
  }
  uint64_t rejectedByProxies() const { return rejected_.load(); }
  uint64_t askedExact() const { return asked_.load(); }

private:
  std::shared_ptr<ccmp::ProxyScene> scene_;
  ompl::base::StateValidityCheckerPtr exact_;
  double reject_below_;
  mutable std::atomic<uint64_t> rejected_{0}, asked_{0};
};
#endif  // CCMP_WITH_OMPL

#endif  // CCMP_OMPL_ADAPTER_HPP
