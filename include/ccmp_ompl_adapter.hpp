// ccmp_ompl_adapter.hpp — header-only C++ host side above the C ABI (include/ccmp.h).
//
// Part 1 (always available, no third-party headers): ccmp::Projector, an RAII owner of a
// ccmp_ctx + ccmp_problem with the reference's method names on raw double[14] buffers, and
// ccmp::SampleBuffer, a refill-on-empty batch of GPU-projected uniform samples.
//
// Part 2 (compiled only with -DCCMP_WITH_OMPL, i.e. inside the reference's catkin workspace where
// OMPL and Eigen exist): drop-in replacements that keep the reference's class names and virtual
// signatures, so src/planner/stefanBiPRM.cpp, src/base/constraints/ConstrainedPlanningCommon.cpp
// and src/main.cpp compile unchanged against them:
//   class KinematicChainConstraint : public ompl::base::Constraint
//       (replaces include/closed_chain_motion_planner/base/constraints/ConstraintFunction.h:21-137)
//   class jy_ProjectedStateSampler : public ompl::base::WrapperStateSampler
//       (replaces src/base/jy_ProjectedStateSpace.cpp:5-29)
//   class jy_ProjectedStateSpace : public ompl::base::ConstrainedStateSpace
//       (replaces src/base/jy_ProjectedStateSpace.cpp:32-96)
// Neither OMPL nor Eigen is installed in the build image of this repository, so part 2 is exercised
// only by inspection; part 1 is compiled and run by tests/test_cpp_adapter.py.
#ifndef CCMP_OMPL_ADAPTER_HPP
#define CCMP_OMPL_ADAPTER_HPP

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
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

// Owner of one execution context and one problem description.  All const methods are re-entrant
// with respect to the problem (it is passed by value into every launch); calls on one Projector
// from several threads must be serialised by the caller, exactly as the reference's graphMutex_
// serialises them today (src/planner/stefanBiPRM.cpp:280,383,449).
class Projector {
public:
  explicit Projector(int device = 0) { check(ccmp_ctx_create(device, &ctx_), "ccmp_ctx_create"); std::memset(&problem_, 0, sizeof problem_); }
  Projector(const std::string &yaml_path, int device = 0) : Projector(device) { loadConfig(yaml_path); }
  ~Projector() { ccmp_ctx_destroy(ctx_); }
  Projector(const Projector &) = delete;
  Projector &operator=(const Projector &) = delete;

  // grasping_point::loadConfig + ConstrainedProblem set-up (src/kinematics/grasping_point.cpp:34-65,
  // src/base/constraints/ConstrainedPlanningCommon.cpp:85-132)
  void loadConfig(const std::string &yaml_path) { check(ccmp_problem_from_yaml(yaml_path.c_str(), &problem_), "ccmp_problem_from_yaml"); }
  void setArmModels(const std::string &name1, int index1, const std::string &name2, int index2)
  {
    double q0[14];
    std::memcpy(q0, problem_.start_joint, sizeof q0);
    const double t1 = problem_.tol_pos, t2 = problem_.tol_rot;
    check(ccmp_problem_init(&problem_, name1.c_str(), index1, name2.c_str(), index2, q0, nullptr, nullptr, nullptr, nullptr), "ccmp_problem_init");
    if (t1 > 0 && t2 > 0) { problem_.tol_pos = t1; problem_.tol_rot = t2; }
  }
  void setInitialPosition(const double *init_joint14) { check(ccmp_set_start(&problem_, init_joint14), "ccmp_set_start"); }
  // throws where the reference throws ompl::Exception (ConstraintFunction.h:106-108)
  void setTolerance(double tolerance1, double tolerance2) { check(ccmp_set_tolerance(&problem_, tolerance1, tolerance2), "setTolerance: tolerance must be positive"); }
  void setJacobianMode(int mode) { problem_.jacobian_mode = mode; }

  // bool KinematicChainConstraint::project(Eigen::Ref<VectorXd> x) const — in place
  bool project(double *x14) const
  {
    uint8_t ok = 0;
    check(ccmp_project_host(ctx_, &problem_, x14, x14, &ok, nullptr, 1), "ccmp_project_host");
    return ok != 0;
  }
  void function(const double *x14, double *out2) const { check(ccmp_function_host(ctx_, &problem_, x14, out2, 1), "ccmp_function_host"); }
  bool isSatisfied(const double *x14) const
  {
    uint8_t ok = 0;
    check(ccmp_is_satisfied_host(ctx_, &problem_, x14, &ok, 1), "ccmp_is_satisfied_host");
    return ok != 0;
  }
  bool jointValid(const double *x14) const
  {
    uint8_t ok = 0;
    check(ccmp_joint_valid_host(ctx_, &problem_, x14, &ok, 1), "ccmp_joint_valid_host");
    return ok != 0;
  }
  // batches on host buffers (q row-major [B][14])
  void projectBatch(const double *q_in, double *q_out, uint8_t *ok, uint16_t *iters, size_t B) const
  {
    check(ccmp_project_host(ctx_, &problem_, q_in, q_out, ok, iters, B), "ccmp_project_host");
  }
  unsigned getCoDimension() const { return 2; }
  unsigned getAmbientDimension() const { return 14; }
  const ccmp_problem &problem() const { return problem_; }
  ccmp_problem &problem() { return problem_; }
  ccmp_ctx *ctx() const { return ctx_; }

private:
  ccmp_ctx *ctx_ = nullptr;
  ccmp_problem problem_;
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
    if (pos_ >= buf_.size() / 14) refill();
    std::memcpy(state14, &buf_[14 * pos_], 14 * sizeof(double));
    return ok_[pos_++] != 0;
  }

private:
  void refill()
  {
    buf_.resize(batch_ * 14);
    ok_.resize(batch_);
    check(ccmp_sample_project_host(proj_.ctx(), &proj_.problem(), seed_, next_index_, buf_.data(), ok_.data(), nullptr, batch_),
          "ccmp_sample_project_host");
    next_index_ += batch_;
    pos_ = 0;
  }
  const Projector &proj_;
  uint64_t seed_, next_index_ = 0;
  size_t batch_, pos_ = 0;
  std::vector<double> buf_;
  std::vector<uint8_t> ok_;
};

// jy_ProjectedStateSpace::discreteGeodesic (src/base/jy_ProjectedStateSpace.cpp:32-96) for one edge on raw
// buffers.  `valid` (nullable) is the host StateValidityChecker; it is consulted only when !interpolate,
// in order, and the list is cut at the first rejected state — where the reference's loop breaks.
template <class ValidFn>
inline bool discreteGeodesic(const Projector &proj, const double *from14, const double *to14, bool interpolate, ValidFn valid,
                             std::vector<std::vector<double>> *geodesic, int max_states = 256)
{
  std::vector<double> states((size_t)max_states * 14);
  int32_t n = 0;
  uint8_t ok = 0;
  check(ccmp_geodesic_host(proj.ctx(), &proj.problem(), from14, to14, 1, max_states, states.data(), &n, &ok), "ccmp_geodesic_host");
  bool good = ok != 0;
  int keep = n;
  if (!interpolate) {
    for (int k = 1; k < n; ++k)
      if (!valid(&states[(size_t)k * 14])) {
        keep = k;
        double d = 0;
        for (int i = 0; i < 14; ++i) { const double df = states[(size_t)(k - 1) * 14 + i] - to14[i]; d += df * df; }
        good = std::sqrt(d) <= proj.problem().delta;
        break;
      }
  }
  if (geodesic) {
    geodesic->clear();
    for (int k = 0; k < keep; ++k) geodesic->emplace_back(&states[(size_t)k * 14], &states[(size_t)k * 14] + 14);
  }
  return good;
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

  void setArmModels(const ArmModelPtr &arm1, const ArmModelPtr &arm2) { impl_->setArmModels(arm1->name, arm1->index, arm2->name, arm2->index); }
  void setInitialPosition(const Eigen::Ref<const Eigen::VectorXd> init_joint)
  {
    Eigen::VectorXd q = init_joint;  // contiguous copy
    impl_->setInitialPosition(q.data());
  }
  void setTolerance(const double tolerance1, const double tolerance2)
  {
    try { impl_->setTolerance(tolerance1, tolerance2); }
    catch (const ccmp::Error &) {
      throw ompl::Exception("ompl::base::Constraint::setProjectionTolerance(): tolerance must be positive.");
    }
  }
  bool project(Eigen::Ref<Eigen::VectorXd> x) const override
  {
    double buf[14];
    for (int i = 0; i < 14; ++i) buf[i] = x[i];
    const bool ok = impl_->project(buf);
    for (int i = 0; i < 14; ++i) x[i] = buf[i];
    return ok;
  }
  void function(const Eigen::Ref<const Eigen::VectorXd> &x, Eigen::Ref<Eigen::VectorXd> out) const override
  {
    double buf[14], f[2];
    for (int i = 0; i < 14; ++i) buf[i] = x[i];
    impl_->function(buf, f);
    out[0] = f[0];
    out[1] = f[1];
  }
  bool isSatisfied(const Eigen::Ref<const Eigen::VectorXd> &x) const override
  {
    double buf[14];
    for (int i = 0; i < 14; ++i) buf[i] = x[i];
    return impl_->isSatisfied(buf);
  }
  bool jointValid(const Eigen::Ref<const Eigen::VectorXd> &q) const
  {
    double buf[14];
    for (int i = 0; i < 14; ++i) buf[i] = q[i];
    return impl_->jointValid(buf);
  }
  ccmp::Projector &impl() const { return *impl_; }

private:
  std::shared_ptr<ccmp::Projector> impl_;
};
typedef std::shared_ptr<KinematicChainConstraint> ChainConstraintPtr;

// jy_ProjectedStateSampler (include/closed_chain_motion_planner/base/jy_ProjectedStateSpace.h:18-29): same
// name, same overrides; sampleUniform pops GPU-projected samples, Near/Gaussian keep the ambient draw of
// the wrapped sampler and project through the constraint (one-state launches).
class jy_ProjectedStateSpace;
typedef std::shared_ptr<jy_ProjectedStateSpace> jy_ProjectedStateSpacePtr;

class jy_ProjectedStateSampler : public ompl::base::WrapperStateSampler {
public:
  jy_ProjectedStateSampler(const jy_ProjectedStateSpace *space, ompl::base::StateSamplerPtr sampler, uint64_t seed = 0);
  void sampleUniform(ompl::base::State *state) override
  {
    auto &&x = *state->as<ompl::base::ConstrainedStateSpace::StateType>();
    double buf[14];
    buffer_.next(buf);  // already projected and wrapped by enforceBounds on the GPU
    for (int i = 0; i < 14; ++i) x[i] = buf[i];
  }
  void sampleUniformNear(ompl::base::State *state, const ompl::base::State *near, const double distance) override
  {
    ompl::base::WrapperStateSampler::sampleUniformNear(state, near, distance);
    constraint_->project(state);
    space_->enforceBounds(state);
  }
  void sampleGaussian(ompl::base::State *state, const ompl::base::State *mean, const double stdDev) override
  {
    ompl::base::WrapperStateSampler::sampleGaussian(state, mean, stdDev);
    constraint_->project(state);
    space_->enforceBounds(state);
  }

protected:
  const std::shared_ptr<KinematicChainConstraint> constraint_;
  ccmp::SampleBuffer buffer_;
};

// jy_ProjectedStateSpace (include/closed_chain_motion_planner/base/jy_ProjectedStateSpace.h:31-54): same name,
// same overrides.  discreteGeodesic runs the traversal on the GPU and applies the space information's
// StateValidityChecker on the host exactly where the reference consults it (src/base/jy_ProjectedStateSpace.cpp:
// 65-68); checkMotion of OMPL's ConstrainedMotionValidator (isSatisfied(s2) && discreteGeodesic(s1, s2),
// src/planner/stefanBiPRM.cpp:397-398,463-464) therefore needs no change.
class jy_ProjectedStateSpace : public ompl::base::ConstrainedStateSpace {
public:
  jy_ProjectedStateSpace(const ompl::base::StateSpacePtr &ambientSpace, const ompl::base::ConstraintPtr &constraint)
    : ompl::base::ConstrainedStateSpace(ambientSpace, constraint), chain_(std::dynamic_pointer_cast<KinematicChainConstraint>(constraint))
  {
    setName("Projected" + space_->getName());
  }
  ~jy_ProjectedStateSpace() override = default;
  ompl::base::StateSamplerPtr allocDefaultStateSampler() const override
  {
    return std::make_shared<jy_ProjectedStateSampler>(this, space_->allocDefaultStateSampler());
  }
  ompl::base::StateSamplerPtr allocStateSampler() const override
  {
    return std::make_shared<jy_ProjectedStateSampler>(this, space_->allocStateSampler());
  }
  bool discreteGeodesic(const ompl::base::State *from, const ompl::base::State *to, bool interpolate = false,
                        std::vector<ompl::base::State *> *geodesic = nullptr) const override
  {
    double a[14], b[14];
    const auto &fa = *from->as<StateType>();
    const auto &tb = *to->as<StateType>();
    for (int i = 0; i < 14; ++i) { a[i] = fa[i]; b[i] = tb[i]; }
    ccmp::Projector &proj = chain_->impl();
    proj.problem().delta = delta_;   // setDelta / setLambda of the base class stay the source of truth
    proj.problem().lambda = lambda_;
    auto &&svc = si_->getStateValidityChecker();
    std::vector<std::vector<double>> states;
    ompl::base::State *scratch = allocState();
    const bool ok = ccmp::discreteGeodesic(proj, a, b, interpolate,
                                           [&](const double *q) {
                                             auto &x = *scratch->as<StateType>();
                                             for (int i = 0; i < 14; ++i) x[i] = q[i];
                                             return svc->isValid(scratch);
                                           },
                                           geodesic ? &states : nullptr);
    freeState(scratch);
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

private:
  std::shared_ptr<KinematicChainConstraint> chain_;
};

// jy_MotionValidator (include/closed_chain_motion_planner/base/jy_ProjectedStateSpace.h:57-69): unchanged logic,
// isSatisfied(s2) is one single-state launch and the traversal runs in jy_ProjectedStateSpace::discreteGeodesic above.
class jy_MotionValidator : public ompl::base::ConstrainedMotionValidator {
public:
  jy_MotionValidator(const ompl::base::SpaceInformationPtr &si) : ompl::base::ConstrainedMotionValidator(si) {}
  bool checkMotion(const ompl::base::State *s1, const ompl::base::State *s2) const override
  {
    return ss_.getConstraint()->isSatisfied(s2) && ss_.discreteGeodesic(s1, s2, false);
  }
};

inline jy_ProjectedStateSampler::jy_ProjectedStateSampler(const jy_ProjectedStateSpace *space, ompl::base::StateSamplerPtr sampler, uint64_t seed)
  : ompl::base::WrapperStateSampler(space, std::move(sampler)),
    constraint_(std::dynamic_pointer_cast<KinematicChainConstraint>(space->getConstraint())),
    buffer_(constraint_->impl(), seed)
{
}
#endif  // CCMP_WITH_OMPL

#endif  // CCMP_OMPL_ADAPTER_HPP
