// Interface mock (see ../../../../README.md): a 14-double state, a wrapped ambient space, the pieces of
// ConstrainedStateSpace / SpaceInformation / StateValidityChecker that adapter part 2 reaches.
#pragma once
#include <Eigen/Dense>
#include <cmath>
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include <stdexcept>
#include "ompl/base/Constraint.h"
#include "ompl/base/MotionValidator.h"
#include "ompl/base/StateSampler.h"
namespace ompl { namespace base {
class StateValidityChecker {
public:
  StateValidityChecker() = default;
  explicit StateValidityChecker(const SpaceInformationPtr &) {}
  virtual ~StateValidityChecker() = default;
  virtual bool isValid(const State *state) const = 0;
};
typedef std::shared_ptr<StateValidityChecker> StateValidityCheckerPtr;
class StateSpace;
class SpaceInformation {
public:
  SpaceInformation() = default;
  explicit SpaceInformation(std::shared_ptr<StateSpace> space) : space_(std::move(space)) {}
  virtual ~SpaceInformation() = default;
  const StateValidityCheckerPtr &getStateValidityChecker() const { return svc_; }
  void setStateValidityChecker(const StateValidityCheckerPtr &s) { svc_ = s; }
  const std::shared_ptr<StateSpace> &getStateSpace() const { return space_; }
  void setStateSpace(const std::shared_ptr<StateSpace> &s) { space_ = s; }
private:
  StateValidityCheckerPtr svc_;
  std::shared_ptr<StateSpace> space_;
};
class StateSpace {
public:
  virtual ~StateSpace() = default;
  template <class T> const T *as() const { return static_cast<const T *>(this); }
  template <class T> T *as() { return static_cast<T *>(this); }
  virtual void setup() {}
  const std::string &getName() const { return name_; }
  void setName(const std::string &n) { name_ = n; }
  virtual StateSamplerPtr allocDefaultStateSampler() const = 0;
  virtual StateSamplerPtr allocStateSampler() const { return allocDefaultStateSampler(); }
  virtual void enforceBounds(State *state) const = 0;
  virtual State *allocState() const = 0;
  virtual void freeState(State *state) const { delete state; }
private:
  std::string name_;
};
typedef std::shared_ptr<StateSpace> StateSpacePtr;
class ConstrainedStateSpace : public StateSpace {
public:
  class StateType : public State, public Eigen::Map<Eigen::VectorXd> {
  public:
    StateType() : Eigen::Map<Eigen::VectorXd>(values, 14) { for (double &v : values) v = 0.0; }
    double values[14];
  };
  ConstrainedStateSpace(const StateSpacePtr &ambientSpace, const ConstraintPtr &constraint)
    : si_(nullptr), space_(ambientSpace), constraint_(constraint), delta_(0.05), lambda_(2.0) {}
  const ConstraintPtr getConstraint() const { return constraint_; }
  void setSpaceInformation(SpaceInformation *si) { si_ = si; }
  // as OMPL's: a ConstrainedStateSpace refuses to be set up before a SpaceInformation is associated with it
  void setup() override
  {
    if (si_ == nullptr) throw std::runtime_error("ConstrainedStateSpace::setup(): Must associate a SpaceInformation object before use.");
  }
  void setDelta(double d) { delta_ = d; }
  void setLambda(double l) { lambda_ = l; }
  double getDelta() const { return delta_; }
  double getLambda() const { return lambda_; }
  void enforceBounds(State *state) const override { space_->enforceBounds(state); }
  State *allocState() const override { return new StateType(); }
  virtual bool discreteGeodesic(const State *from, const State *to, bool interpolate = false, std::vector<State *> *geodesic = nullptr) const = 0;
protected:
  SpaceInformation *si_;
  const StateSpacePtr space_;
  const ConstraintPtr constraint_;
  double delta_, lambda_;
};
class ConstrainedMotionValidator : public MotionValidator {
public:
  explicit ConstrainedMotionValidator(SpaceInformation *si) : MotionValidator(si), ss_(*static_cast<const StateSpace *>(si->getStateSpace().get())->as<ConstrainedStateSpace>()) {}
  explicit ConstrainedMotionValidator(const SpaceInformationPtr &si) : ConstrainedMotionValidator(si.get()) {}
  bool checkMotion(const State *s1, const State *s2) const override { return ss_.getConstraint()->isSatisfied(s2) && ss_.discreteGeodesic(s1, s2, false); }
protected:
  const ConstrainedStateSpace &ss_;
};
typedef std::shared_ptr<ConstrainedStateSpace> ConstrainedStateSpacePtr;
// ompl/base/ConstrainedSpaceInformation.h: associates itself with the constrained space it is built on
class ConstrainedSpaceInformation : public SpaceInformation {
public:
  explicit ConstrainedSpaceInformation(StateSpacePtr space) : SpaceInformation(std::move(space))
  {
    getStateSpace()->as<ConstrainedStateSpace>()->setSpaceInformation(this);
  }
};
typedef std::shared_ptr<ConstrainedSpaceInformation> ConstrainedSpaceInformationPtr;
inline bool Constraint::project(State *state) const { return project(Eigen::Ref<Eigen::VectorXd>(*state->as<ConstrainedStateSpace::StateType>())); }
inline bool Constraint::isSatisfied(const State *state) const { return isSatisfied(Eigen::Ref<const Eigen::VectorXd>(*state->as<ConstrainedStateSpace::StateType>())); }
inline void WrapperStateSampler::sampleUniform(State *s) { sampler_->sampleUniform(s); }
inline void WrapperStateSampler::sampleUniformNear(State *s, const State *near, double d) { sampler_->sampleUniformNear(s, near, d); }
inline void WrapperStateSampler::sampleGaussian(State *s, const State *mean, double sd) { sampler_->sampleGaussian(s, mean, sd); }
} }
