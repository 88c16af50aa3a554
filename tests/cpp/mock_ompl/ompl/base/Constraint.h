// Interface mock (see ../../README.md): ompl::base::Constraint as the reference uses it.
#pragma once
#include <Eigen/Dense>
#include <memory>
#include "ompl/base/State.h"
namespace ompl { namespace base {
class Constraint {
public:
  Constraint(unsigned int ambientDim, unsigned int coDim, double tolerance = 1e-4) : n_(ambientDim), k_(coDim), tolerance_(tolerance), maxIterations_(50) {}
  virtual ~Constraint() = default;
  virtual void function(const Eigen::Ref<const Eigen::VectorXd> &x, Eigen::Ref<Eigen::VectorXd> out) const = 0;
  virtual bool project(Eigen::Ref<Eigen::VectorXd> x) const { (void)x; return false; }
  virtual bool isSatisfied(const Eigen::Ref<const Eigen::VectorXd> &x) const { (void)x; return false; }
  bool project(State *state) const;          // maps the state's vector and calls the override (ConstrainedStateSpace.h below)
  bool isSatisfied(const State *state) const;
  unsigned int getAmbientDimension() const { return n_; }
  unsigned int getCoDimension() const { return k_; }
  void setMaxIterations(unsigned int it) { maxIterations_ = it; }
protected:
  unsigned int n_, k_;
  double tolerance_;
  unsigned int maxIterations_;
};
typedef std::shared_ptr<Constraint> ConstraintPtr;
} }
