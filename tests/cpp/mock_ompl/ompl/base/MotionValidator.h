// Interface mock (see ../../README.md): ompl::base::MotionValidator as jy_MotionValidator's base chain uses it
// (the reference's jy_ProjectedStateSpace.h:3 includes this header)
#pragma once
#include <memory>
#include "ompl/base/State.h"
namespace ompl { namespace base {
class SpaceInformation;
typedef std::shared_ptr<SpaceInformation> SpaceInformationPtr;
class MotionValidator {
public:
  explicit MotionValidator(SpaceInformation *si) : si_(si) {}
  explicit MotionValidator(const SpaceInformationPtr &si) : si_(si.get()) {}
  virtual ~MotionValidator() = default;
  virtual bool checkMotion(const State *s1, const State *s2) const = 0;
protected:
  SpaceInformation *si_;
};
typedef std::shared_ptr<MotionValidator> MotionValidatorPtr;
} }
