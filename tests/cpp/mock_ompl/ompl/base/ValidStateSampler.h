// Interface mock (see ../../README.md): the reference's jy_ProjectedStateSpace.h:6 includes this header; the adapter's
// part 2 uses nothing of it
#pragma once
#include <memory>
#include "ompl/base/State.h"
namespace ompl { namespace base {
class SpaceInformation;
class ValidStateSampler {
public:
  explicit ValidStateSampler(const SpaceInformation *si) : si_(si) {}
  virtual ~ValidStateSampler() = default;
  virtual bool sample(State *state) = 0;
  virtual bool sampleNear(State *state, const State *near, double distance) = 0;
protected:
  const SpaceInformation *si_;
};
typedef std::shared_ptr<ValidStateSampler> ValidStateSamplerPtr;
} }
