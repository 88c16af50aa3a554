// Interface mock (see ../../README.md)
#pragma once
#include <memory>
#include "ompl/base/State.h"
namespace ompl { namespace base {
class StateSpace;
class StateSampler {
public:
  explicit StateSampler(const StateSpace *space) : space_(space) {}
  virtual ~StateSampler() = default;
  virtual void sampleUniform(State *state) = 0;
  virtual void sampleUniformNear(State *state, const State *near, double distance) = 0;
  virtual void sampleGaussian(State *state, const State *mean, double stdDev) = 0;
protected:
  const StateSpace *space_;
};
typedef std::shared_ptr<StateSampler> StateSamplerPtr;
class WrapperStateSampler : public StateSampler {
public:
  WrapperStateSampler(const StateSpace *space, StateSamplerPtr sampler) : StateSampler(space), sampler_(std::move(sampler)) {}
  void sampleUniform(State *state) override;
  void sampleUniformNear(State *state, const State *near, double distance) override;
  void sampleGaussian(State *state, const State *mean, double stdDev) override;
protected:
  StateSamplerPtr sampler_;
};
} }
