// Interface mock (see ../../README.md): the reference's jy_ProjectedStateSpace.h:4 includes this header and
// ConstrainedPlanningCommon.h:75 constructs a PlannerData from the space information; the adapter's part 2 uses nothing of it
#pragma once
#include <memory>
namespace ompl { namespace base {
class SpaceInformation;
class PlannerData {
public:
  explicit PlannerData(std::shared_ptr<SpaceInformation> si) : si_(std::move(si)) {}
  const std::shared_ptr<SpaceInformation> &getSpaceInformation() const { return si_; }
private:
  std::shared_ptr<SpaceInformation> si_;
};
} }
