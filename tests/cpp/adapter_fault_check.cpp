// Exception safety of Part 2 of include/ccmp_ompl_adapter.hpp: the reference's overrides return bool / void and
// are called from the planner's solution-checker thread too (src/planner/stefanBiPRM.cpp:848-849, ConstraintFunction.h:57,114);
// a HIP error inside them must not escape.  Links lib/libccmp_debug.so: ccmp_debug_fail_calls (include/ccmp_debug.h) makes every
// compute entry point of the constraint's context return CCMP_EHIP, from a SECOND thread, while that thread calls everything the
// planner calls; then the fault is lifted and the same calls must work again.  Built against tests/cpp/mock_ompl (NOT OMPL).
// usage: adapter_fault_check <start_joint 14 values...>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <ompl/base/Constraint.h>
#include <ompl/base/ConstrainedSpaceInformation.h>
#include <ompl/base/spaces/constraint/ConstrainedStateSpace.h>
#include <ompl/base/spaces/constraint/ProjectedStateSpace.h>

#include <closed_chain_motion_planner/kinematics/panda_rbdl.h>

using namespace std;
#define CCMP_WITH_OMPL
#include "ccmp_debug.h"
#include "ccmp_ompl_adapter.hpp"

namespace ob = ompl::base;

class AmbientSampler : public ob::StateSampler {
public:
  using ob::StateSampler::StateSampler;
  void sampleUniform(ob::State *) override {}
  void sampleUniformNear(ob::State *, const ob::State *, double) override {}
  void sampleGaussian(ob::State *, const ob::State *, double) override {}
};
class AmbientSpace : public ob::StateSpace {
public:
  AmbientSpace() { setName("KinematicChainSpace"); }
  ob::StateSamplerPtr allocDefaultStateSampler() const override { return std::make_shared<AmbientSampler>(this); }
  void enforceBounds(ob::State *) const override {}
  ob::State *allocState() const override { return new ob::ConstrainedStateSpace::StateType(); }
};
class YesChecker : public ob::StateValidityChecker {
public:
  bool isValid(const ob::State *) const override { return true; }
};

int main(int argc, char **argv)
{
  if (argc < 15) return 2;
  Eigen::VectorXd start(14);
  for (int i = 0; i < 14; i++) start[i] = std::atof(argv[1 + i]);
  auto arm1 = std::make_shared<ArmModel>();
  auto arm2 = std::make_shared<ArmModel>();
  arm1->name = "panda_left"; arm1->index = 0;
  arm2->name = "panda_right"; arm2->index = 1;
  arm1->t_wb.translation()(1) = 0.3;  arm1->t_wb.translation()(2) = 1.006;
  arm2->t_wb.translation()(1) = -0.3; arm2->t_wb.translation()(2) = 1.006;
  ChainConstraintPtr constraint = std::make_shared<KinematicChainConstraint>(14);
  constraint->setArmModels(arm1, arm2);
  constraint->setInitialPosition(start);
  constraint->setTolerance(1e-3, 5e-3);
  auto ambient = std::make_shared<AmbientSpace>();
  auto space = std::make_shared<jy_ProjectedStateSpace>(ambient, constraint);
  auto si_ptr = std::make_shared<ob::SpaceInformation>();
  si_ptr->setStateSpace(space);
  si_ptr->setStateValidityChecker(std::make_shared<YesChecker>());
  space->setSpaceInformation(si_ptr.get());
  space->setDelta(0.25);
  space->setLambda(2.0);
  ob::StateSamplerPtr sampler = space->allocDefaultStateSampler();

  ob::State *a = space->allocState(), *b = space->allocState(), *c = space->allocState();
  auto &xa = *a->as<ob::ConstrainedStateSpace::StateType>();
  auto &xb = *b->as<ob::ConstrainedStateSpace::StateType>();
  auto &xc = *c->as<ob::ConstrainedStateSpace::StateType>();
  for (int i = 0; i < 14; i++) { xa[i] = start[i] + 0.05 * ((i % 3) - 1); xb[i] = start[i]; xc[i] = 7.0 + i; }
  double before[14];
  for (int i = 0; i < 14; i++) before[i] = xa[i];

  int escaped = 0, wrong = 0;
  // ---- the fault, injected and met on a second thread (the reference's checkForSolution thread calls the same virtuals) ----
  std::thread worker([&] {
    try {
      if (ccmp_debug_fail_calls(constraint->impl().ctx(), 1 << 20) != CCMP_OK) wrong++;
      if (constraint->project(a)) wrong++;                                    // "no", state as it was
      for (int i = 0; i < 14; i++) if (xa[i] != before[i]) wrong++;
      if (constraint->isSatisfied(b)) wrong++;
      Eigen::VectorXd xe(14), f(2);
      for (int i = 0; i < 14; i++) xe[i] = xb[i];
      if (constraint->jointValid(xe)) wrong++;
      constraint->function(xe, f);
      if (!(std::isnan(f[0]) && std::isnan(f[1]))) wrong++;                   // nothing looks satisfied
      sampler->sampleUniform(c);
      sampler->sampleUniformNear(c, b, 0.2);
      sampler->sampleGaussian(c, b, 0.05);
      for (int i = 0; i < 14; i++) if (xc[i] != 7.0 + i) wrong++;             // untouched
      std::vector<ob::State *> states;
      if (space->discreteGeodesic(b, a, false, &states) || !states.empty()) wrong++;
      if (space->checkMotion(b, a)) wrong++;
      std::vector<std::vector<ob::State *>> lists;
      std::vector<char> reached;
      space->discreteGeodesics({b, b}, a, true, &lists, &reached);
      if (lists.size() != 2 || !lists[0].empty() || reached.size() != 2 || reached[0] || reached[1]) wrong++;
    } catch (...) {
      escaped++;
    }
  });
  worker.join();
  std::printf("fault escaped %d wrong %d lastError %d message_names_the_call %d\n", escaped, wrong, constraint->lastError(),
              constraint->lastErrorMessage().find("ccmp_project_host") != std::string::npos ? 1 : 0);
  bool threw = false;  // setTolerance stays the one thrower (ConstraintFunction.h:104-108)
  try { constraint->setTolerance(-1.0, 1.0); } catch (const ompl::Exception &) { threw = true; }
  std::printf("setTolerance throws %d\n", threw ? 1 : 0);
  // ---- the fault lifted: the same objects work, the error stays readable until cleared ----------------------------------
  ccmp_debug_fail_calls(constraint->impl().ctx(), 0);
  const bool ok = constraint->project(a);
  sampler->sampleUniform(c);
  int moved = 0;
  for (int i = 0; i < 14; i++) moved += xc[i] != 7.0 + i;
  std::vector<ob::State *> states;
  const bool reached = space->discreteGeodesic(b, a, true, &states);
  std::printf("after project %d satisfied %d sampled %d geodesic %d states %zu sticky %d\n", ok ? 1 : 0, constraint->isSatisfied(a) ? 1 : 0,
              moved > 0 ? 1 : 0, reached ? 1 : 0, states.size(), constraint->lastError());
  constraint->clearError();
  std::printf("cleared %d\n", constraint->lastError());
  return 0;
}
