// Part 2 of include/ccmp_ompl_adapter.hpp (the classes with the reference's names) against the interface mock in
// tests/cpp/mock_ompl (NOT OMPL): type-checks the overrides and runs their control flow on the GPU.
// usage: adapter_ompl_check <start_joint 14 values...>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// exactly what INTEGRATION.md §2's replacement ConstraintFunction.h consists of: the original header's includes
// (include/closed_chain_motion_planner/base/constraints/ConstraintFunction.h:3-17), its `using namespace std;` (:20),
// then the adapter in place of the class body
#include <iostream>
#include <vector>
#include <string>
#include <fstream>
#include <memory>

#include <ompl/base/Constraint.h>
#include <ompl/base/ConstrainedSpaceInformation.h>
#include <ompl/base/spaces/constraint/ConstrainedStateSpace.h>
#include <ompl/base/spaces/constraint/ProjectedStateSpace.h>

#include <closed_chain_motion_planner/kinematics/panda_rbdl.h>

using namespace std;
#define CCMP_WITH_OMPL
#include "ccmp_ompl_adapter.hpp"

namespace ob = ompl::base;

static void print_hex(const char *tag, const double *v, int n)
{
  std::printf("%s", tag);
  for (int i = 0; i < n; i++) {
    uint64_t u;
    std::memcpy(&u, &v[i], 8);
    std::printf(" %016" PRIx64, u);
  }
  std::printf("\n");
}

// ambient space stand-in: KinematicChainSpace's enforceBounds (KinematicChain.h:118-130) and a fixed "sampler"
class AmbientSampler : public ob::StateSampler {
public:
  using ob::StateSampler::StateSampler;
  void sampleUniform(ob::State *s) override { fill(s, 0.1); }
  void sampleUniformNear(ob::State *s, const ob::State *near, double d) override
  {
    auto &x = *s->as<ob::ConstrainedStateSpace::StateType>();
    const auto &n = *near->as<ob::ConstrainedStateSpace::StateType>();
    for (int i = 0; i < 14; i++) x[i] = n[i] + ((i & 1) ? d : -d) * 0.5;
  }
  void sampleGaussian(ob::State *s, const ob::State *mean, double sd) override { sampleUniformNear(s, mean, sd); }
private:
  static void fill(ob::State *s, double v)
  {
    auto &x = *s->as<ob::ConstrainedStateSpace::StateType>();
    for (int i = 0; i < 14; i++) x[i] = v;
  }
};
class AmbientSpace : public ob::StateSpace {
public:
  AmbientSpace() { setName("KinematicChainSpace"); }
  ob::StateSamplerPtr allocDefaultStateSampler() const override { return std::make_shared<AmbientSampler>(this); }
  void enforceBounds(ob::State *s) const override
  {
    auto &x = *s->as<ob::ConstrainedStateSpace::StateType>();
    for (int i = 0; i < 14; i++) {
      double v = std::fmod(x[i], 2.0 * M_PI);
      if (v < -M_PI) v += 2.0 * M_PI;
      else if (v >= M_PI) v -= 2.0 * M_PI;
      x[i] = v;
    }
  }
  ob::State *allocState() const override { return new ob::ConstrainedStateSpace::StateType(); }
};
class CountingChecker : public ob::StateValidityChecker {
public:
  explicit CountingChecker(int accept) : accept_(accept) {}
  bool isValid(const ob::State *) const override { return calls_++ < accept_; }
  mutable int calls_ = 0;
  int accept_;
};

int main(int argc, char **argv)
{
  if (argc < 15) return 2;
  try {
    Eigen::VectorXd start(14);
    for (int i = 0; i < 14; i++) start[i] = std::atof(argv[1 + i]);
    // the set-up of ConstrainedProblem::setConstrainedOptions (ConstrainedPlanningCommon.cpp:116-132)
    auto arm1 = std::make_shared<ArmModel>();
    auto arm2 = std::make_shared<ArmModel>();
    arm1->name = "panda_left"; arm1->index = 0;
    arm2->name = "panda_right"; arm2->index = 1;
    // t_wb as ConstrainedProblem::_setEnvironment fills it from config->t_wb[index] (ConstrainedPlanningCommon.cpp:98;
    // frames of src/kinematics/grasping_point.cpp:11-13): the adapter takes the base frame from the ArmModel, by value
    arm1->t_wb.translation()(1) = 0.3;  arm1->t_wb.translation()(2) = 1.006;
    arm2->t_wb.translation()(1) = -0.3; arm2->t_wb.translation()(2) = 1.006;
    ChainConstraintPtr constraint = std::make_shared<KinematicChainConstraint>(14);
    constraint->setArmModels(arm1, arm2);
    constraint->setInitialPosition(start);
    constraint->setTolerance(1e-3, 5e-3);
    constraint->setMaxIterations(1000);
    bool threw = false;
    try { constraint->setTolerance(-1.0, 1.0); } catch (const ompl::Exception &) { threw = true; }
    std::printf("throws %d codim %u\n", threw ? 1 : 0, constraint->getCoDimension());

    auto ambient = std::make_shared<AmbientSpace>();
    auto space = std::make_shared<jy_ProjectedStateSpace>(ambient, constraint);
    auto si_ptr = std::make_shared<ob::SpaceInformation>();
    ob::SpaceInformation &si = *si_ptr;
    si.setStateSpace(space);
    space->setSpaceInformation(&si);
    space->setDelta(0.25);
    space->setLambda(2.0);
    std::printf("name %s\n", space->getName().c_str());

    // Constraint::project(State*) -> the Eigen::Ref override -> GPU
    ob::State *a = space->allocState(), *b = space->allocState();
    auto &xa = *a->as<ob::ConstrainedStateSpace::StateType>();
    auto &xb = *b->as<ob::ConstrainedStateSpace::StateType>();
    for (int i = 0; i < 14; i++) xa[i] = start[i] + 0.05 * ((i % 3) - 1);
    const bool oka = constraint->project(a);
    std::printf("project %d satisfied %d\n", oka ? 1 : 0, constraint->isSatisfied(a) ? 1 : 0);
    print_hex("xa", xa.values, 14);
    Eigen::VectorXd f(2);
    constraint->function(xa, f);
    print_hex("fa", f.data(), 2);

    // samplers: every sampler of the space has its own counter-based stream (seed = splitmix64(space seed + n));
    // sampleUniform / Near / Gaussian pop GPU-projected, wrapped samples
    ob::StateSamplerPtr sampler = space->allocDefaultStateSampler();
    ob::StateSamplerPtr sampler2 = space->allocStateSampler();
    sampler->sampleUniform(b);
    print_hex("uniform", xb.values, 14);
    sampler2->sampleUniform(b);
    print_hex("uniform2", xb.values, 14);
    sampler->sampleGaussian(b, a, 0.05);
    print_hex("gauss", xb.values, 14);
    sampler->sampleUniformNear(b, a, 0.2);
    print_hex("near0", xb.values, 14);
    sampler->sampleUniformNear(b, a, 0.2);  // same reference state: served from the look-ahead buffer
    std::printf("near_satisfied %d\n", constraint->isSatisfied(b) ? 1 : 0);
    print_hex("near", xb.values, 14);

    // extend step a -> b with a validity checker that accepts everything, then only the first two states
    for (int accept : {1000000, 2}) {
      auto svc = std::make_shared<CountingChecker>(accept);
      si.setStateValidityChecker(svc);
      std::vector<ob::State *> geo;
      const bool g = space->discreteGeodesic(a, b, false, &geo);
      std::printf("geodesic accept %d ok %d n %zu checker_calls %d\n", accept, g ? 1 : 0, geo.size(), svc->calls_);
      for (ob::State *s : geo) {
        print_hex("g", s->as<ob::ConstrainedStateSpace::StateType>()->values, 14);
        space->freeState(s);
      }
    }
    // growTree's neighbour loop through discreteGeodesics: edges a->b and b->b in one launch, checker accepting everything
    {
      auto svc = std::make_shared<CountingChecker>(1000000);
      si.setStateValidityChecker(svc);
      std::vector<std::vector<ob::State *>> lists;
      std::vector<char> reached;
      space->discreteGeodesics({a, b}, b, false, &lists, &reached);
      std::printf("geodesics ok %d %d n %zu %zu checker_calls %d\n", (int)reached[0], (int)reached[1], lists[0].size(), lists[1].size(), svc->calls_);
      print_hex("gl", lists[0].back()->as<ob::ConstrainedStateSpace::StateType>()->values, 14);
      for (auto &l : lists)
        for (ob::State *st : l) space->freeState(st);
    }
    si.setStateValidityChecker(std::make_shared<CountingChecker>(1000000));
    jy_MotionValidator mv(si_ptr);
    std::printf("checkMotion %d %d\n", mv.checkMotion(a, b) ? 1 : 0, mv.checkMotion(b, a) ? 1 : 0);
    const bool gi = space->discreteGeodesic(a, b, true);
    std::printf("geodesic_interpolate ok %d\n", gi ? 1 : 0);
    // proxy pre-filter in front of the exact checker: a sphere on either hand's fingertips, hands NOT allowed against
    // each other -> state a (hands a bottle's width apart) is refused by the proxies alone when they are fat, and handed
    // to the exact checker when they are thin
    for (double radius : {0.2, 0.01}) {
      std::vector<ccmp_sphere> sph(2);
      std::memset(sph.data(), 0, sph.size() * sizeof(ccmp_sphere));
      sph[0].frame = CCMP_FRAME(0, 7); sph[0].group = 0; sph[0].r = radius;
      sph[1].frame = CCMP_FRAME(1, 7); sph[1].group = 1; sph[1].r = radius;
      auto scene = std::make_shared<ccmp::ProxyScene>(constraint->impl(), sph, std::vector<ccmp_box>{});
      auto exact = std::make_shared<CountingChecker>(1000000);
      PrefilteredValidityChecker pre(si_ptr, scene, exact);
      const bool v = pre.isValid(a);
      std::printf("prefilter radius %.2f valid %d exact_calls %d rejected %llu\n", radius, v ? 1 : 0, exact->calls_,
                  (unsigned long long)pre.rejectedByProxies());
    }
    // the reference's configuration, in its order (ConstrainedProblem::setConstrainedOptions,
    // src/base/constraints/ConstrainedPlanningCommon.cpp:116-131): setArmModels -> setInitialPosition -> setTolerance ->
    // setMaxIterations(c_opt.tries = 1000) on the constraint, then setDelta / setLambda on the space — written the way an
    // includer of the original header writes it (unqualified shared_ptr / make_shared through `using namespace std;`)
    {
      struct { double delta = 0.25, lambda = 2.0, tolerance1 = 0.001, tolerance2 = 0.005; int tries = 1000; } c_opt;
      shared_ptr<KinematicChainConstraint> constraint2 = make_shared<KinematicChainConstraint>(14);
      auto css = make_shared<jy_ProjectedStateSpace>(ambient, constraint2);
      constraint2->setArmModels(arm1, arm2);
      constraint2->setInitialPosition(start);
      constraint2->setTolerance(c_opt.tolerance1, c_opt.tolerance2);
      constraint2->setMaxIterations(c_opt.tries);
      css->setDelta(c_opt.delta);
      css->setLambda(c_opt.lambda);
      // setMaxIterations reaches OMPL's base-class field only; project() keeps the cap of ConstraintFunction.h:26,68
      std::printf("replay cap %d delta %.2f lambda %.1f tol %.3g %.3g\n", (int)constraint2->impl().problem().max_iter, css->getDelta(),
                  css->getLambda(), constraint2->impl().problem().tol_pos, constraint2->impl().problem().tol_rot);
      ob::State *c2 = css->allocState();
      auto &xc = *c2->as<ob::ConstrainedStateSpace::StateType>();
      for (int i = 0; i < 14; i++) xc[i] = start[i] + 0.05 * ((i % 3) - 1);
      const bool okc = constraint2->project(c2);
      std::printf("replay project %d\n", okc ? 1 : 0);
      print_hex("xc", xc.values, 14);
      css->freeState(c2);
    }
    // jy_ProjectedStateSampler's two-argument constructor (jy_ProjectedStateSpace.h:21), as user code may call it directly:
    // the third sampler of `space`, with its own stream
    {
      jy_ProjectedStateSampler direct(space.get(), ambient->allocDefaultStateSampler());
      direct.sampleUniform(b);
      print_hex("uniform3", xb.values, 14);
    }
    // the same single-state calls through the context's resident service kernel (opt-in): the same bits, no launch on the call path
    {
      constraint->setResident(true);
      ob::State *r = space->allocState();
      auto &xr = *r->as<ob::ConstrainedStateSpace::StateType>();
      for (int i = 0; i < 14; i++) xr[i] = start[i] + 0.05 * ((i % 3) - 1);
      const bool okr = constraint->project(r);
      std::printf("resident project %d satisfied %d\n", okr ? 1 : 0, constraint->isSatisfied(r) ? 1 : 0);
      print_hex("xr", xr.values, 14);
      Eigen::VectorXd fr(2);
      constraint->function(xr, fr);
      print_hex("fr", fr.data(), 2);
      constraint->setResident(false);
      space->freeState(r);
    }
    space->freeState(a);
    space->freeState(b);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
