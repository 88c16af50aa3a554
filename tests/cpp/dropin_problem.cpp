// First translation unit of the drop-in link test (tests/test_cpp_adapter.py::test_dropin_*): the two replacement headers of
// include/reference_overlay/ in the order ConstrainedPlanningCommon.h:13,21 includes the originals, and a replay of what the
// reference does with the classes they declare — ConstrainedProblem's constructor (src/base/constraints/
// ConstrainedPlanningCommon.cpp:5-12), _setEnvironment (:85-112: ArmModels in std::map order, t_wb from config->t_wb[index])
// and setConstrainedOptions (:116-132).  Linked with dropin_planner.cpp, which includes the same headers: every class of the
// adapter is then defined in two translation units of one program (ODR / inline).  Compiled against tests/cpp/mock_ompl
// (an interface mock, NOT OMPL); on the GPU box the program runs and its output is compared with the oracle.
// usage: dropin_check <arm1 name> <arm1 index> <arm2 name> <arm2 index> <shift of arm2's base along x> <start_joint x 14>
#include <closed_chain_motion_planner/base/constraints/ConstraintFunction.h>
#include <closed_chain_motion_planner/base/jy_ProjectedStateSpace.h>

#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "dropin_shared.h"

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

// stand-in for KinematicChainSpace (include/closed_chain_motion_planner/kinematics/KinematicChain.h:69-171): the wrap of
// enforceBounds (:118-130) is all the drop-in classes ask of the ambient space here
class AmbientSampler : public ob::StateSampler {
public:
  using ob::StateSampler::StateSampler;
  void sampleUniform(ob::State *) override {}
  void sampleUniformNear(ob::State *, const ob::State *, double) override {}
  void sampleGaussian(ob::State *, const ob::State *, double) override {}
};
class KinematicChainSpace : public ob::RealVectorStateSpace {
public:
  explicit KinematicChainSpace(unsigned int links) : ob::RealVectorStateSpace(links) { setName("KinematicChainSpace"); }
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
class AcceptAll : public ob::StateValidityChecker {
public:
  bool isValid(const ob::State *) const override { calls_++; return true; }
  mutable int calls_ = 0;
};

// grasping_point (src/kinematics/grasping_point.cpp:5-20): the three base frames, indexed left / right / top
struct grasping_point {
  std::vector<Eigen::Isometry3d> t_wb;
  Eigen::VectorXd start;
  std::string arm_name1, arm_name2;
  int arm_index1 = 0, arm_index2 = 0;
  grasping_point() : start(14)
  {
    Eigen::Isometry3d base_left, base_right, base_top;
    base_left.setIdentity(); base_right.setIdentity(); base_top.setIdentity();
    base_left.translation()(0) = 0;    base_left.translation()(1) = 0.3;   base_left.translation()(2) = 1.006;
    base_right.translation()(0) = 0;   base_right.translation()(1) = -0.3; base_right.translation()(2) = 1.006;
    base_top.translation()(0) = 1.35;  base_top.translation()(1) = 0.3;    base_top.translation()(2) = 1.006;
    base_top.linear()(0, 0) = -1; base_top.linear()(1, 1) = -1;
    t_wb.push_back(base_left);
    t_wb.push_back(base_right);
    t_wb.push_back(base_top);
  }
};
typedef std::shared_ptr<grasping_point> ConfigPtr;

// the members of ConstrainedProblem that touch the drop-in classes, in the reference's order and spelling
class ConstrainedProblem {
public:
  ConstrainedProblem(ob::StateSpacePtr space_, ChainConstraintPtr constraint_, ConfigPtr config_)
    : space(std::move(space_)), constraint(std::move(constraint_)), config(std::move(config_))
  {
    css = std::make_shared<jy_ProjectedStateSpace>(space, constraint);   // ConstrainedPlanningCommon.cpp:8
    csi = std::make_shared<ob::ConstrainedSpaceInformation>(css);        // :9
    css->setup();                                                        // :10
    arm_name_map_[config->arm_name1] = config->arm_index1;               // :13-14
    arm_name_map_[config->arm_name2] = config->arm_index2;
    _setEnvironment(arm_name_map_);
    setConstrainedOptions();
  }
  void _setEnvironment(const std::map<std::string, int> &arm_name_map)   // :85-112 (the ArmModel fields the projector reads)
  {
    for (auto it = arm_name_map.begin(); it != arm_name_map.end(); ++it) {
      arm_names_.push_back(it->first);
      arm_models_[it->first] = std::make_shared<ArmModel>();
      arm_models_[it->first]->name = it->first;
      arm_models_[it->first]->index = it->second;
      arm_models_[it->first]->t_wb = config->t_wb[it->second];
    }
  }
  void setConstrainedOptions()                                           // :116-132
  {
    c_opt.delta = 0.25; c_opt.lambda = 2.0; c_opt.tolerance1 = 0.001; c_opt.tolerance2 = 0.005; c_opt.tries = 1000;
    constraint->setArmModels(arm_models_[arm_names_[0]], arm_models_[arm_names_[1]]);
    constraint->setInitialPosition(config->start);
    constraint->setTolerance(c_opt.tolerance1, c_opt.tolerance2);
    constraint->setMaxIterations(c_opt.tries);
    css->setDelta(c_opt.delta);
    css->setLambda(c_opt.lambda);
  }
  struct { double delta, lambda, tolerance1, tolerance2; int tries; } c_opt;
  ob::StateSpacePtr space;
  ChainConstraintPtr constraint;
  ConfigPtr config;
  ob::ConstrainedStateSpacePtr css;            // ConstrainedPlanningCommon.h:168: held through the OMPL base class
  ob::ConstrainedSpaceInformationPtr csi;      // :170
  std::map<std::string, int> arm_name_map_;
  std::vector<std::string> arm_names_;
  std::map<std::string, ArmModelPtr> arm_models_;
};

int main(int argc, char **argv)
{
  if (argc < 20) return 2;
  try {
    int links = 14;                                                      // src/main.cpp:37-42
    auto ss = std::make_shared<KinematicChainSpace>(links);
    auto constraint = std::make_shared<KinematicChainConstraint>(links);
    ConfigPtr config = std::make_shared<grasping_point>();
    config->arm_name1 = argv[1]; config->arm_index1 = std::atoi(argv[2]);
    config->arm_name2 = argv[3]; config->arm_index2 = std::atoi(argv[4]);
    config->t_wb[config->arm_index2].translation()(0) += std::atof(argv[5]);  // an edit of grasping_point.cpp reaches the GPU by value
    for (int i = 0; i < 14; i++) config->start[i] = std::atof(argv[6 + i]);
    ConstrainedProblem cp(ss, constraint, config);
    const ccmp_problem &P = constraint->impl().problem();
    std::printf("arms %d %d name %s cap %d delta %.2f lambda %.1f\n", (int)P.arm_index[0], (int)P.arm_index[1], cp.css->getName().c_str(),
                (int)P.max_iter, cp.css->getDelta(), cp.css->getLambda());
    print_hex("base_p", &P.base_p[0][0], 6);
    print_hex("init_p", P.init_p, 3);

    auto svc = std::make_shared<AcceptAll>();
    cp.csi->setStateValidityChecker(svc);
    ob::SpaceInformationPtr si = cp.csi;
    ob::State *a = cp.css->allocState(), *b = cp.css->allocState(), *c = cp.css->allocState();
    auto &xa = *a->as<ob::ConstrainedStateSpace::StateType>();
    auto &xb = *b->as<ob::ConstrainedStateSpace::StateType>();
    auto &xc = *c->as<ob::ConstrainedStateSpace::StateType>();
    for (int i = 0; i < 14; i++) {
      xa[i] = config->start[i] + 0.05 * ((i % 3) - 1);
      xb[i] = config->start[i] - 0.04 * ((i % 4) - 1.5);
      xc[i] = config->start[i] + 0.03 * ((i % 5) - 2);
    }
    const bool oka = constraint->project(a), okb = constraint->project(b), okc = constraint->project(c);  // Constraint::project(State*)
    std::printf("project %d %d %d\n", oka ? 1 : 0, okb ? 1 : 0, okc ? 1 : 0);
    print_hex("xa", xa.values, 14);
    print_hex("xb", xb.values, 14);
    print_hex("xc", xc.values, 14);
    Eigen::VectorXd f(2);
    constraint->function(xa, f);
    print_hex("fa", f.data(), 2);

    // the planner's translation unit: growTree's loop, call by call and as one launch; checkMotion
    vector<vector<ob::State *>> lists, lists2;
    const int n1 = dropin_grow_tree(si, {a, b}, c, &lists);
    const int calls1 = svc->calls_;
    const int n2 = dropin_grow_tree_batched(si, {a, b}, c, &lists2);
    std::printf("grow connected %d %d n %zu %zu | %zu %zu checker_calls %d %d\n", n1, n2, lists[0].size(), lists[1].size(), lists2[0].size(),
                lists2[1].size(), calls1, svc->calls_ - calls1);
    for (int e = 0; e < 2; e++) {
      bool same = lists[e].size() == lists2[e].size();
      for (size_t k = 0; same && k < lists[e].size(); k++)
        same = std::memcmp(lists[e][k]->as<ob::ConstrainedStateSpace::StateType>()->values,
                           lists2[e][k]->as<ob::ConstrainedStateSpace::StateType>()->values, 14 * sizeof(double)) == 0;
      std::printf("edge %d same %d\n", e, same ? 1 : 0);
      for (ob::State *s : lists[e]) print_hex("g", s->as<ob::ConstrainedStateSpace::StateType>()->values, 14);
      for (ob::State *s : lists[e]) cp.css->freeState(s);
      for (ob::State *s : lists2[e]) cp.css->freeState(s);
    }
    std::printf("checkMotion %d\n", dropin_check_motion(si, a, c) ? 1 : 0);
    // a sampler of the space through the base-class pointer the reference holds (css is a ConstrainedStateSpacePtr)
    ob::StateSamplerPtr sampler = cp.css->allocDefaultStateSampler();
    sampler->sampleUniform(b);
    std::printf("sampled_satisfied %d\n", constraint->isSatisfied(b) ? 1 : 0);
    cp.css->freeState(a); cp.css->freeState(b); cp.css->freeState(c);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
