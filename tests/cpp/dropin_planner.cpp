// Second translation unit of the drop-in link test (tests/test_cpp_adapter.py::test_dropin_*): what a planner source does with
// the two replacement headers — src/planner/stefanBiPRM.cpp:4 includes jy_ProjectedStateSpace.h and growTree (:307-351) calls
// si_->getStateSpace()->as<jy_ProjectedStateSpace>()->discreteGeodesic(neighbour, new, false, &states) per neighbour.
// Header order differs from dropin_problem.cpp on purpose: either replacement header must stand on its own.  Compiled
// against tests/cpp/mock_ompl (an interface mock, NOT OMPL).
#include <closed_chain_motion_planner/base/jy_ProjectedStateSpace.h>
#include <closed_chain_motion_planner/base/constraints/ConstraintFunction.h>

#include "dropin_shared.h"

// growTree's neighbour loop, one call per neighbour as the reference writes it
int dropin_grow_tree(const ob::SpaceInformationPtr &si_, const vector<const ob::State *> &neighbors, const ob::State *new_state,
                     vector<vector<ob::State *>> *lists)
{
  int connected = 0;
  lists->clear();
  for (auto n : neighbors) {
    std::vector<ob::State *> states;
    if (si_->getStateSpace()->as<jy_ProjectedStateSpace>()->discreteGeodesic(n, new_state, false, &states)) connected++;
    lists->push_back(states);
  }
  return connected;
}

// the same loop as ONE launch (the adapter's extension)
int dropin_grow_tree_batched(const ob::SpaceInformationPtr &si_, const vector<const ob::State *> &neighbors, const ob::State *new_state,
                             vector<vector<ob::State *>> *lists)
{
  std::vector<char> reached;
  si_->getStateSpace()->as<jy_ProjectedStateSpace>()->discreteGeodesics(neighbors, new_state, false, lists, &reached);
  int connected = 0;
  for (char r : reached) connected += r ? 1 : 0;
  return connected;
}

// checkMotion through the validator class of the header (jy_ProjectedStateSpace.h:57-69)
bool dropin_check_motion(const ob::SpaceInformationPtr &si_, const ob::State *s1, const ob::State *s2)
{
  jy_MotionValidator mv(si_);
  return mv.checkMotion(s1, s2);
}
