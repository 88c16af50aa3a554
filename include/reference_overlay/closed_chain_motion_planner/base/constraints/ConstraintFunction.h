// REPLACEMENT for the reference's include/closed_chain_motion_planner/base/constraints/ConstraintFunction.h
// (copy this file over the original).
//
// What stays, line for line, is everything the original header hands to its includers besides the class: its standard and
// OMPL includes (original :3-17) and the `using namespace std;` of :20 — jy_ConstrainedValidStateSampler.h:15 and
// ConstrainedPlanningCommon.h:13 include this header and the sources behind them use unqualified vector / string /
// shared_ptr / cout and the OMPL types through it.  What goes is the class body (original :21-140): class
// KinematicChainConstraint and the typedef ChainConstraintPtr now come from the adapter, same names, same signatures, the
// arithmetic on the GPU behind libccmp's C ABI (include/ccmp.h).
#pragma once

#include <iostream>
#include <vector>
#include <string>
#include <fstream>
#include <memory>

#include <ompl/base/Constraint.h>
#include <ompl/base/ConstrainedSpaceInformation.h>
#include <ompl/base/spaces/constraint/ConstrainedStateSpace.h>
#include <ompl/base/spaces/constraint/ProjectedStateSpace.h>

#include <closed_chain_motion_planner/kinematics/panda_rbdl.h>   // PandaModel: still used by ArmModel and elsewhere

using namespace std;               // as the original's :20 — its includers rely on it

#ifndef CCMP_WITH_OMPL
#define CCMP_WITH_OMPL             // part 2 of the adapter: the classes with the reference's names
#endif
#include <ccmp_ompl_adapter.hpp>   // class KinematicChainConstraint : public ompl::base::Constraint; ChainConstraintPtr
