// REPLACEMENT for the reference's include/closed_chain_motion_planner/base/jy_ProjectedStateSpace.h
// (copy this file over the original AND drop src/base/jy_ProjectedStateSpace.cpp from SOURCES in
// CMakeLists.txt:77 — the adapter defines the members that file defined, inline).
//
// What stays: the original header's includes (:3-13) and its `namespace ob = ompl::base;` (:14), which
// ConstrainedPlanningCommon.h:21 and src/planner/stefanBiPRM.cpp:4 rely on.  What goes: the forward declaration and typedef
// (:15-16) and the three class bodies (:18-69) — jy_ProjectedStateSpacePtr, jy_ProjectedStateSampler,
// jy_ProjectedStateSpace and jy_MotionValidator now come from the adapter: same names, same base classes, same virtual
// signatures; sampling and the extend step run on the GPU behind libccmp's C ABI (include/ccmp.h).
#pragma once

#include "ompl/base/MotionValidator.h"
#include "ompl/base/PlannerData.h"
#include "ompl/base/StateSampler.h"
#include "ompl/base/ValidStateSampler.h"
#include <ompl/base/Constraint.h>

#include <ompl/base/spaces/RealVectorStateSpace.h>
#include <ompl/base/spaces/constraint/ConstrainedStateSpace.h>

#include <Eigen/Core>
#include <utility>
namespace ob = ompl::base;

#ifndef CCMP_WITH_OMPL
#define CCMP_WITH_OMPL             // part 2 of the adapter: the classes with the reference's names
#endif
#include <ccmp_ompl_adapter.hpp>   // jy_ProjectedStateSpacePtr, jy_ProjectedStateSampler, jy_ProjectedStateSpace, jy_MotionValidator
