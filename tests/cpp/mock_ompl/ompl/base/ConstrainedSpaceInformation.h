// Interface mock (see ../../README.md): the reference's ConstraintFunction.h:13 includes this header; part 2 of the
// adapter uses nothing from it beyond what ConstrainedStateSpace.h declares.
#pragma once
#include "ompl/base/spaces/constraint/ConstrainedStateSpace.h"
