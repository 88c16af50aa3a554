// Interface mock (see ../../../../README.md): included by the reference's ConstraintFunction.h:15; the reference derives
// its own jy_ProjectedStateSpace from ConstrainedStateSpace, nothing of OMPL's ProjectedStateSpace is used.
#pragma once
#include "ompl/base/spaces/constraint/ConstrainedStateSpace.h"
