// Interface mock (see ../../README.md): stands where the reference's panda_rbdl.h (PandaModel over RBDL) is included by
// ConstraintFunction.h:17; adapter part 2 reads only ArmModel's name and index.
#pragma once
#include <closed_chain_motion_planner/kinematics/panda_model.h>
