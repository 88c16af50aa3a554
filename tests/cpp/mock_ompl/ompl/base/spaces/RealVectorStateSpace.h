// Interface mock (see ../../../README.md): the reference's jy_ProjectedStateSpace.h:9 includes this header
// (KinematicChainSpace derives from RealVectorStateSpace, KinematicChain.h:69); the adapter's part 2 uses nothing of it —
// the ambient space reaches it as a StateSpacePtr
#pragma once
#include "ompl/base/spaces/constraint/ConstrainedStateSpace.h"
namespace ompl { namespace base {
class RealVectorStateSpace : public StateSpace {
public:
  explicit RealVectorStateSpace(unsigned int dim = 0) : dimension_(dim) {}
  unsigned int getDimension() const { return dimension_; }
protected:
  unsigned int dimension_;
};
} }
