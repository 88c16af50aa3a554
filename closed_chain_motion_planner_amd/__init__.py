"""closed_chain_motion_planner_amd — MI355X-native batched closed-chain constraint projector.

Only the projector hot path of jkw0701/closed_chain_motion_planner: the C-ABI library
`libccmp.so` (include/ccmp.h, csrc/) and this thin host mirror of the reference's constraint
interface.  Importing the package needs neither a GPU nor the built library; using it does.
"""
from ._lib import CCMP_JAC_ANALYTIC, CCMP_JAC_FD, CcmpError, CcmpProblem, describe, get_option, option_table  # noqa: F401
from .constraint import ArmModel, Communicator, Context, KinematicChainConstraint, load_config  # noqa: F401

from .space import (check_motion, format_graphml, format_graphviz, format_path_matrix, geodesic_interpolate,  # noqa: F401
                    jy_ProjectedStateSampler, jy_ProjectedStateSpace, next_sampler_seed, parse_graphml, parse_path_matrix,
                    splitmix64)

from .scene import ProxyScene, ProxyValidityChecker, default_allowed, skeleton_spheres  # noqa: F401

__version__ = "0.6.0"
