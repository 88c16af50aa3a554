import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402
ctx = Context(0)
c = KinematicChainConstraint.from_yaml("tests/golden/config/Wine_Bottle.yaml", ctx=ctx)
B = 262144
q = c.ambient_uniform_batch(0xC3, 0, B)
out = torch.empty_like(q)
a = timed(lambda: c.project_batch(q, out=out), reps=4)
b = timed(lambda: c.sample_project_batch(0xC3, 0, B), reps=4)
print("project_batch %.3f ms   sample_project_batch (fused sampler + wrap) %.3f ms" % (a, b))
