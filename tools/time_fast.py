import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402
ctx = Context(0)
c = KinematicChainConstraint.from_yaml("tests/golden/config/Wine_Bottle.yaml", ctx=ctx)
c.setJacobianMode(1)
for B in (4096, 262144, 1048576):
    q = c.ambient_uniform_batch(0xC3, 0, B)
    out = torch.empty_like(q)
    res = []
    for wpc in (2, 4, 8):
        ctx.set_waves_per_cu(wpc)
        ms = timed(lambda: c.project_batch(q, out=out), reps=5)
        res.append("wpc%d %7.3f ms (%.3e/s)" % (wpc, ms, B / ms * 1e3))
    print("analytic B=%-8d " % B + "  ".join(res), flush=True)
