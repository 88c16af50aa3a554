import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402
ctx = Context(0)
for obj in ("Wine_Bottle", "stefan"):
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    q = c.ambient_uniform_batch(0xC3, 0, 262144)
    out = torch.empty_like(q)
    res = []
    for wpc in (6, 8, 9, 10, 11, 12):
        ctx.set_waves_per_cu(wpc)
        ms = timed(lambda: c.project_batch(q, out=out), reps=4)
        res.append("wpc%-2d %7.3f" % (wpc, ms))
    print("%-12s " % obj + "  ".join(res), flush=True)
