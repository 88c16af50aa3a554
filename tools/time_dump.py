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
    for thr in (10, 8, 6, 4, 2):
        ctx.set_schedule(1, 0)
        ctx.set_option("handover_threshold", thr)
        ms = timed(lambda: c.project_batch(q, out=out), reps=3)
        print("%-12s dump_threshold=%-2d %9.3f ms  %.3e proj/s" % (obj, thr, ms, 262144 / ms * 1e3), flush=True)
