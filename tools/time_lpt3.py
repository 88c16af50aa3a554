import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402
ctx = Context(0)
for obj in ("Wine_Bottle", "stefan"):
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (65536, 131072, 262144):
        q = c.ambient_uniform_batch(0xC3, 0, B)
        out = torch.empty_like(q)
        res = []
        for mode, thr in ((0, 10), (1, 10), (1, 5), (1, 3), (1, 1), (2, 10)):
            ctx.set_lpt(mode, 0)
            ctx.set_schedule(1, 0)
            ctx.set_option("handover_threshold", thr)
            ms = timed(lambda: c.project_batch(q, out=out), reps=4)
            res.append("lpt%d/thr%-2d %7.3f" % (mode, thr, ms))
        print("%-12s B=%-7d " % (obj, B) + "  ".join(res), flush=True)
