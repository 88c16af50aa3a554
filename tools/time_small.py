import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402
ctx = Context(0)
c = KinematicChainConstraint.from_yaml("tests/golden/config/Wine_Bottle.yaml", ctx=ctx)
for B in (1, 64, 1024, 2048, 4096, 8192, 16384):
    q = c.ambient_uniform_batch(0xC2, 0, B)
    out = torch.empty_like(q)
    res = []
    for name, sched, small, flat in (("flat", 2, 0, 1), ("single-wave", 2, 0, 0), ("default", 1, -1, 1)):
        ctx.set_schedule(sched, small if small >= 0 else None)
        ctx.set_option("flat_kernel", flat)
        ms = timed(lambda: c.project_batch(q, out=out), reps=5)
        res.append("%s %7.3f ms (%.2e/s)" % (name, ms, B / ms * 1e3))
    print("B=%-5d " % B + "   ".join(res), flush=True)
