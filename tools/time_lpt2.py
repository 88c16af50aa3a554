"""In-process comparison of the scheduling policies: index order vs FP32-scout longest-first (with / without hand-over)."""
import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402
ctx = Context(0)
for obj in ("Wine_Bottle", "stefan"):
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (65536, 262144, 1048576):
        q = c.ambient_uniform_batch(0xC3, 0, B)
        out = torch.empty_like(q)
        ctx.set_lpt(0, 0)
        c.project_batch(q, out=out)
        ref = out.clone()
        for mode, name in ((0, "index order"), (1, "scout+LPT+hand-over"), (2, "scout+LPT")):
            ctx.set_lpt(mode, 0)
            ms = timed(lambda: c.project_batch(q, out=out), reps=4)
            print("%-12s B=%-8d %-20s %9.3f ms  %.3e proj/s  identical=%s" % (obj, B, name, ms, B / ms * 1e3, torch.equal(out, ref)), flush=True)
