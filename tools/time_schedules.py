"""A/B of the projector schedules (group kernel only / + wave-per-sample stragglers / wave only)."""
import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402

obj = sys.argv[1] if len(sys.argv) > 1 else "Wine_Bottle"
ctx = Context(0)
c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
for B in (4096, 262144):
    q = c.ambient_uniform_batch(0xC3, 0, B)
    out = torch.empty_like(q)
    for sched, name in ((0, "group only"), (1, "group+wave"), (2, "wave only")):
        if sched == 2 and B > 300000:
            continue
        for wpc in (8, 12):
            ctx.set_waves_per_cu(wpc)
            ctx.set_schedule(sched, 0)
            ms = timed(lambda: c.project_batch(q, out=out), reps=3)
            print("%-12s B=%-8d %-11s waves/CU=%-2d %9.3f ms  %.3e proj/s" % (obj, B, name, wpc, ms, B / ms * 1e3), flush=True)
