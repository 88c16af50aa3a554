"""kernel time vs batch size at fixed occupancy: separates steady-state rate from the tail."""
import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402

ctx = Context(0)
c = KinematicChainConstraint.from_yaml("tests/golden/config/Wine_Bottle.yaml", ctx=ctx)
wpcs = [int(a) for a in sys.argv[1:]] or [12]
for wpc in wpcs:
    ctx.set_waves_per_cu(wpc)
    prev = None
    for B in (32768, 65536, 131072, 262144, 524288, 1048576):
        q = c.ambient_uniform_batch(0xC3, 0, B)
        out = torch.empty_like(q)
        ms = timed(lambda: c.project_batch(q, out=out), reps=2)
        extra = "" if prev is None else "  marginal %.3f ms per 262144" % ((ms - prev[1]) / (B - prev[0]) * 262144)
        print("waves/CU=%d B=%-8d %9.3f ms  %.3e proj/s%s" % (wpc, B, ms, B / ms * 1e3, extra), flush=True)
        prev = (B, ms)
