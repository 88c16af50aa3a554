"""Batched discreteGeodesic throughput: E edges between projected (valid) states, as growTree produces them."""
import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402
ctx = Context(0)
c = KinematicChainConstraint.from_yaml("tests/golden/config/Wine_Bottle.yaml", ctx=ctx)
q, ok, _, _ = c.sample_project_batch(0x6E0, 0, 400000, want_iters=False)
good = q[ok == 1]
for E in (5, 1024, 16384):
    frm, to = good[:E].contiguous(), good[E:2 * E].contiguous()
    # targets 1 rad away along the straight line (a handful of delta = 0.25 steps each)
    d = to - frm
    to = (frm + d / d.norm(dim=1, keepdim=True)).contiguous()
    ms = timed(lambda: c.discrete_geodesic_batch(frm, to, 16), reps=3)
    st, n, okg, its = c.discrete_geodesic_batch(frm, to, 16)
    print("E=%-6d %8.3f ms  %.3e edges/s  mean states %.2f  reached %.3f  Newton iterations per edge %.1f"
          % (E, ms, E / ms * 1e3, n.float().mean().item(), okg.float().mean().item(), its.float().mean().item()), flush=True)
