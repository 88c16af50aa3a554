"""Latency of the reference-signature single-state calls (batch of one through the *_host entry points)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
ctx = Context(0)
c = KinematicChainConstraint.from_yaml("tests/golden/config/Wine_Bottle.yaml", ctx=ctx)
import torch
q = c.ambient_uniform_batch(1, 0, 64).cpu().numpy()
for name, fn in (("project", lambda x: c.project(x)), ("function", lambda x: c.function(x)), ("isSatisfied", lambda x: c.isSatisfied(x))):
    ts = []
    for i in range(64):
        x = q[i].copy()
        t0 = time.perf_counter(); fn(x); ts.append(time.perf_counter() - t0)
    ts = np.array(ts[4:]) * 1e6
    print("%-12s median %.1f us  p90 %.1f us  min %.1f us" % (name, np.median(ts), np.percentile(ts, 90), ts.min()))
# near-manifold projections (the geodesic regime): few Newton iterations
x0 = np.array(c.problem.start_joint[:])
ts = []
for i in range(40):
    x = x0 + 0.02 * np.sin(np.arange(14) + i)
    t0 = time.perf_counter(); ok = c.project(x); ts.append(time.perf_counter() - t0)
print("project near the manifold: median %.1f us" % (np.median(np.array(ts[4:])) * 1e6))
