"""default policy with the one-round 128-thread latency kernel (flat) vs the wave/pair kernels, interleaved"""
import statistics
import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402

ctx = Context(0)
for obj in ("Wine_Bottle", "stefan"):
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (4096, 16384, 32768, 65536, 131072, 262144):
        q = c.ambient_uniform_batch(0xC3, 0, B)
        out = torch.empty_like(q)
        t = {0: [], 1: []}
        for rep in range(5):
            for flat in (0, 1):
                ctx.set_option("flat_kernel", flat)
                t[flat].append(timed(lambda: c.project_batch(q, out=out), reps=1))
        m0, m1 = statistics.median(t[0]), statistics.median(t[1])
        print("%-12s B=%-7d wave/pair %8.3f ms   flat %8.3f ms   ratio %.3f   (%.2e/s)" % (obj, B, m0, m1, m1 / m0, B / m1 * 1e3), flush=True)
