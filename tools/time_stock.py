"""kernels specialised for the stock Panda structure vs the general kernels, interleaved, bitwise comparison"""
import statistics
import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402

ctx = Context(0)
for obj in ("Wine_Bottle", "stefan"):
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (1, 4096, 32768, 262144):
        q = c.ambient_uniform_batch(0xC3, 0, B)
        outs = {}
        t = {0: [], 1: []}
        for rep in range(5):
            for stock in (0, 1):
                ctx.set_option("stock_kernels", stock)
                out = torch.empty_like(q)
                t[stock].append(timed(lambda: c.project_batch(q, out=out), reps=1))
                outs[stock] = out
        m0, m1 = statistics.median(t[0]), statistics.median(t[1])
        print("%-12s B=%-7d general %8.3f ms   stock %8.3f ms   ratio %.3f   identical %s" % (obj, B, m0, m1, m1 / m0, torch.equal(outs[0], outs[1])), flush=True)
