#!/usr/bin/env python3
"""Hand-over threshold of the throughput kernel re-swept with the split launch on (the longest samples no longer wait for the
hand-over, so the rest may stay on the throughput kernel longer).  Interleaved; bit-identical by assertion.  GPU box."""
import sys

import torch

sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402

ctx = Context(0)


def mean(fn, reps=10):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for obj in sys.argv[1:] or ["Wine_Bottle", "stefan"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (12288, 16384, 20480, 24576, 32768, 40960, 49152, 57344, 65536):
        q = c.ambient_uniform_batch(0xC3, 0, B)
        out = torch.empty_like(q)
        c.project_batch(q, out=out)
        torch.cuda.synchronize()
        ref = out.clone()
        row = []
        for rnd in range(2):
            for thr, longrem in ((-1, 24), (60, 24), (70, 24), (80, 24), (100, 24), (10, 24), (-1, 12), (-1, 40)):
                ctx.set_option("handover_threshold", thr)
                ctx.set_option("pool_long_remaining", longrem)
                c.project_batch(q, out=out)
                torch.cuda.synchronize()
                assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), (obj, B, thr)
                row.append("t%d/l%d %.3f" % (thr, longrem, mean(lambda: c.project_batch(q, out=out))))
        ctx.set_option("handover_threshold", -1)
        ctx.set_option("pool_long_remaining", 24)
        print("%-11s B=%6d ms  %s" % (obj, B, "  ".join(row)), flush=True)
