#!/usr/bin/env python3
"""Latency kernel alone: the scout ending early (once no more samples iterate than blocks are resident) against the full scout.
Six seeds per size; interleaved; bit-identical by assertion.  GPU box."""
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


for obj in sys.argv[1:] or ["Wine_Bottle", "stefan", "dumbbell"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (3072, 4096, 6144, 8192, 10240, 14336):
        tot = {0: 0.0, 1: 0.0}
        for seed in (0xC2, 0xC3, 0x11, 0x22, 0x33, 0x44):
            q = c.ambient_uniform_batch(seed, 0, B)
            out = torch.empty_like(q)
            ctx.set_option("latency_scout_early", 0)
            c.project_batch(q, out=out)
            torch.cuda.synchronize()
            ref = out.clone()
            t = {}
            for on in (0, 1, 0, 1):
                ctx.set_option("latency_scout_early", on)
                c.project_batch(q, out=out)
                torch.cuda.synchronize()
                assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), (obj, B, seed, on)
                t[on] = min(t.get(on, 1e9), mean(lambda: c.project_batch(q, out=out)))
            tot[0] += t[0] / 6
            tot[1] += t[1] / 6
        ctx.set_option("latency_scout_early", 0)
        print("%-11s B=%6d full scout %.3f | early end %.3f ms (%+.1f %%)" % (obj, B, tot[0], tot[1], 100.0 * (tot[1] / tot[0] - 1.0)), flush=True)
