#!/usr/bin/env python3
"""Latency kernel alone (batches <= small_batch): index order against the FP32 scout's longest-predicted-first order.  Several
seeds per size (where the long samples fall in the index order is luck).  Interleaved; bit-identical by assertion.  GPU box."""
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
    for B in (3072, 4096, 6144, 8192, 10240):
        tot = {0: 0.0, 1: 0.0}
        rows = []
        for seed in (0xC2, 0xC3, 0x11, 0x22, 0x33, 0x44):
            q = c.ambient_uniform_batch(seed, 0, B)
            out = torch.empty_like(q)
            ctx.set_option("latency_order_min", 1 << 40)
            c.project_batch(q, out=out)
            torch.cuda.synchronize()
            ref = out.clone()
            t = {}
            for on in (0, 1, 0, 1):
                ctx.set_option("latency_order_min", 0 if on else 1 << 40)
                c.project_batch(q, out=out)
                torch.cuda.synchronize()
                assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), (obj, B, seed, on)
                t[on] = min(t.get(on, 1e9), mean(lambda: c.project_batch(q, out=out)))
            tot[0] += t[0]
            tot[1] += t[1]
            rows.append("%.3f|%.3f" % (t[0], t[1]))
        ctx.set_option("latency_order_min", 1 << 40)
        print("%-11s B=%6d index|scout ms per seed: %s   mean %.3f | %.3f (%+.1f %%)" % (obj, B, "  ".join(rows), tot[0] / 6, tot[1] / 6,
                                                                                        100.0 * (tot[1] / tot[0] - 1.0)), flush=True)
