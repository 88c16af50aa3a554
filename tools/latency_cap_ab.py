#!/usr/bin/env python3
"""Latency kernel alone in the scout's order: how far must the scout predict?  Its run time is its longest lane's (cap x ~1 us);
the order only decides who gets into the first fill of the 2 048 blocks.  Caps 96 (the scout's own) ... 16, six seeds per size.
Interleaved; bit-identical by assertion.  GPU box."""
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


CAPS = (0, 64, 48, 40, 32, 24, 16)
for obj in sys.argv[1:] or ["Wine_Bottle", "stefan"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (3072, 4096, 6144, 8192, 10240, 14336):
        tot = {cap: 0.0 for cap in CAPS}
        for seed in (0xC2, 0xC3, 0x11, 0x22, 0x33, 0x44):
            q = c.ambient_uniform_batch(seed, 0, B)
            out = torch.empty_like(q)
            ctx.set_option("latency_scout_cap", 0)
            c.project_batch(q, out=out)
            torch.cuda.synchronize()
            ref = out.clone()
            for cap in CAPS:
                ctx.set_option("latency_scout_cap", cap)
                c.project_batch(q, out=out)
                torch.cuda.synchronize()
                assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), (obj, B, seed, cap)
                tot[cap] += mean(lambda: c.project_batch(q, out=out)) / 6
        ctx.set_option("latency_scout_cap", 0)
        print("%-11s B=%6d mean ms by cap: %s" % (obj, B, "  ".join("%s %.3f" % (cap or 96, tot[cap]) for cap in CAPS)), flush=True)
