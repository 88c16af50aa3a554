#!/usr/bin/env python3
"""Where should the latency kernel alone (in the scout's order) hand over to scout + throughput kernel + split launch?
Both paths at 6 144 ... 32 768 samples, three objects, two seeds.  Interleaved; bit-identical by assertion.  GPU box."""
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


ctx.set_option("latency_order_min", 0)
ctx.set_option("fd_split_min", 0)
for obj in sys.argv[1:] or ["Wine_Bottle", "stefan", "dumbbell"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (6144, 8192, 10240, 12288, 14336, 16384, 20480, 24576, 32768):
        cols = {"lat": 0.0, "split": 0.0}
        for seed in (0xC3, 0x51):
            q = c.ambient_uniform_batch(seed, 0, B)
            out = torch.empty_like(q)
            ref = None
            for rnd in range(2):
                for name, small in (("lat", 1 << 30), ("split", 0)):
                    ctx.set_schedule(1, small)
                    c.project_batch(q, out=out)
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = out.clone()
                    assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), (obj, B, name)
                    cols[name] += mean(lambda: c.project_batch(q, out=out)) / 4
        ctx.set_schedule(1)
        print("%-11s B=%6d  ordered latency kernel alone %.3f | scout + throughput + split %.3f ms" % (obj, B, cols["lat"], cols["split"]), flush=True)
