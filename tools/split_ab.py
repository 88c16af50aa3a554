#!/usr/bin/env python3
"""A/B of the split launch of mid-size reference-arithmetic batches (option "fd_split": the predicted-longest samples on latency
blocks beside the throughput kernel), interleaved on one device, results compared bit for bit.  Development aid (GPU box)."""
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


import os  # noqa: E402

# (on, predicted iterations, blocks, wavefronts per CU left out[, samples of the front: 0 = one per block]);
# SPLIT_CFGS="1,56,256,2,1024;1,40,512,2,0" and SPLIT_SIZES="16384,24576" override
CFGS = ((0, 0, 0, 2), (1, 56, 256, 2), (1, 56, 256, 1), (1, 56, 128, 1), (1, 40, 256, 2))
if os.environ.get("SPLIT_CFGS"):
    CFGS = ((0, 0, 0, 2),) + tuple(tuple(int(v) for v in c.split(",")) for c in os.environ["SPLIT_CFGS"].split(";"))
SIZES = (8192, 10240, 12288, 14336, 16384, 24576, 28672, 32768, 36864, 40960, 45056, 57344, 65536, 81920)
if os.environ.get("SPLIT_SIZES"):
    SIZES = tuple(int(v) for v in os.environ["SPLIT_SIZES"].split(","))
objs = sys.argv[1:] or ["Wine_Bottle", "stefan", "dumbbell"]
for obj in objs:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in SIZES:
        q = c.ambient_uniform_batch(0xC3, 0, B)
        out = torch.empty_like(q)
        small = B <= 10240  # the default policy's latency-kernel-alone range
        if small:  # below the default thresholds: scout + throughput kernel + hand-over forced, against the default policy
            ctx.set_option("fd_split", 0)
            c.project_batch(q, out=out)
            torch.cuda.synchronize()
            ref = out.clone()
            base = mean(lambda: c.project_batch(q, out=out))
            ctx.set_schedule(1, 0)
            ctx.set_lpt(1, 0)
        else:
            ctx.set_option("fd_split", 0)
            c.project_batch(q, out=out)
            torch.cuda.synchronize()
            ref = out.clone()
            base = None
        row = []
        for rnd in range(2):
            for on, pred, front, cut, *rest in CFGS:
                samples = rest[0] if rest else 0
                ctx.set_option("fd_split", on)
                ctx.set_option("fd_split_min", 0)
                if on:
                    ctx.set_option("fd_split_pred", pred)
                    ctx.set_option("fd_split_front", front)
                    ctx.set_option("fd_split_group_cut", cut)
                    ctx.set_option("fd_split_samples", samples)
                c.project_batch(q, out=out)
                torch.cuda.synchronize()
                assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), (obj, B, on, pred, front, cut)
                row.append("%s %.3f" % ("off" if not on else "p%d/f%d/c%d/s%d" % (pred, front, cut, samples), mean(lambda: c.project_batch(q, out=out))))
        ctx.set_option("fd_split", 1)
        ctx.set_option("fd_split_min", 0)
        for name in ("fd_split_pred", "fd_split_front", "fd_split_group_cut", "fd_split_samples"):
            ctx.set_option(name, -1)  # by the batch size again
        ctx.set_schedule(1)
        ctx.set_lpt(1)
        print("%-11s B=%6d ms %s %s" % (obj, B, ("default %.3f |" % base) if base else "", "  ".join(row)), flush=True)
