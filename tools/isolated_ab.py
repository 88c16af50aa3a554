#!/usr/bin/env python3
"""Isolated calls (synchronise after each, as a planner issues them) against back-to-back launches, split launch on / off."""
import statistics
import sys

import torch

sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402

ctx = Context(0)
for obj in sys.argv[1:] or ["Wine_Bottle", "stefan"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (4096, 8192, 14336, 16384, 20480, 32768, 65536):
        q = c.ambient_uniform_batch(0xC3, 0, B)
        out = torch.empty_like(q)
        row = []
        for split in (1, 0, 1, 0):
            ctx.set_option("fd_split", split)
            for _ in range(3):
                c.project_batch(q, out=out)
            torch.cuda.synchronize()
            ts = []
            for _ in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                c.project_batch(q, out=out)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                c.project_batch(q, out=out)
            e1.record()
            torch.cuda.synchronize()
            row.append("split=%d isolated %.3f (min %.3f) back-to-back %.3f" % (split, statistics.median(ts), min(ts), e0.elapsed_time(e1) / 10))
        ctx.set_option("fd_split", 1)
        print("%-11s B=%6d ms  %s" % (obj, B, " | ".join(row)), flush=True)
