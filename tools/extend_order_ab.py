#!/usr/bin/env python3
"""Extend step with a round budget (the callers' large-batch shape: lists of 16, 128 rounds per edge and call): which processing
order pays — index order, far-apart edges first, FP32 scout order at several round caps — now that no edge can hold the launch.
Interleaved on one device; results compared bit for bit.  Development aid (GPU box)."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402
from measure import near_edges, timed  # noqa: E402

ctx = Context(0)
for obj in sys.argv[1:] or ["Wine_Bottle", "stefan"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for E in (4096, 8192, 16384, 32768, 65536):
        frm, to = near_edges(c, E)
        call = lambda: c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)
        ref = call()
        torch.cuda.synchronize()
        row = []
        for rnd in range(2):
            for order, rounds in ((2, 64), (2, 32), (2, 16), (1, 0), (0, 0)):
                ctx.set_option("geodesic_order", order)
                ctx.set_option("geodesic_order_min", 0)
                ctx.set_option("geodesic_scout_min", 0)
                if rounds:
                    ctx.set_option("geodesic_scout_rounds", rounds)
                got = call()
                torch.cuda.synchronize()
                assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3]), (obj, E, order, rounds)
                row.append("%s %.3f" % ({2: "scout%d" % rounds, 1: "far-first", 0: "index"}[order], timed(call, 5)))
        ctx.set_option("geodesic_order", 2)
        ctx.set_option("geodesic_order_min", 4096)
        ctx.set_option("geodesic_scout_min", _lib.get_option(None, "geodesic_scout_min"))
        ctx.set_option("geodesic_scout_rounds", 64)
        print("%-11s E=%6d best ms  %s" % (obj, E, "  ".join(row)), flush=True)
