#!/usr/bin/env python3
"""Extend step: head start (the first resident-capacity edges in index order on a side stream while the scout orders the batch)
against the scout alone.  Interleaved; compared bit for bit.  Development aid (GPU box)."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from measure import near_edges, timed  # noqa: E402

ctx = Context(0)
for obj in sys.argv[1:] or ["Wine_Bottle", "stefan"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for E in (4096, 6144, 8192, 16384, 32768, 65536):
        frm, to = near_edges(c, E)
        for budget, cap in ((128, 16), (0, 16), (0, 64)):
            call = lambda: c.discrete_geodesic_batch(frm, to, cap, want_carry=True, round_budget=budget)
            ctx.set_option("geodesic_head_start", 0)
            ctx.set_option("geodesic_flavour", 1)
            ref = call()
            torch.cuda.synchronize()
            row = []
            for rnd in range(2):
                for hs in (0, 1):
                    ctx.set_option("geodesic_head_start", hs)
                    got = call()
                    torch.cuda.synchronize()
                    live = torch.arange(cap, device=frm.device)[None, :] < ref[1].clamp(max=cap)[:, None]
                    assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3]), (obj, E, hs)
                    assert torch.equal(got[0][live].view(torch.int64), ref[0][live].view(torch.int64)) and torch.equal(got[4].view(torch.int64), ref[4].view(torch.int64))
                    row.append("%s %.3f" % ("head" if hs else "scout", timed(call, 5)))
            ctx.set_option("geodesic_head_start", 0)
            ctx.set_option("geodesic_flavour", 0)
            print("%-11s E=%6d budget %3d lists %2d best ms  %s" % (obj, E, budget, cap, "  ".join(row)), flush=True)
