#!/usr/bin/env python3
"""Bulk extend calls (lists of 16, 128 rounds per edge): geodesic_flat_kernel alone against the hybrid — short edges on the
throughput layout (geodesic_group_kernel), the front of the scout's order on latency blocks beside it.  Interleaved on one device;
every output compared bit for bit.  GEO_CFGS="pred,front_per_cu,waves_per_cu[,permille[,handover_pct]];..." ("-1,-1,8" = the default
policy) and GEO_SIZES override.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from measure import near_edges  # noqa: E402

ctx = Context(0)


def mean(fn, reps=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


CFGS = [(32, 4, 8), (24, 4, 8), (16, 4, 8), (48, 4, 8), (32, 2, 10), (32, 4, 6)]
if os.environ.get("GEO_CFGS"):
    CFGS = [tuple(int(v) for v in c.split(",")) for c in os.environ["GEO_CFGS"].split(";")]
SIZES = [int(v) for v in os.environ.get("GEO_SIZES", "8192,16384,32768,65536").split(",")]
if os.environ.get("GEO_SCOUT_ROUNDS"):  # the scout's cap of predicted rounds (default 64) — applies to every variant, the baseline too
    ctx.set_option("geodesic_scout_rounds", int(os.environ["GEO_SCOUT_ROUNDS"]))
for obj in sys.argv[1:] or ["Wine_Bottle", "stefan"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for E in SIZES:
        frm, to = near_edges(c, E)
        call = lambda: c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)
        ctx.set_option("geodesic_group", 0)
        ref = call()
        torch.cuda.synchronize()
        live = torch.arange(16, device=frm.device)[None, :] < ref[1].clamp(max=16)[:, None]
        best = {}
        for rnd in range(2):
            for cfg in [None] + CFGS:
                ctx.set_option("geodesic_group", 0 if cfg is None else 1)
                ctx.set_option("geodesic_group_min", 16384 if cfg and cfg[0] == -1 else 0)  # "-1,-1,8" = the default policy
                if cfg:
                    ctx.set_option("geodesic_group_pred", cfg[0])
                    ctx.set_option("geodesic_group_front_per_cu", cfg[1])
                    ctx.set_option("geodesic_group_waves_per_cu", cfg[2])
                    ctx.set_option("geodesic_group_permille", cfg[3] if len(cfg) > 3 else 0)
                    ctx.set_option("geodesic_group_handover_pct", cfg[4] if len(cfg) > 4 else (50 if cfg[0] == -1 else 0))
                got = call()
                torch.cuda.synchronize()
                same = all(torch.equal(got[i], ref[i]) for i in (1, 2, 3, 4)) and torch.equal(got[0][live], ref[0][live])
                assert same, (obj, E, cfg, [bool(torch.equal(got[i], ref[i])) for i in (1, 2, 3, 4)])
                name = "flat alone" if cfg is None else ("default" if cfg[0] == -1 else "p%d/f%d/w%d" % cfg[:3]) + ("/m%d" % cfg[3] if len(cfg) > 3 and cfg[3] else "") + ("/h%d" % cfg[4] if len(cfg) > 4 else "")
                best[name] = min(best.get(name, 1e9), mean(call))
        base = best["flat alone"]
        print("%-11s E=%6d ms  %s" % (obj, E, "  ".join("%s %.3f (%+.1f %%)" % (k, v, 100 * (v / base - 1)) for k, v in best.items())), flush=True)
ctx.set_option("geodesic_group", 1)
for name in ("geodesic_group_pred", "geodesic_group_front_per_cu"):
    ctx.set_option(name, -1)
ctx.set_option("geodesic_group_handover_pct", 50)
