#!/usr/bin/env python3
"""Above the split launch's range (90 112 samples) and around the end of the hand-over (120 000): does the persistent front of
DESIGN.md §5.2d still pay?  Variants: default policy | split launch (one block per CU, predicted >= 56, 4 samples per CU) with the
hand-over at once (threshold 10) | hand-over at once without the split | no hand-over at all (lpt 2).  Interleaved on one device;
bit-identical by assertion.  Development aid (GPU box)."""
import sys

import torch

sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402

ctx = Context(0)


def mean(fn, reps=6):
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


VARIANTS = (("default", {}), ("split+handover", {"fd_split_max": 1 << 30, "handover_threshold": 10}),
            ("split c1", {"fd_split_max": 1 << 30, "handover_threshold": 10, "fd_split_group_cut": 1}),
            ("split s2048", {"fd_split_max": 1 << 30, "handover_threshold": 10, "fd_split_samples": 2048}),
            ("handover only", {"fd_split": 0, "handover_threshold": 10}))
RESET = {"fd_split": 1, "fd_split_max": 90112, "handover_threshold": -1, "fd_split_group_cut": -1, "fd_split_samples": -1}
for obj in sys.argv[1:] or ["Wine_Bottle", "stefan"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for B in (90112, 98304, 114688, 131072, 163840, 196608, 262144):
        q = c.ambient_uniform_batch(0xC3, 0, B)
        out = torch.empty_like(q)
        c.project_batch(q, out=out)
        torch.cuda.synchronize()
        ref = out.clone()
        best = {}
        for rnd in range(2):
            for name, opts in VARIANTS:
                for k, v in RESET.items():
                    ctx.set_option(k, v)
                for k, v in opts.items():
                    ctx.set_option(k, v)
                c.project_batch(q, out=out)
                torch.cuda.synchronize()
                assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), (obj, B, name)
                best[name] = min(best.get(name, 1e9), mean(lambda: c.project_batch(q, out=out)))
        for k, v in RESET.items():
            ctx.set_option(k, v)
        print("%-11s B=%6d ms  %s" % (obj, B, "  ".join("%s %.3f (%+.1f %%)" % (n, best[n], 100 * (best[n] / best["default"] - 1)) for n, _ in VARIANTS)), flush=True)
