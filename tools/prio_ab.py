#!/usr/bin/env python3
"""Wave priority 3 (s_setprio) for the head of a longest-first order: does the serial chain a launch ends on run faster when its
wavefronts are first in line on their SIMDs?  Four places: the latency kernel alone in the scout's order (latency_prio_cut), the
split launch's front blocks beside the throughput kernel (fd_split_prio), the hand-over pool (pool_prio_cut), the extend step
(geodesic_prio_cut).  Interleaved on one device; results compared bit for bit.  Development aid (GPU box)."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from measure import near_edges  # noqa: E402

ctx = Context(0)
BIG = 0xFFFFFFFF


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


def sweep(label, call, same, settings, rounds=3):
    """settings: list of (name, {option: value}); the first is the baseline"""
    best = {}
    ref = None
    for _ in range(rounds):
        for name, opts in settings:
            for k, v in opts.items():
                ctx.set_option(k, v)
            got = call()
            torch.cuda.synchronize()
            if ref is None:
                ref = got
            else:
                assert same(got, ref), (label, name)
            best[name] = min(best.get(name, 1e9), mean(call))
    for k, v in settings[0][1].items():
        ctx.set_option(k, v)
    base = best[settings[0][0]]
    print("%s  %s" % (label, "  ".join("%s %.3f (%+.1f %%)" % (n, best[n], 100.0 * (best[n] / base - 1.0)) for n, _ in settings)), flush=True)


what = sys.argv[1] if len(sys.argv) > 1 else "all"
objs = sys.argv[2:] or ["Wine_Bottle", "stefan"]
for obj in objs:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    if what in ("all", "small"):
        for B in (3072, 4096, 8192, 14336):
            q = c.ambient_uniform_batch(0xC2, 0, B)
            out = torch.empty_like(q)

            def call():
                c.project_batch(q, out=out)
                return out.clone()

            sweep("%-11s latency alone B=%6d" % (obj, B), call, lambda a, b: torch.equal(a.view(torch.int64), b.view(torch.int64)),
                  [("off", {"latency_prio_cut": 0})] + [("first%d" % n, {"latency_prio_cut": n}) for n in (16, 64, 256, 1024)])
    if what in ("all", "mid"):
        for B in (16384, 24576, 32768, 57344, 81920, 131072):
            q = c.ambient_uniform_batch(0xC3, 0, B)
            out = torch.empty_like(q)

            def call():
                c.project_batch(q, out=out)
                return out.clone()

            sweep("%-11s mid-size      B=%6d" % (obj, B), call, lambda a, b: torch.equal(a.view(torch.int64), b.view(torch.int64)),
                  [("off", {"fd_split_prio": 0, "pool_prio_cut": 0}), ("front", {"fd_split_prio": 1, "pool_prio_cut": 0}),
                   ("front+pool64", {"fd_split_prio": 1, "pool_prio_cut": 64}), ("front+pool512", {"fd_split_prio": 1, "pool_prio_cut": 512}),
                   ("pool64", {"fd_split_prio": 0, "pool_prio_cut": 64}), ("front+pool-all", {"fd_split_prio": 1, "pool_prio_cut": BIG})])
    if what in ("all", "extend"):
        for E in (16384,):
            frm, to = near_edges(c, E)

            def call():
                return c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)

            sweep("%-11s extend step   E=%6d" % (obj, E), call, lambda a, b: all(torch.equal(a[i], b[i]) for i in (1, 2, 3)),
                  [("off", {"geodesic_prio_cut": 0})] + [("first%d" % n, {"geodesic_prio_cut": n}) for n in (32, 128, 512, 2048)])
