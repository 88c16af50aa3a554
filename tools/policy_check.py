#!/usr/bin/env python3
"""Does every size boundary of the scheduling policy still pay on THIS box?  (a dozen thresholds were tuned on
boxes that differ by 5-10 %, and nothing checked them where the suite runs.)

For each boundary of ccmp_policy.cpp — projector: latency_order_min, small_batch, the wide / narrow split shape, the front's length,
the occupancy hand-over rule, fd_split_max, the end of the hand-over; extend step with a round budget: the latency build's
capacity, geodesic_order_min = geodesic_scout_min, geodesic_group_min, the low cut — the call is timed 512 below and 512 above the
boundary, under the default policy AND with the neighbouring regime forced through ccmp_ctx_set_option (interleaved on one
device, warm clocks, best mean of several rounds; every output compared bit for bit with the default's).  A boundary FAILS when

  * the default is more than 5 % slower than the forced neighbour on BOTH objects (Wine_Bottle, stefan) on the same side, or
  * the time per sample / edge RISES by more than 12 % from below the boundary to above it under the default policy (a fall is
    amortisation: a batch of about one fill of the latency kernels lasts as long as its longest sample whatever its size).

    python tools/policy_check.py [--log profiles/r05_policy_check.log] [--reps 8] [--rounds 3]

ccmp_ctx_describe names the regime of every timed call in the log.  GPU box; tests/test_gpu_policy.py runs it in the suite."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402

CFG = os.path.join(ROOT, "tests", "golden", "config", "%s.yaml")
BIG = 1 << 40
OBJECTS = ("Wine_Bottle", "stefan")


def boundaries(cus):
    """(name, kind, boundary, options that force the regime ABOVE onto a call below it, options that force the regime BELOW onto a
    call above it); kind 'p' = project_batch, 'g' = budgeted extend call"""
    wide = {"fd_split_pred": 40, "fd_split_front": 2 * cus, "fd_split_samples": 4 * cus, "fd_split_group_cut": 3}
    narrow = {"fd_split_pred": 56, "fd_split_front": cus, "fd_split_samples": 3 * cus, "fd_split_group_cut": 2}
    D = lambda name: _lib.get_option(None, name)  # the library's own built-in value: the check follows the code, not a copy of it
    return [
        ("latency_order_min", "p", max(D("latency_order_min"), 2560), {"latency_order_min": 0}, {"latency_order_min": BIG}),
        ("small_batch", "p", D("small_batch"), {"small_batch": 0}, {"small_batch": BIG}),
        ("split shape (kSplitWideMax)", "p", 24576, narrow, wide),
        ("front length (3 -> 4 samples per CU)", "p", 40960, {"fd_split_samples": 4 * cus}, {"fd_split_samples": 3 * cus}),
        ("occupancy hand-over (kOccupancyHandoverBelow)", "p", 53248, {"handover_threshold": 10}, {"handover_threshold": 80}),
        ("fd_split_max", "p", D("fd_split_max"), {"fd_split_max": 0}, {"fd_split_max": BIG}),
        ("end of the hand-over (kNoHandoverFrom)", "p", 131072, {"lpt": 2}, {"handover_threshold": 10}),
        ("latency build's capacity", "g", 4 * cus, {"geodesic_flavour": 1}, {"geodesic_flavour": 2}),
        ("geodesic_order_min = geodesic_scout_min", "g", max(D("geodesic_scout_min"), 3072), {"geodesic_order_min": 0, "geodesic_scout_min": 0},
         {"geodesic_order_min": BIG, "geodesic_scout_min": BIG}),
        ("geodesic_group_min", "g", D("geodesic_group_min"), {"geodesic_group_min": 0}, {"geodesic_group_min": BIG}),
        ("low cut (kGeoGroupHighCut)", "g", 20480, {"geodesic_group_low_cut": 48}, {"geodesic_group_low_cut": 40}),
        ("group kernel's hand-over (kGeoGroupLateHandoverFrom)", "g", 32768, {"geodesic_group_handover_pct": 80}, {"geodesic_group_handover_pct": 50}),
        ("low cut (kGeoGroupHigherCut)", "g", 65536, {"geodesic_group_low_cut": 56}, {"geodesic_group_low_cut": 48}),
    ]


def mean_ms(fn, reps):
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


def near_edges(c, E, seed=0x6E0, dist=0.6):
    q, ok, _, _ = c.sample_project_batch(seed, 0, 8 * E, want_iters=False)
    frm = q[ok == 1][:E].contiguous()
    to, _, _, _ = c.sample_near_project_batch(seed + 1, 0, frm, dist, E, want_iters=False)
    return frm, to


def run(reps=8, rounds=3, log=None, out=print):
    # ONE context for both policies (the forced regime's options are set for its timed calls and put back): inside the GPU suite's
    # process, where dozens of contexts had lived before, a second context measured 8-13 % faster than the first on every forked
    # launch whatever its options said (streams share hardware queues) — a comparison across contexts compares the contexts.
    ctx = Context(0)
    cus = ctx.num_cus
    lines, failures = [], []

    def say(s):
        lines.append(s)
        out(s)

    say("policy check on %s, %d CUs; ms = best mean of %d rounds x %d calls; +/-512 around each boundary" % (torch.cuda.get_device_name(0), cus, rounds, reps))
    cons = {obj: KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx) for obj in OBJECTS}
    edges = {}
    for name, kind, b, force_above, force_below in boundaries(cus):
        slow = {"below": [], "above": []}
        for obj in OBJECTS:
            c = cons[obj]
            per_unit = {}
            for side, n, forced in (("below", b - 512, force_above), ("above", b + 512, force_below)):
                if kind == "p":
                    q = c.ambient_uniform_batch(0x9C, 0, n)
                    call = lambda: c.project_batch(q)
                    same = lambda x, y: all(torch.equal(u, v) for u, v in zip(x, y))
                    ck = _lib.CALL_PROJECT
                else:
                    if (obj, n) not in edges:
                        edges[(obj, n)] = near_edges(c, n)
                    frm, to = edges[(obj, n)]
                    call = lambda: c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)
                    same = lambda x, y: all(torch.equal(x[i], y[i]) for i in (1, 2, 3, 4))
                    ck = _lib.CALL_GEODESIC_BUDGET

                def force(on):
                    for k, v in forced.items():
                        ctx.set_option(k, v if on else _lib.get_option(None, k))

                ref = call()
                desc_d = ctx.describe(ck, n)
                force(True)
                got = call()
                desc_a = ctx.describe(ck, n)
                torch.cuda.synchronize()
                force(False)
                assert same(ref, got), ("forced regime changed a result", name, obj, n, forced)
                t_d, t_a = 1e9, 1e9
                for _ in range(rounds):
                    t_d = min(t_d, mean_ms(call, reps))
                    force(True)
                    t_a = min(t_a, mean_ms(call, reps))
                    force(False)
                per_unit[side] = t_d / n
                rel = t_d / t_a - 1.0
                if rel > 0.05:
                    slow[side].append(obj)
                say("%-46s %-11s n=%6d  default %.3f ms | forced neighbour %.3f ms (%+5.1f %%)%s" % (name, obj, n, t_d, t_a, 100 * rel, "  <-- slower" if rel > 0.05 else ""))
                say("    default: %s" % desc_d)
                say("    forced:  %s" % desc_a)
            # a batch of about one fill of the latency kernels lasts as long as its longest sample whatever its size, so the time per
            # unit FALLS across such a boundary (amortisation); what a mis-placed boundary shows as is a RISE
            jump = per_unit["above"] / per_unit["below"] - 1.0
            say("%-46s %-11s time per unit across the boundary: %.1f ns -> %.1f ns (%+.1f %%)%s"
                % (name, obj, per_unit["below"] * 1e6, per_unit["above"] * 1e6, 100 * jump, "  <-- jump" if jump > 0.12 else ""))
            if jump > 0.12:
                failures.append("%s / %s: time per unit rises by %.1f %% across the boundary" % (name, obj, 100 * jump))
        for side in ("below", "above"):
            if len(slow[side]) == len(OBJECTS):
                failures.append("%s: the default is > 5 %% slower than the neighbouring regime %s the boundary on every object" % (name, side))
    say("RESULT: %s" % ("all boundaries pay" if not failures else "; ".join(failures)))
    if log:
        os.makedirs(os.path.dirname(os.path.abspath(log)), exist_ok=True)
        open(log, "w").write("\n".join(lines) + "\n")
    return failures, lines


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--log", default="")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    bad, _ = run(a.reps, a.rounds, a.log or None)
    sys.exit(1 if bad else 0)
