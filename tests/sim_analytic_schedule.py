"""A discrete-event model of the analytic mode's schedules (round 6; DESIGN.md §5.4, DESIGN_experiments.md §13).  Test infrastructure
like tests/parity_campaign.py: the iteration counts of the batch come from the CPU oracle's analytic mode.

    python tests/sim_analytic_schedule.py [Wine_Bottle|stefan]

Inputs: the rounds every sample of the C3 batch needs (oracle, seed 0xC3), and the time of one Newton round of each kernel against
the number of wavefronts that share a SIMD, as measured on the MI355X (tools/exp_r6.py rounds -> profiles/r06_analytic_rounds.log).
Model: 1 024 SIMDs; persistent wavefronts (32 samples each for the lane-pair kernel, 4 for the latency kernel) refill finished slots
from one queue in index order; a round of a wavefront lasts t(kind, live wavefronts on its SIMD); generations run one after the other
(+ 8 us per launch), or — `concurrent` — the latency engine takes handed-over samples while the lane-pair kernel still runs.
It reproduced the measured calls of round 6 to +-10 % (sequential default 2.0 vs 2.39 ms before the ticket-word fix, whose cost it
does not model; 0.95 vs 0.91 ms for generation 0 after it) and is what the variants were screened with before they were built.
"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NS = 1024
PAIR = {1: 5.1, 2: 7.0, 3: 10.2, 4: 13.6}   # us per round, lane-pair kernel, k wavefronts per SIMD (4.2 for a lone wavefront on the chip)
ROW = {1: 3.1, 2: 4.2, 3: 6.3, 4: 8.4}      # sixteen-lanes-per-sample kernel (2.67 lone)


def load(name="Wine_Bottle", B=262144):
    from oracle_binding import Oracle

    O = Oracle("det")
    P = O.checker_problem(os.path.join(ROOT, "tests", "golden", "config", name + ".yaml"))
    P.jacobian_mode = 1
    q = O.ambient_uniform_batch(P, 0xC3, 0, B)
    _, _, it = O.project_batch(P, q, min(8, os.cpu_count() or 1))
    return np.asarray(it) + 1  # rounds = updates + 1


def run_gen(rem, age, kind, wps, dump, cap, t0):
    """one launch: rem / age = rounds still needed / done per sample; returns (end, handed over by the cap, by the dump rule, rem, age)"""
    n = len(rem)
    per = 32 if kind == "pair" else 4
    T = PAIR if kind == "pair" else ROW
    nw = min((n + per - 1) // per, NS * wps)
    live = np.zeros(NS, int)
    for w in range(nw):
        live[w % NS] += 1
    nxt, slots, ev, out_old, out_young = 0, [], [], [], []
    for w in range(nw):
        k = min(per, n - nxt)
        slots.append(list(range(nxt, nxt + k)))
        nxt += k
        heapq.heappush(ev, (t0 + T[live[w % NS]], w))
    r, a, end = rem.copy(), age.copy(), t0
    while ev:
        t, w = heapq.heappop(ev)
        keep = []
        for i in slots[w]:
            r[i] -= 1
            a[i] += 1
            if r[i] <= 0:
                continue
            if cap and a[i] >= cap:
                out_old.append(i)
                continue
            keep.append(i)
        while len(keep) < per and nxt < n:
            keep.append(nxt)
            nxt += 1
        if nxt >= n and dump is not None and len(keep) <= dump:
            out_young += keep
            keep = []
        slots[w] = keep
        if keep:
            heapq.heappush(ev, (t + T[max(1, live[w % NS])], w))
        else:
            live[w % NS] -= 1
            end = max(end, t)
    return end, out_old, out_young, r, a


def sim(it, plan):
    """plan: [(kind, wavefronts per SIMD, dump threshold or None, cap or 0, source in all / old / young / both)]"""
    rem, age, t = it.astype(int).copy(), np.zeros(len(it), int), 0.0
    old, young, log = [], [], []
    for g, (kind, wps, dump, cap, src) in enumerate(plan):
        if src == "all":
            idx = np.arange(len(it))
        elif src == "old":
            idx, old = np.array(old, int), []
        elif src == "young":
            idx, young = np.array(young, int), []
        else:
            idx, old, young = np.array(old + young, int), [], []
        if len(idx) == 0:
            continue
        end, oo, oy, r, a = run_gen(rem[idx], age[idx], kind, wps, dump, cap, t + (8.0 if g else 0.0))
        rem[idx], age[idx] = r, a
        old += list(idx[oo])
        young += list(idx[oy])
        log.append("%s n=%d %.0f->%.0f us (cap %d, dump %d)" % (kind, len(idx), t, end, len(oo), len(oy)))
        t = end
    return t, log


if __name__ == "__main__":
    it = load(sys.argv[1] if len(sys.argv) > 1 else "Wine_Bottle")
    print("rounds in all %d; at 9.65 G sample-rounds/s (three wavefronts per SIMD, no refills): %.0f us" % (it.sum(), it.sum() / 9650.0))
    for thr in (60, 100, 150, 200, 250):
        print("  samples with >= %d rounds: %d" % (thr, int((it >= thr + 1).sum())))
    plans = {
        "lane-pair kernel alone, 3 per SIMD": [("pair", 3, None, 0, "all")],
        "lane-pair kernel alone, 2 per SIMD": [("pair", 2, None, 0, "all")],
        "DEFAULT: hand-over at <= 8 per wavefront -> latency kernel": [("pair", 3, 8, 0, "all"), ("row", 2, None, 0, "both")],
        "hand-over at <= 16 -> latency kernel": [("pair", 3, 16, 0, "all"), ("row", 2, None, 0, "both")],
        "cap 56 + hand-over at <= 16 -> latency kernel": [("pair", 3, 16, 56, "all"), ("row", 2, None, 0, "both")],
        "2 per SIMD, cap 56, <= 16 -> latency kernel": [("pair", 2, 16, 56, "all"), ("row", 2, None, 0, "both")],
        "<= 24 -> lane-pair (2 per SIMD, cap 48, <= 8) -> latency": [("pair", 3, 24, 0, "all"), ("pair", 2, 8, 48, "both"), ("row", 2, None, 0, "both")],
        "cap 64, <= 24 -> lane-pair on the young (cap 64, <= 4) -> latency": [("pair", 3, 24, 64, "all"), ("pair", 2, 4, 64, "young"), ("row", 2, None, 0, "both")],
    }
    for k, p in plans.items():
        t, log = sim(it, p)
        print("%-62s %5.0f us | " % (k, t) + " ; ".join(log))
