#!/usr/bin/env python3
"""Round-5 experiments (GPU box, from the repo root):

    python tools/exp_r5.py geo_rounds [obj ...]     rounds per edge of a budgeted bulk extend call (lists of 16 + 128 rounds):
                                                    what the scheduling of geodesic_group_kernel + front + hand-over has to balance
    python tools/exp_r5.py bulk_ab [obj ...]        bulk extend calls under named option sets, interleaved on one device, every
                                                    output compared bit for bit with the latency kernel alone.
                                                    R5_CFGS="name:opt=val,opt=val;name2:..." (a name starting with "B/" runs on
                                                    the second build, R5_LIBB or lib/libccmp_B.so: tools/ab.py build "<flags>"),
                                                    R5_SIZES="16384,65536", R5_REPS=8, R5_ROUNDS=2
    python tools/exp_r5.py timeline [E [obj]]       five bulk calls for `rocprofv3 --kernel-trace` (the program behind `--`)
    python tools/exp_r5.py timeline_report <dir>    the last call of that trace on the device's clock
"""
import csv
import ctypes as C
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402
from measure import CFG, near_edges, timed  # noqa: E402

vp = C.c_void_p


def geo_rounds(argv):
    ctx = Context(0)
    for obj in argv or ["Wine_Bottle", "stefan"]:
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        for E in (16384, 65536):
            frm, to = near_edges(c, E)
            st, n, okg, its, carry = c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)
            torch.cuda.synchronize()
            n_, its_ = n.cpu().numpy().clip(max=16), its.cpu().numpy()
            rounds = its_ + n_ - 1
            print("%s E=%d: rounds per edge mean %.1f  p50/p75/p90/p99/p99.9/max %s %d; total %d" %
                  (obj, E, rounds.mean(), np.percentile(rounds, [50, 75, 90, 99, 99.9]).round(0), rounds.max(), rounds.sum()))
            for cut in (8, 12, 16, 24, 32, 40, 48, 64, 96, 128):
                m = rounds >= cut
                print("   edges with >= %3d rounds: %6d (%.1f %%), carrying %.1f %% of the rounds" % (cut, m.sum(), 100.0 * m.mean(), 100.0 * rounds[m].sum() / rounds.sum()))
            print("   ok==2 (budget spent): %d, list full: %d" % ((okg == 2).sum().item(), (n > 16).sum().item()))
            for on in (0, 1):
                ctx.set_option("geodesic_group", on)
                ms = timed(lambda: c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128), reps=8)
                print("   geodesic_group=%d: %.3f ms = %.2f M edges/s" % (on, ms, E / ms / 1e3), flush=True)


class Raw:
    """one context of one build of libccmp, called through ctypes (the package binds a single library)"""

    def __init__(self, L):
        self.L = L
        L.ccmp_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.ccmp_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_long]
        L.ccmp_geodesic_batch_ex.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
        self.h = vp()
        assert L.ccmp_ctx_create(0, C.byref(self.h)) == 0

    def set(self, name, value):
        rc = self.L.ccmp_ctx_set_option(self.h, name.encode(), int(value))
        assert rc == 0, (name, value, rc)

    def bulk(self, P, frm, to, out, max_states=16, budget=128, check_target=0):
        st, n, ok, its, carry = out
        rc = self.L.ccmp_geodesic_batch_ex(self.h, C.byref(P), frm.data_ptr(), to.data_ptr(), frm.shape[0], max_states, st.data_ptr(), n.data_ptr(),
                                           ok.data_ptr(), its.data_ptr(), None, carry.data_ptr(), budget, check_target, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc


def new_out(E, dev, max_states=16):
    return (torch.zeros((E, max_states, 14), dtype=torch.float64, device=dev), torch.zeros(E, dtype=torch.int32, device=dev),
            torch.zeros(E, dtype=torch.uint8, device=dev), torch.zeros(E, dtype=torch.int32, device=dev), torch.zeros((E, 2), dtype=torch.float64, device=dev))


def mean_ms(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


DEFAULT_CFGS = "flat:geodesic_group=0;r4:geodesic_group_live=0;live"


def bulk_ab(argv):
    cfgs = []
    for item in os.environ.get("R5_CFGS", DEFAULT_CFGS).split(";"):
        name, _, opts = item.partition(":")
        cfgs.append((name, [(o.split("=")[0], int(o.split("=")[1])) for o in opts.split(",") if o]))
    sizes = [int(v) for v in os.environ.get("R5_SIZES", "16384,65536").split(",")]
    reps, rounds = int(os.environ.get("R5_REPS", "8")), int(os.environ.get("R5_ROUNDS", "2"))
    LA = _lib.lib()
    LB = None
    if any(n.startswith("B/") for n, _ in cfgs):
        LB = C.CDLL(os.environ.get("R5_LIBB", os.path.join("closed_chain_motion_planner_amd", "lib", "libccmp_B.so")))
    ctx = Context(0)
    raws = {}
    for name, opts in cfgs:
        r = Raw(LB if name.startswith("B/") else LA)
        for k, v in opts:
            r.set(k, v)
        raws[name] = r
    ref_raw = Raw(LA)
    ref_raw.set("geodesic_group", 0)
    for obj in argv or ["Wine_Bottle", "stefan"]:
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        P = c.problem
        for E in sizes:
            frm, to = near_edges(c, E)
            ref = new_out(E, frm.device)
            ref_raw.bulk(P, frm, to, ref)
            torch.cuda.synchronize()
            live = torch.arange(16, device=frm.device)[None, :] < ref[1].clamp(max=16)[:, None]
            best = {}
            for rnd in range(rounds):
                for name, _ in cfgs:
                    r = raws[name]
                    got = new_out(E, frm.device)
                    r.bulk(P, frm, to, got)
                    torch.cuda.synchronize()
                    same = all(torch.equal(got[i], ref[i]) for i in (1, 2, 3, 4)) and torch.equal(got[0][live], ref[0][live])
                    assert same, (obj, E, name, [bool(torch.equal(got[i], ref[i])) for i in (1, 2, 3, 4)])
                    best[name] = min(best.get(name, 1e9), mean_ms(lambda: r.bulk(P, frm, to, got), reps))
            base = best[cfgs[0][0]]
            print("%-11s E=%6d ms  %s" % (obj, E, "  ".join("%s %.3f (%+.1f %%)" % (k, v, 100 * (v / base - 1)) for k, v in best.items())), flush=True)


def one_ctx_ab(argv):
    """bulk_ab's comparison on ONE context whose options are switched between the calls: for option sets that make the context
    create hardware queues of its own (CU-masked streams) — a context per set would oversubscribe the device's queues and slow
    every one of them down, the baseline included.  R5_CFGS / R5_SIZES / R5_REPS / R5_ROUNDS as for bulk_ab."""
    cfgs = []
    for item in os.environ.get("R5_CFGS", "base").split(";"):
        name, _, opts = item.partition(":")
        cfgs.append((name, [(o.split("=")[0], int(o.split("=")[1])) for o in opts.split(",") if o]))
    sizes = [int(v) for v in os.environ.get("R5_SIZES", "16384,65536").split(",")]
    reps, rounds = int(os.environ.get("R5_REPS", "8")), int(os.environ.get("R5_ROUNDS", "2"))
    L = _lib.lib()
    ctx = Context(0)
    r, ref_raw = Raw(L), Raw(L)
    ref_raw.set("geodesic_group", 0)
    keys = sorted({k for _, opts in cfgs for k, _ in opts})
    defaults = {k: _lib.get_option(None, k) for k in keys}
    for obj in argv or ["Wine_Bottle", "stefan"]:
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        P = c.problem
        for E in sizes:
            frm, to = near_edges(c, E)
            ref = new_out(E, frm.device)
            ref_raw.bulk(P, frm, to, ref)
            torch.cuda.synchronize()
            live = torch.arange(16, device=frm.device)[None, :] < ref[1].clamp(max=16)[:, None]
            best = {}
            for rnd in range(rounds):
                for name, opts in cfgs:
                    for k in keys:
                        r.set(k, defaults[k])
                    for k, v in opts:
                        r.set(k, v)
                    got = new_out(E, frm.device)
                    r.bulk(P, frm, to, got)
                    torch.cuda.synchronize()
                    same = all(torch.equal(got[i], ref[i]) for i in (1, 2, 3, 4)) and torch.equal(got[0][live], ref[0][live])
                    assert same, (obj, E, name, [bool(torch.equal(got[i], ref[i])) for i in (1, 2, 3, 4)])
                    best[name] = min(best.get(name, 1e9), mean_ms(lambda: r.bulk(P, frm, to, got), reps))
            base = best[cfgs[0][0]]
            print("%-11s E=%6d ms  %s" % (obj, E, "  ".join("%s %.3f (%+.1f %%)" % (k, v, 100 * (v / base - 1)) for k, v in best.items())), flush=True)


def front_rounds(argv):
    """what a Newton round costs the front's blocks of a bulk extend call, by how much runs beside them (variant B built with
    -DCCMP_GEO_TRACE, AB_UNIT=ccmp_kernels_geo.hip: per-edge start / end stamps of geodesic_flat_kernel).  The front is held at two
    blocks per CU; geodesic_group_kernel beside it at 8, 4, 2, 1 wavefronts per CU.  If the rounds' slow-down beside the group
    kernel is a matter of sharing issue slots or LDS, it shrinks with the group kernel's wavefronts; if it is the two kernels'
    code not fitting the instruction cache together (42 + 32 KB against 64 KB per CU pair), it does not."""
    E = int(argv[0]) if argv else 16384
    LB = C.CDLL(os.environ.get("R5_LIBB", os.path.join("closed_chain_motion_planner_amd", "lib", "libccmp_B.so")))
    LB.ccmp_debug_geo_trace.argtypes = [vp, C.c_size_t]
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    P = c.problem
    frm, to = near_edges(c, E)
    r = Raw(LB)
    out = new_out(E, frm.device)
    for waves, front, label in ((8, 8, "default"), (8, 2, ""), (4, 2, ""), (2, 2, ""), (1, 2, ""), (0, 8, "latency kernel alone (geodesic_group=0)")):
        if waves:
            r.set("geodesic_group", 1)
            r.set("geodesic_group_waves_per_cu", waves)
            r.set("geodesic_group_front_per_cu", front)
        else:
            r.set("geodesic_group", 0)
        for _ in range(3):
            r.bulk(P, frm, to, out)
        torch.cuda.synchronize()
        ms = mean_ms(lambda: r.bulk(P, frm, to, out), 5)
        import time
        time.sleep(0.02)  # a gap in front of the traced call: its stamps are the only ones of the last milliseconds
        r.bulk(P, frm, to, out)
        torch.cuda.synchronize()
        tr = np.zeros(3 * E, dtype=np.uint64)
        assert LB.ccmp_debug_geo_trace(tr.ctypes.data, E) == 0
        tr = tr.reshape(E, 3)
        n, its = out[1].cpu().numpy().clip(max=16), out[3].cpu().numpy()
        rounds = (its + n - 1).astype(np.float64)
        # stamps of this call only (an edge the group kernel finished this time keeps the stamps of an earlier call's hand-over)
        torch.cuda.synchronize()
        fresh = tr[:, 0].astype(np.float64) > float(tr[:, 1].max()) - ms * 1e5  # 100 MHz stamps; the call before ended about `ms` before this one
        t0 = tr[fresh, 0].min()
        start, end = (tr[:, 0].astype(np.float64) - t0) / 100.0, (tr[:, 1].astype(np.float64) - t0) / 100.0
        start[~fresh] = 1e9
        per = (end - start) / np.maximum(rounds, 1.0)
        row = []
        for lo, hi in ((64, 96), (96, 130), (130, 1000)):
            m = (rounds >= lo) & (rounds < hi) & (start < 100.0) & (end > start)  # the front's own edges: started with the launch
            row.append("%d-%d rounds: %d edges, %.2f us per round (p10 %.2f, p90 %.2f)" % (lo, hi, m.sum(), np.median(per[m]) if m.sum() else 0,
                                                                                         np.percentile(per[m], 10) if m.sum() else 0, np.percentile(per[m], 90) if m.sum() else 0))
        print("group wavefronts per CU %d, front blocks per CU %d %s: call %.3f ms, last stamp %.0f us; edges that started in the first 100 us — %s" %
              (waves, front, label, ms, end[fresh].max(), "; ".join(row)), flush=True)


def edge_costs(argv):
    """where the time of an extend-step launch on the latency kernel alone goes, per edge (variant B with -DCCMP_GEO_TRACE as for
    front_rounds): duration of an edge against its Newton rounds (least squares: us per round + us per edge), the gap between one
    edge's end and the next one's start on the same block, and how busy the blocks are over the launch"""
    E = int(argv[0]) if argv else 16384
    LB = C.CDLL(os.environ.get("R5_LIBB", os.path.join("closed_chain_motion_planner_amd", "lib", "libccmp_B.so")))
    LB.ccmp_debug_geo_trace.argtypes = [vp, C.c_size_t]
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    P = c.problem
    frm, to = near_edges(c, E)
    r = Raw(LB)
    r.set("geodesic_group", 0)
    out = new_out(E, frm.device)
    for _ in range(3):
        r.bulk(P, frm, to, out)
    torch.cuda.synchronize()
    ms = mean_ms(lambda: r.bulk(P, frm, to, out), 5)
    import time
    time.sleep(0.02)
    r.bulk(P, frm, to, out)
    torch.cuda.synchronize()
    tr = np.zeros(3 * E, dtype=np.uint64)
    assert LB.ccmp_debug_geo_trace(tr.ctypes.data, E) == 0
    tr = tr.reshape(E, 3)
    n, its = out[1].cpu().numpy().clip(max=16).astype(np.float64), out[3].cpu().numpy().astype(np.float64)
    rounds = its + n - 1
    t0 = tr[:, 0].min()
    start, end = (tr[:, 0].astype(np.float64) - t0) / 100.0, (tr[:, 1].astype(np.float64) - t0) / 100.0
    dur = end - start
    A = np.stack([rounds, n, np.ones(E)], axis=1)
    coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
    print("E=%d, call %.3f ms, kernel spans %.0f us; %d rounds, %.1f per edge; duration of an edge = %.2f us x rounds + %.2f us x states + %.2f us (least squares, residual rms %.1f us)"
          % (E, ms, end.max(), rounds.sum(), rounds.mean(), coef[0], coef[1], coef[2], np.sqrt(np.mean((A @ coef - dur) ** 2))))
    for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 64), (64, 1000)):
        m = (rounds >= lo) & (rounds < hi)
        print("   edges with %3d-%3d rounds: %6d, median duration %.1f us = %.2f us per round" % (lo, hi, m.sum(), np.median(dur[m]), np.median(dur[m] / np.maximum(rounds[m], 1))))
    blk = (tr[:, 2] >> np.uint64(32)).astype(np.int64)
    gaps, busy = [], 0.0
    for b in np.unique(blk):
        idx = np.where(blk == b)[0]
        idx = idx[np.argsort(start[idx])]
        gaps.extend((start[idx][1:] - end[idx][:-1]).tolist())
        busy += dur[idx].sum()
    gaps = np.array(gaps)
    print("   %d blocks; inside edges %.1f %% of blocks x span; gap between an edge's end and the block's next start: median %.2f us, p90 %.2f, mean %.2f"
          % (len(np.unique(blk)), 100.0 * busy / (len(np.unique(blk)) * end.max()), np.median(gaps), np.percentile(gaps, 90), gaps.mean()))
    T = end.max()
    for k in range(8):
        lo, hi = k * T / 8, (k + 1) * T / 8
        print("   %5.0f-%5.0f us: %5d edges in flight at some point, %5d start" % (lo, hi, ((start < hi) & (end > lo)).sum(), ((start >= lo) & (start < hi)).sum()))


def timeline(argv):
    E = int(argv[0]) if argv else 16384
    obj = argv[1] if len(argv) > 1 else "Wine_Bottle"
    ctx = Context(0)
    for item in os.environ.get("R5_OPTS", "").split(","):
        if item:
            ctx.set_option(item.split("=")[0], int(item.split("=")[1]))
    c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
    frm, to = near_edges(c, E)
    for _ in range(5):
        c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)
        torch.cuda.synchronize()


def two_contexts(argv):
    """(under rocprofv3 --kernel-trace) two contexts in one process, the same split-launch call on each in turn: which hardware
    queues do their kernels land on, and do the front and the throughput kernel still overlap on both?"""
    B = int(argv[0]) if argv else 16384
    a, b = Context(0), Context(0)
    ca = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=a)
    cb = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=b)
    q = ca.ambient_uniform_batch(0x9C, 0, B)
    for c in (ca, cb, ca, cb):
        for _ in range(3):
            c.project_batch(q)
            torch.cuda.synchronize()
        print("context %s: %.3f ms" % ("A" if c is ca else "B", timed(lambda: c.project_batch(q), reps=5)), flush=True)


def two_contexts_report(argv):
    files = sorted(glob.glob(os.path.join(argv[0], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(files[-1])), key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        name = r["Kernel_Name"]
        short = "front/flat" if "project_fd_flat" in name else "group" if "project_fd_kernel" in name else "scout" if "scout" in name else None
        if short in ("front/flat", "group"):
            print("%-10s queue %s stream %s  %9.0f -> %9.0f us  (%6.0f us)  blocks %d" % (short, r["Queue_Id"], r.get("Stream_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
                  (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))


def resident_ab(argv):
    """single-state calls through the resident service kernel of two builds (lib/libccmp.so, lib/libccmp_B.so — tools/ab.py build with
    AB_UNIT=ccmp_kernels_resident.hip), launched path beside them: median us of ccmp_project_host / ccmp_is_satisfied_host"""
    import time

    LB = C.CDLL(os.environ.get("R5_LIBB", os.path.join("closed_chain_motion_planner_amd", "lib", "libccmp_B.so")))
    LA = _lib.lib()
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    far = c.ambient_uniform_batch(0xC1, 0, 72).cpu().numpy()
    q, ok, _ = c.project_batch(c.ambient_uniform_batch(0xC7, 0, 2048))
    near = q[ok == 1][:72].cpu().numpy() + np.random.default_rng(0xC7).uniform(-0.05, 0.05, (72, 14))
    torch.cuda.synchronize()
    dp = C.POINTER(C.c_double)
    res = {}
    for rnd in range(3):
        for name, L in (("A", LA), ("B", LB)):
            L.ccmp_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
            L.ccmp_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_long]
            L.ccmp_project_host.argtypes = [vp, vp, dp, dp, vp, vp, C.c_size_t]
            L.ccmp_is_satisfied_host.argtypes = [vp, vp, dp, vp, C.c_size_t]
            L.ccmp_ctx_destroy.argtypes = [vp]
            h = vp()
            assert L.ccmp_ctx_create(0, C.byref(h)) == 0
            for on in (0, 1):
                assert L.ccmp_ctx_set_option(h, b"resident", on) == 0
                for kind, xs in (("uniform", far), ("near", near)):
                    tp, ts = [], []
                    okb = (C.c_uint8 * 1)()
                    for x in xs:
                        xi, xo = np.ascontiguousarray(x), np.zeros(14)
                        t0 = time.perf_counter()
                        L.ccmp_project_host(h, C.byref(c.problem), xi.ctypes.data_as(dp), xo.ctypes.data_as(dp), okb, None, 1)
                        t1 = time.perf_counter()
                        L.ccmp_is_satisfied_host(h, C.byref(c.problem), xo.ctypes.data_as(dp), okb, 1)
                        t2 = time.perf_counter()
                        tp.append(t1 - t0), ts.append(t2 - t1)
                    key = (name, "resident" if on else "launched", kind)
                    res.setdefault(key, []).append((float(np.median(tp[8:]) * 1e6), float(np.median(ts[8:]) * 1e6)))
            L.ccmp_ctx_set_option(h, b"resident", 0)
            L.ccmp_ctx_destroy(h)
    for key in sorted(res):
        print("%s %-9s %-8s project %6.1f us   isSatisfied %5.1f us" % (key + (min(v[0] for v in res[key]), min(v[1] for v in res[key]))))


def timeline_report(argv):
    files = sorted(glob.glob(os.path.join(argv[0], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(files[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last call = from the last scout launch on
    last = max(i for i, r in enumerate(rows) if "scout_geodesic" in r["Kernel_Name"])
    first = last
    while first > 0 and "clear_words" in rows[first - 1]["Kernel_Name"]:
        first -= 1
    t0 = int(rows[first]["Start_Timestamp"])
    print("| kernel | queue | start us | end us | blocks |\n|---|---|---|---|---|")
    for r in rows[first:]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        print("| `%s` | %s | %.0f | %.0f | %d |" % (name, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                   int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) // max(1, int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1))))))


if __name__ == "__main__":
    {"geo_rounds": geo_rounds, "bulk_ab": bulk_ab, "one_ctx_ab": one_ctx_ab, "front_rounds": front_rounds, "edge_costs": edge_costs, "timeline": timeline, "timeline_report": timeline_report, "two_contexts": two_contexts,
     "two_contexts_report": two_contexts_report, "resident_ab": resident_ab}[sys.argv[1]](sys.argv[2:])
