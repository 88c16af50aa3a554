#!/usr/bin/env python3
"""round-3 scratch experiments on the GPU box (not part of the product)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

import os
os.environ.setdefault("CCMP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "closed_chain_motion_planner_amd", "lib", "libccmp_debug.so"))  # reads the scout's predictions: a hook of include/ccmp_debug.h
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402
from tools.measure import near_edges, timed, CFG  # noqa: E402


def geo_stats():
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    frm, to = near_edges(c, 16384)
    st, n, okg, its = c.discrete_geodesic_batch(frm, to, 1024)
    torch.cuda.synchronize()
    n_, its_ = n.cpu().numpy(), its.cpu().numpy()
    d = (to - frm).norm(dim=1).cpu().numpy()
    print("dist percentiles 1/50/99/max", np.percentile(d, [1, 50, 99]), d.max())
    print("n_states percentiles 50/90/99/99.9/max", np.percentile(n_, [50, 90, 99, 99.9]), n_.max())
    print("newton its percentiles 50/90/99/99.9/max", np.percentile(its_, [50, 90, 99, 99.9]), its_.max())
    rounds = its_ + n_  # one evaluation round per projected state on top of the updates
    print("rounds mean %.1f p50/p90/p99/p99.9/max" % rounds.mean(), np.percentile(rounds, [50, 90, 99, 99.9]), rounds.max())
    for cap in (4, 8, 12, 16, 24, 32, 64, 128):
        print("edges with n > %d: %d" % (cap, (n_ > cap).sum()))
    print("corr(dist/delta, n) = %.3f, corr(dist, rounds) = %.3f" % (np.corrcoef(d / 0.25, n_)[0, 1], np.corrcoef(d, rounds)[0, 1]))
    order = np.argsort(-rounds)
    print("top-10 rounds", rounds[order[:10]], "their n", n_[order[:10]], "their dist", d[order[:10]])
    for cap in (8, 16, 32, 64):
        ms = timed(lambda: c.discrete_geodesic_batch(frm, to, cap), reps=5)
        print("E=16384 cap=%d: %.3f ms -> %.2f M edges/s" % (cap, ms, 16384 / ms / 1e3), flush=True)
    # order sensitivity: longest-first by true rounds (upper bound for any predictor), cap 16 and 64
    idx = torch.as_tensor(order.copy(), device=frm.device)
    f2, t2 = frm[idx].contiguous(), to[idx].contiguous()
    for cap in (16, 64):
        ms = timed(lambda: c.discrete_geodesic_batch(f2, t2, cap), reps=5)
        print("E=16384 cap=%d longest-first (true rounds): %.3f ms" % (cap, ms), flush=True)
    idx = torch.as_tensor(np.argsort(-d).copy(), device=frm.device)
    f2, t2 = frm[idx].contiguous(), to[idx].contiguous()
    for cap in (16, 64):
        ms = timed(lambda: c.discrete_geodesic_batch(f2, t2, cap), reps=5)
        print("E=16384 cap=%d longest-first (dist): %.3f ms" % (cap, ms), flush=True)
    L = _lib.lib()
    for E in (1, 5, 64, 256, 1024, 4096):
        f, t = frm[:E].contiguous(), to[:E].contiguous()
        states = torch.empty((E, 64, 14), dtype=torch.float64, device=f.device)
        nn = torch.empty(E, dtype=torch.int32, device=f.device)
        ok = torch.empty(E, dtype=torch.uint8, device=f.device)
        it = torch.empty(E, dtype=torch.int32, device=f.device)
        s = torch.cuda.current_stream().cuda_stream

        def raw():
            L.ccmp_geodesic_batch(ctx.handle, C.byref(c.problem), f.data_ptr(), t.data_ptr(), E, 64, states.data_ptr(), nn.data_ptr(),
                                  ok.data_ptr(), it.data_ptr(), s)
        ms_raw = timed(raw, reps=10)
        ms_py = timed(lambda: c.discrete_geodesic_batch(f, t, 64), reps=10)
        raw()
        torch.cuda.synchronize()
        r = (it + nn).max().item()
        print("E=%d raw C call %.3f ms, python mirror %.3f ms, longest edge %d rounds -> %.2f us per round" % (E, ms_raw, ms_py, r, ms_raw * 1e3 / r), flush=True)
    # host-call latency for 5 edges
    sp_f, sp_t = frm[:5].cpu().numpy(), to[:5].cpu().numpy()
    from closed_chain_motion_planner_amd.space import jy_ProjectedStateSpace
    sp = jy_ProjectedStateSpace(c)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        sp.discreteGeodesicBatch(sp_f, sp_t, True)
        ts.append(time.perf_counter() - t0)
    print("space.discreteGeodesicBatch 5 edges host->host: median %.1f us" % (np.median(ts[5:]) * 1e6))




def geo_order():
    """extend step, 16384 edges, lists of 16: index order / far-apart first / FP32 scout order with several round caps"""
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    frm, to = near_edges(c, 16384)
    ref = c.discrete_geodesic_batch(frm, to, 16)
    for E in (16384, 8192, 65536):
        if E != 16384:
            frm, to = near_edges(c, E)
            ref = c.discrete_geodesic_batch(frm, to, 16)
        for name, order, rounds in (("index", 0, 48), ("far-first", 1, 48), ("scout32", 2, 32), ("scout48", 2, 48), ("scout64", 2, 64), ("scout96", 2, 96)):
            ctx.set_option("geodesic_order", order)
            ctx.set_option("geodesic_scout_min", 0)
            ctx.set_option("geodesic_scout_rounds", rounds)
            ms = min(timed(lambda: c.discrete_geodesic_batch(frm, to, 16), reps=5) for _ in range(2))
            got = c.discrete_geodesic_batch(frm, to, 16)
            same = all(torch.equal(a, b) for a, b in zip(got[1:], ref[1:]))
            print("E=%d %-10s %.3f ms  %.2f M edges/s  same counts/flags/iterations %s" % (E, name, ms, E / ms / 1e3, same), flush=True)
    L = _lib.lib()
    L.ccmp_ctx_debug_lpt_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    frm, to = near_edges(c, 16384)
    ctx.set_option("geodesic_scout_rounds", 255)
    st, n, ok, its = c.discrete_geodesic_batch(frm, to, 16)
    pred = np.zeros(16384, dtype=np.uint16)
    assert L.ccmp_ctx_debug_lpt_pred(ctx.handle, pred.ctypes.data, 16384) == 0
    rounds = (its + n.clamp(max=16) - 1).cpu().numpy()
    p = pred.astype(np.int64)
    print("scout vs true rounds (cap 255): corr %.3f, |d|<=2: %.3f, of the %d edges with > 60 true rounds the scout says > 40 for %.3f"
          % (np.corrcoef(p, rounds)[0, 1], (np.abs(p - rounds) <= 2).mean(), (rounds > 60).sum(), (p[rounds > 60] > 40).mean()))


def phases():
    """cycle counters per phase of the latency kernel's Newton round (variant B built with -DCCMP_FLAT_TIMING for the flat unit)"""
    import os
    LIBB = os.path.join("closed_chain_motion_planner_amd", "lib", "libccmp_B.so")
    from closed_chain_motion_planner_amd import load_config
    LB = C.CDLL(LIBB)
    P = load_config(CFG % "Wine_Bottle")
    h = C.c_void_p()
    LB.ccmp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    assert LB.ccmp_ctx_create(0, C.byref(h)) == 0
    LB.ccmp_project_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    LB.ccmp_ambient_uniform_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]
    LB.ccmp_debug_flat_timing.argtypes = [C.c_void_p, C.c_int]
    q = torch.empty((64, 14), dtype=torch.float64, device="cuda")
    LB.ccmp_ambient_uniform_batch(h, C.byref(P), 0xC1, 0, q.data_ptr(), 64, None)
    out = torch.empty_like(q)
    ok = torch.empty(64, dtype=torch.uint8, device="cuda")
    it = torch.empty(64, dtype=torch.int16, device="cuda")
    t = (C.c_ulonglong * 8)()
    LB.ccmp_debug_flat_timing(t, 1)
    iters = 0
    for i in range(64):
        assert LB.ccmp_project_batch(h, C.byref(P), q[i:i + 1].data_ptr(), out[i:i + 1].data_ptr(), ok[i:i + 1].data_ptr(), it[i:i + 1].data_ptr(), 1, None) == 0
    torch.cuda.synchronize()
    iters = int(it.to(torch.int32).sum())
    LB.ccmp_debug_flat_timing(t, 0)
    tot = sum(t)
    names = ["A angles", "B chain", "C residual + stencil", "(3)", "E solve + update", "", "", ""]
    rounds = iters + 64
    print("64 single-state projections, %d Newton updates, %d rounds" % (iters, rounds))
    for k in range(5):
        print("  %-22s %8.0f cycles per round  %5.1f %%" % (names[k], t[k] / rounds, 100.0 * t[k] / tot))
    print("  total %.0f cycles per round (100 MHz-ish counter? s_memrealtime vs clock: compare with the wall-clock per round)" % (tot / rounds))




def ab_latency():
    """A (libccmp.so) against B (libccmp_B.so), interleaved: single-state project, 5-edge and 16384-edge extend steps, 4096 batch"""
    import os, statistics
    from closed_chain_motion_planner_amd import load_config
    LA = _lib.lib()
    LB = C.CDLL(os.path.join("closed_chain_motion_planner_amd", "lib", "libccmp_B.so"))
    P = load_config(CFG % "Wine_Bottle")
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    frm, to = near_edges(c, 16384)
    vp = C.c_void_p
    H = {}
    for name, L in (("A", LA), ("B", LB)):
        h = C.c_void_p()
        L.ccmp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        assert L.ccmp_ctx_create(0, C.byref(h)) == 0
        L.ccmp_project_batch.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
        L.ccmp_geodesic_batch.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, vp, vp, vp, vp, vp]
        L.ccmp_project_host.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t]
        L.ccmp_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_long]
        if name == "B" and os.environ.get("AB_B_BLOCKS"):  # variant B with another number of persistent blocks per CU
            assert L.ccmp_ctx_set_option(h, b"latency_blocks_per_cu", int(os.environ["AB_B_BLOCKS"])) == 0
        if name == "B" and os.environ.get("AB_B_GEO_BLOCKS"):
            assert L.ccmp_ctx_set_option(h, b"geodesic_blocks_per_cu", int(os.environ["AB_B_GEO_BLOCKS"])) == 0
        L.ccmp_geodesic_batch_ex.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
        H[name] = (L, h)
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for E, cap in ((5, 64), (1024, 64), (16384, 16), (16384, 64)):
        f, t = frm[:E].contiguous(), to[:E].contiguous()
        states = {n: torch.empty((E, cap, 14), dtype=torch.float64, device=f.device) for n in H}
        nn = torch.empty(E, dtype=torch.int32, device=f.device)
        ok = torch.empty(E, dtype=torch.uint8, device=f.device)
        it = torch.empty(E, dtype=torch.int32, device=f.device)
        ts = {n: [] for n in H}
        for rep in range(8):
            for n, (L, h) in H.items():
                fn = lambda: L.ccmp_geodesic_batch(h, C.byref(P), f.data_ptr(), t.data_ptr(), E, cap, states[n].data_ptr(), nn.data_ptr(), ok.data_ptr(), it.data_ptr(), s)
                ts[n].append(timed(fn, reps=1))
        same = torch.equal(states["A"][:, :4], states["B"][:, :4])
        print("geodesic E=%d cap=%d: A %.3f ms  B %.3f ms  (B/A %.3f) same %s" % (E, cap, statistics.median(ts["A"]), statistics.median(ts["B"]),
              statistics.median(ts["B"]) / statistics.median(ts["A"]), same), flush=True)
    # the bench's shape: lists of 16, 128 Newton rounds per edge and call
    E, cap = 16384, 16
    states = {n: torch.zeros((E, cap, 14), dtype=torch.float64, device=frm.device) for n in H}
    nn = {n: torch.zeros(E, dtype=torch.int32, device=frm.device) for n in H}
    ok = torch.empty(E, dtype=torch.uint8, device=frm.device)
    it = torch.empty(E, dtype=torch.int32, device=frm.device)
    carry = torch.empty((E, 2), dtype=torch.float64, device=frm.device)
    ts = {n: [] for n in H}
    for rep in range(10):
        for n, (L, h) in H.items():
            fn = lambda: L.ccmp_geodesic_batch_ex(h, C.byref(P), frm.data_ptr(), to.data_ptr(), E, cap, states[n].data_ptr(), nn[n].data_ptr(), ok.data_ptr(),
                                                  it.data_ptr(), None, carry.data_ptr(), 128, 0, s)
            ts[n].append(timed(fn, reps=1))
    print("geodesic E=16384 cap=16 budget=128: A %.3f ms  B %.3f ms  (B/A %.3f) same counts %s" % (statistics.median(ts["A"]), statistics.median(ts["B"]),
          statistics.median(ts["B"]) / statistics.median(ts["A"]), torch.equal(nn["A"], nn["B"])), flush=True)
    for B in (1, 256, 4096, 10240, 16384, 32768):
        q = c.ambient_uniform_batch(0xC2, 0, B)
        out = {n: torch.empty_like(q) for n in H}
        ok = torch.empty(B, dtype=torch.uint8, device=q.device)
        it = torch.empty(B, dtype=torch.int16, device=q.device)
        ts = {n: [] for n in H}
        for rep in range(8):
            for n, (L, h) in H.items():
                fn = lambda: L.ccmp_project_batch(h, C.byref(P), q.data_ptr(), out[n].data_ptr(), ok.data_ptr(), it.data_ptr(), B, s)
                ts[n].append(timed(fn, reps=1))
        print("project B=%d: A %.4f ms  B %.4f ms  (B/A %.3f) same %s" % (B, statistics.median(ts["A"]), statistics.median(ts["B"]),
              statistics.median(ts["B"]) / statistics.median(ts["A"]), torch.equal(out["A"], out["B"])), flush=True)
    x = c.ambient_uniform_batch(0xC1, 0, 64).cpu().numpy()
    for n, (L, h) in H.items():
        ts = []
        for i in range(64):
            xi, xo = x[i].copy(), np.zeros(14)
            okb = (C.c_uint8 * 1)()
            t0 = time.perf_counter()
            L.ccmp_project_host(h, C.byref(P), xi.ctypes.data, xo.ctypes.data, okb, None, 1)
            ts.append(time.perf_counter() - t0)
        print("%s single project_host: median %.1f us" % (n, np.median(ts[8:]) * 1e6))




def handover():
    """mid-size batches: occupancy threshold of the hand-over (throughput kernel -> latency kernel), scout on/off"""
    ctx = Context(0)
    for obj in ("Wine_Bottle", "stefan"):
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        for B in [int(b) for b in os.environ.get("HO_SIZES", "12288,16384,24576,32768,49152,65536").split(",")]:
            q = c.ambient_uniform_batch(0xC3, 0, B)
            out = torch.empty_like(q)
            row = []
            ctx.set_schedule(2, 0)
            row.append("flat-only %.3f" % timed(lambda: c.project_batch(q, out=out), reps=5))
            ctx.set_schedule(1)
            for lpt_min in (0, 1 << 30):
                ctx.set_lpt(1, lpt_min)
                for thr in (-1, 60, 70, 80, 90, 100):
                    ctx.set_option("handover_threshold", thr if thr < 0 else 10 + thr)
                    row.append("%s%s %.3f" % ("scout " if lpt_min == 0 else "", "auto" if thr < 0 else "%d%%" % thr, timed(lambda: c.project_batch(q, out=out), reps=5)))
            ctx.set_option("handover_threshold", -1)
            ctx.set_lpt(1)
            print("%s B=%d: %s" % (obj, B, " | ".join(row)), flush=True)






def blocks_per_cu():
    """persistent blocks of the latency kernel per CU: extend step and projector at loaded sizes"""
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    frm, to = near_edges(c, 16384)
    q4, q16 = c.ambient_uniform_batch(0xC2, 0, 4096), c.ambient_uniform_batch(0xC2, 0, 10240)
    o4, o16 = torch.empty_like(q4), torch.empty_like(q16)
    for k in (2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
        ctx.set_option("latency_blocks_per_cu", k)
        ctx.set_option("geodesic_blocks_per_cu", k)
        g = min(timed(lambda: c.discrete_geodesic_batch(frm, to, 16), reps=5) for _ in range(2))
        g2 = min(timed(lambda: c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128), reps=5) for _ in range(2))
        g3 = min(timed(lambda: c.discrete_geodesic_batch(frm[:2048], to[:2048], 16), reps=5) for _ in range(2))
        print("blocks per CU %2d: 16384 edges, 128 rounds %.3f ms | 2048 edges %.3f ms" % (k, g2, g3))
        a = min(timed(lambda: c.project_batch(q4, out=o4), reps=5) for _ in range(2))
        b = min(timed(lambda: c.project_batch(q16, out=o16), reps=5) for _ in range(2))
        print("blocks per CU %2d: 16384 edges %.3f ms | 4096 samples %.3f ms | 10240 samples %.3f ms" % (k, g, a, b), flush=True)
    ctx.set_option("latency_blocks_per_cu", 8)
    ctx.set_option("geodesic_blocks_per_cu", 4)






def geo_trace():
    """timeline of a 16384-edge extend-step launch (variant B built with -DCCMP_GEO_TRACE, AB_UNIT=ccmp_kernels_geo.hip: the
    throughput flavour, which the traced context is pinned to)"""
    import os
    from closed_chain_motion_planner_amd import load_config
    LB = C.CDLL(os.path.join("closed_chain_motion_planner_amd", "lib", "libccmp_B.so"))
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    vp = C.c_void_p
    h = C.c_void_p()
    LB.ccmp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    assert LB.ccmp_ctx_create(0, C.byref(h)) == 0
    LB.ccmp_geodesic_batch.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, vp, vp, vp, vp, vp]
    LB.ccmp_debug_geo_trace.argtypes = [vp, C.c_size_t]
    LB.ccmp_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    assert LB.ccmp_ctx_set_option(h, b"geodesic_flavour", 1) == 0
    P = c.problem
    for E, cap in ((16384, 16), (16384, 64)):
        frm, to = near_edges(c, E)
        states = torch.empty((E, cap, 14), dtype=torch.float64, device=frm.device)
        nn = torch.empty(E, dtype=torch.int32, device=frm.device)
        ok = torch.empty(E, dtype=torch.uint8, device=frm.device)
        it = torch.empty(E, dtype=torch.int32, device=frm.device)
        s = torch.cuda.current_stream().cuda_stream
        for order in (2, 0):
            LB.ccmp_ctx_set_option(h, b"geodesic_order", order)
            for _ in range(3):
                assert LB.ccmp_geodesic_batch(h, C.byref(P), frm.data_ptr(), to.data_ptr(), E, cap, states.data_ptr(), nn.data_ptr(), ok.data_ptr(), it.data_ptr(), s) == 0
            torch.cuda.synchronize()
            tr = np.zeros(3 * E, dtype=np.uint64)
            assert LB.ccmp_debug_geo_trace(tr.ctypes.data, E) == 0
            tr = tr.reshape(E, 3)
            t0 = tr[:, 0].min()
            start, end = (tr[:, 0] - t0).astype(np.float64) / 100.0, (tr[:, 1] - t0).astype(np.float64) / 100.0  # us
            rounds = (it + nn.clamp(max=cap) - 1).cpu().numpy().astype(np.float64)
            dur = end - start
            last = np.argsort(-end)[:12]
            print("E=%d lists of %d order %d: launch spans %.0f us; edges finishing last:" % (E, cap, order, end.max()))
            for e in last:
                print("   edge %5d ticket %5d block %4d start %7.1f end %7.1f us rounds %4d -> %.2f us per round" % (e, int(tr[e, 2] & 0xffffffff), int(tr[e, 2] >> 32), start[e], end[e], rounds[e], dur[e] / max(1.0, rounds[e])))
            for lo, hi in ((0, 200), (200, 500), (500, 1000), (1000, 1500), (1500, 2000), (2000, 3000), (3000, 1e9)):
                m = (start >= lo) & (start < hi)
                busy = ((start < hi) & (end > lo)).sum()
                if m.sum() or busy:
                    rr = dur[m] / np.maximum(1.0, rounds[m]) if m.sum() else np.zeros(1)
                    print("   window %5.0f-%5.0f us: %5d edges start, %5d in flight at some point, median us per round of those starting %.2f" % (lo, min(hi, end.max()), m.sum(), busy, np.median(rr)))




def flat_trace():
    """timeline of the latency kernel's part of a mid-size projection (variant B built with -DCCMP_GEO_TRACE): per pool
    ticket start / end and the rounds it ran"""
    import os
    LB = C.CDLL(os.path.join("closed_chain_motion_planner_amd", "lib", "libccmp_B.so"))
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    vp = C.c_void_p
    h = C.c_void_p()
    LB.ccmp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    assert LB.ccmp_ctx_create(0, C.byref(h)) == 0
    LB.ccmp_project_batch.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    LB.ccmp_debug_flat_trace.argtypes = [vp, C.c_size_t]
    P = c.problem
    for B in (32768, 65536, 4096, 10240):
        q = c.ambient_uniform_batch(0xC3 if B > 16384 else 0xC2, 0, B)
        out = torch.empty_like(q)
        ok = torch.empty(B, dtype=torch.uint8, device=q.device)
        it = torch.empty(B, dtype=torch.int16, device=q.device)
        s = torch.cuda.current_stream().cuda_stream
        n = min(B, 65536)
        z = np.zeros(3 * 65536, dtype=np.uint64)
        for _ in range(3):
            assert LB.ccmp_project_batch(h, C.byref(P), q.data_ptr(), out.data_ptr(), ok.data_ptr(), it.data_ptr(), B, s) == 0
        torch.cuda.synchronize()
        tr = np.zeros(3 * 65536, dtype=np.uint64)
        assert LB.ccmp_debug_flat_trace(tr.ctypes.data, 65536) == 0
        tr = tr.reshape(65536, 3)
        used = tr[:, 1] > 0
        # stale entries of earlier launches have older time stamps: keep those of the last launch (start within 50 ms of the latest end)
        tmax = tr[used, 1].max()
        used &= tr[:, 0] + 5000000 > tmax
        tr = tr[used]
        t0 = tr[:, 0].min()
        start, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0
        rounds = ((tr[:, 2] >> 16) & 0xffff).astype(np.float64) + 1
        done_before = (tr[:, 2] & 0xffff).astype(np.float64)
        print("B=%d: latency kernel ran %d samples, %.0f rounds, spans %.0f us -> %.0f rounds/us; iterations already done at hand-over: mean %.1f"
              % (B, len(tr), rounds.sum(), end.max(), rounds.sum() / end.max(), done_before.mean()))
        last = np.argsort(-end)[:8]
        for e in last:
            print("   start %7.1f end %7.1f us rounds %4d (%.2f us per round), %d iterations before the hand-over" % (start[e], end[e], rounds[e], (end[e] - start[e]) / rounds[e], done_before[e]))
        T = end.max()
        for lo in np.arange(0, T, T / 8):
            hi = lo + T / 8
            busy = ((start < hi) & (end > lo)).sum()
            print("   %6.0f-%6.0f us: %5d samples in flight, %5d start" % (lo, hi, busy, ((start >= lo) & (start < hi)).sum()))




def lpt_min():
    """from which batch size the scout (order + two-class hand-over) pays"""
    ctx = Context(0)
    for obj in ("Wine_Bottle", "stefan"):
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        for B in (12288, 14336, 16384, 18432, 20480, 22528, 24576, 26624):
            q = c.ambient_uniform_batch(0xC3, 0, B)
            out = torch.empty_like(q)
            row = []
            for name, mn in (("no scout", 1 << 30), ("scout", 0)):
                ctx.set_lpt(1, mn)
                row.append("%s %.3f" % (name, min(timed(lambda: c.project_batch(q, out=out), reps=5) for _ in range(2))))
            ctx.set_schedule(2, 0)
            row.append("latency kernel alone %.3f" % min(timed(lambda: c.project_batch(q, out=out), reps=5) for _ in range(2)))
            ctx.set_schedule(1)
            ctx.set_lpt(1)
            print("%s B=%d  %s" % (obj, B, " | ".join(row)), flush=True)


def pool_classes():
    """hand-over in two classes: threshold on the predicted remaining iterations"""
    ctx = Context(0)
    for obj in ("Wine_Bottle", "stefan"):
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        for B in (28672, 32768, 40960, 49152, 65536, 98304):
            q = c.ambient_uniform_batch(0xC3, 0, B)
            out = torch.empty_like(q)
            ctx.set_option("pool_long_remaining", 0)
            ref = c.project_batch(q)
            row = []
            for thr in (0, 8, 16, 24, 32, 48, 64, 96):
                ctx.set_option("pool_long_remaining", thr)
                ms = min(timed(lambda: c.project_batch(q, out=out), reps=5) for _ in range(2))
                got = c.project_batch(q)
                same = all(torch.equal(a, b) for a, b in zip(got, ref))
                row.append("%d: %.3f%s" % (thr, ms, "" if same else " DIFFERENT"))
            print("%s B=%d  %s" % (obj, B, " | ".join(row)), flush=True)
    ctx.set_option("pool_long_remaining", 24)


if __name__ == "__main__":
    globals()[sys.argv[1]]()
