#!/usr/bin/env python3
"""One parametrised timing harness for the GPU box (development aid; bench.py is the contract).

    python tools/measure.py sizes   [obj] [--schedules]    run time against batch size (1 .. 1048576), default policy;
                                                           --schedules adds latency-kernel-only and throughput-only columns
    python tools/measure.py single                          latency of the reference-signature single-state calls
    python tools/measure.py geodesic [E ...]                batched discreteGeodesic (near-neighbour edges)
    python tools/measure.py analytic                        analytic mode against batch size and waves per CU
    python tools/measure.py host                            PCIe-inclusive rate of ccmp_project_host (pageable / pinned)
    python tools/measure.py sharded [n_gpus [B]]            one process, n GPUs, RCCL all-gather inside the C ABI: per-GPU stream times
    python tools/measure.py sampler                         project_batch vs the fused sampler
    python tools/measure.py soak                            25 repeats of the default policy, outputs compared bit for bit
    python tools/measure.py soak_resident [N]               N states as single calls through the resident service kernel against the batched kernels
    python tools/measure.py scout                           FP32 scout's predictions against the true iteration counts
    python tools/measure.py run <workload> [reps]           a fixed workload for rocprofv3 (tools/profile.sh):
                                                           c3 | flat4096 | mid<B> | flat1 | geodesic | analytic[<B>] | stefan | stefan_tight | calibrated | clearance
"""
import ctypes as C
import sys
import time

import numpy as np
import torch

import os
if len(sys.argv) > 1 and sys.argv[1] == "scout":
    os.environ.setdefault("CCMP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "closed_chain_motion_planner_amd", "lib", "libccmp_debug.so"))  # reads the scout's predictions: a hook of include/ccmp_debug.h
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402

CFG = "tests/golden/config/%s.yaml"


def timed(fn, reps=3):
    """best of `reps`, HIP events on the launch stream"""
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def near_edges(c, E, seed=0x6E0, dist=0.6):
    """growTree-shaped edges: a valid projected state -> a projected sampleUniformNear state within `dist` per joint"""
    q, ok, _, _ = c.sample_project_batch(seed, 0, 8 * E, want_iters=False)
    frm = q[ok == 1][:E].contiguous()
    to, _, _, _ = c.sample_near_project_batch(seed + 1, 0, frm, dist, E, want_iters=False)
    return frm, to


def sizes(argv):
    obj = next((a for a in argv if not a.startswith("-")), "Wine_Bottle")
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
    variants = [("default", None)]
    if "--schedules" in argv:
        variants += [("latency-only", (2, 0)), ("throughput-only", (0, 0))]
    for B in (1, 64, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
        q = c.ambient_uniform_batch(0xC2 if B <= 16384 else 0xC3, 0, B)
        out = torch.empty_like(q)
        res = []
        for name, sched in variants:
            if sched is None:
                ctx.set_schedule(1)
            else:
                if name == "latency-only" and B > 65536:
                    continue
                ctx.set_schedule(*sched)
            ms = timed(lambda: c.project_batch(q, out=out), reps=3 if B > 65536 else 5)
            res.append("%s %8.3f ms (%.3e/s)" % (name, ms, B / ms * 1e3))
        ctx.set_schedule(1)
        print("%s B=%-8d " % (obj, B) + "   ".join(res), flush=True)


def single(argv):
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    q = c.ambient_uniform_batch(1, 0, 64).cpu().numpy()
    for name, fn in (("project", c.project), ("function", c.function), ("isSatisfied", c.isSatisfied), ("jointValid", c.jointValid)):
        ts = []
        for i in range(64):
            x = q[i].copy()
            t0 = time.perf_counter()
            fn(x)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts[4:]) * 1e6
        print("%-12s median %.1f us  p90 %.1f us  min %.1f us" % (name, np.median(ts), np.percentile(ts, 90), ts.min()))
    x0 = np.array(c.problem.start_joint[:])
    ts = []
    for i in range(40):  # near-manifold projections (the geodesic regime): few Newton iterations
        x = x0 + 0.02 * np.sin(np.arange(14) + i)
        t0 = time.perf_counter()
        c.project(x)
        ts.append(time.perf_counter() - t0)
    print("project near the manifold: median %.1f us" % (np.median(np.array(ts[4:])) * 1e6))


def geodesic(argv):
    """batched extend step: the call with lists of 16 and of 64 states, the edges that did not fit, and the complete
    operation (continuation of those edges until every list is whole)"""
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    for E in [int(a) for a in argv] or [5, 64, 1024, 16384]:
        frm, to = near_edges(c, E)
        row = []
        for cap in (16, 64):
            budget = 128 if cap == 16 else 0  # bench.py's first pass: lists of 16 states, 128 Newton rounds per edge
            ms = timed(lambda: c.discrete_geodesic_batch(frm, to, cap, want_carry=True, round_budget=budget), reps=3)
            st, n, okg, its, carry = c.discrete_geodesic_batch(frm, to, cap, want_carry=True, round_budget=budget)
            over = int(((n > cap) | (okg == 2)).sum())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = c.discrete_geodesic_batch(frm, to, cap, want_carry=True, round_budget=budget)
            whole = c.continue_geodesics(to, r[0], r[1], r[2], r[3], r[4], cap, round_budget=budget)
            torch.cuda.synchronize()
            ms_all = (time.perf_counter() - t0) * 1e3
            row.append("lists of %d: %.3f ms, %d edges did not fit, %.3e complete edges/s; everything continued to the end: %.2f ms"
                       % (cap, ms, over, (E - over) / ms * 1e3, ms_all))
        print("E=%-6d %s | mean states %.2f (first %d), reached %.3f, Newton iterations per edge %.1f"
              % (E, "; ".join(row), n.clamp(max=cap).float().mean().item(), cap, (okg == 1).float().mean().item(), its.float().mean().item()), flush=True)


_ANALYTIC = ("analytic_small_batch", "analytic_waves_per_cu", "analytic_handover")


def analytic(argv):
    """analytic mode: run time against batch size — the latency kernel alone ('lat'), the lane-pair kernel alone (h0) and with
    its hand-over to the latency kernel at <= h samples per wavefront, for w wavefronts per CU; argv: sizes"""
    ctx = Context(0)
    sizes_ = [int(a) for a in argv] or [1024, 4096, 8192, 16384, 32768, 65536, 262144, 2097152]
    shapes = [(12, 0), (12, 4), (12, 8), (12, 12), (12, 16), (12, 24), (8, 8), (10, 8)]
    for obj in ("Wine_Bottle", "stefan"):
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        c.setJacobianMode(1)
        for B in sizes_:
            q = c.ambient_uniform_batch(0xC3, 0, B)
            out = torch.empty_like(q)
            res = []
            if B <= 65536:
                ctx.set_option("analytic_small_batch", 1 << 40)
                res.append("lat %.3f" % timed(lambda: c.project_batch(q, out=out), reps=7))
            ctx.set_option("analytic_small_batch", 0)
            for w, h in shapes:
                ctx.set_option("analytic_waves_per_cu", w)
                ctx.set_option("analytic_handover", h)
                res.append("w%d h%d %.3f" % (w, h, timed(lambda: c.project_batch(q, out=out), reps=7)))
            for name in _ANALYTIC:
                ctx.set_option(name, _lib.get_option(None, name))
            ms = timed(lambda: c.project_batch(q, out=out), reps=7)
            print("analytic %s B=%-8d DEFAULT %7.3f ms (%.3e/s) | " % (obj, B, ms, B / ms * 1e3) + "  ".join(res), flush=True)


def host(argv):
    ctx = Context(0)
    L = _lib.lib()
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    B = 262144
    q = c.ambient_uniform_batch(0xC3, 0, B).cpu()
    for name, pin in (("pageable", False), ("pinned", True)):
        mk = (lambda t: t.pin_memory()) if pin else (lambda t: t)
        qi, qo = mk(q.clone()), mk(torch.empty_like(q))
        ok, it = mk(torch.empty(B, dtype=torch.uint8)), mk(torch.empty(B, dtype=torch.int16))
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            rc = L.ccmp_project_host(ctx.handle, C.byref(c.problem), C.cast(qi.data_ptr(), C.POINTER(C.c_double)),
                                     C.cast(qo.data_ptr(), C.POINTER(C.c_double)), C.cast(ok.data_ptr(), C.POINTER(C.c_uint8)),
                                     C.cast(it.data_ptr(), C.POINTER(C.c_uint16)), B)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        ms = np.median(ts[1:]) * 1e3
        print("%-9s host buffers: %.2f ms per 262144 -> %.2e projections/s (device-resident: bench.py)" % (name, ms, B / ms * 1e3))


def sharded(argv):
    """one process, every visible GPU (ccmp_comm_* + ccmp_sample_project_sharded: the reference's shape, RCCL inside the C
    ABI): wall time per call, per-GPU stream time of the shard's kernels and of the all-gather behind them, and the
    identity of the gathered valid states with those of the same samples on GPU 0 alone"""
    from closed_chain_motion_planner_amd import Communicator

    n = int(argv[0]) if argv else torch.cuda.device_count()
    B = int(argv[1]) if len(argv) > 1 else 262144 * n
    ctxs = [Context(g) for g in range(n)]
    comm = Communicator(ctxs)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctxs[0])
    c.sample_project_sharded(0xC5, 0, B, comm, want_full=False)  # communicator and workspaces
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        valid, counts, _ = c.sample_project_sharded(0xC5, 0, B, comm, want_full=False)
        ts.append(time.perf_counter() - t0)
    k, g = comm.last_timing()
    print("%d GPUs, %d samples: %.2f ms per call (host wall, upload-free sampler form) -> %.3e projections/s" % (n, B, min(ts) * 1e3, B / min(ts)))
    print("  per-GPU kernel ms:", " ".join("%.2f" % v for v in k), "| all-gather ms behind them:", " ".join("%.3f" % v for v in g))
    print("  valid states per GPU:", counts)
    probe = 4096
    vp, cp, _ = c.sample_project_sharded(0xC5, 0, probe, comm, want_full=False)
    q, ok, _, _ = c.sample_project_batch(0xC5, 0, probe, want_iters=False)
    alone = q[ok == 1].cpu().numpy()
    print("  4096-sample probe: gathered == one GPU alone, bit for bit: %s (%d valid)"
          % (bool(alone.shape == np.asarray(vp).shape and np.array_equal(alone.view(np.uint64), np.asarray(vp).view(np.uint64))), len(alone)))
    comm.close()


def afterload(argv):
    """Does a latency-bound batch (4 096 / 32 768 samples: the serial chain of its longest sample) run slower right behind a
    saturated launch?  bench.py times its secondaries behind 20 C3 steps (0.33 s of FP64 at the power limit: the chip holds
    2.27 GHz there, 2.4 idle).  Mean of 10 launches back to back: cold, immediately behind 20 x C3, and after pauses."""
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    big = c.ambient_uniform_batch(0xC3, 0, 262144)
    bout = torch.empty_like(big)
    qs = {B: c.ambient_uniform_batch(0xC2 if B == 4096 else 0xC3, 0, B) for B in (4096, 32768)}
    outs = {B: torch.empty_like(q) for B, q in qs.items()}

    def mean10(B):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c.project_batch(qs[B], out=outs[B])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10

    def load():
        for _ in range(20):
            c.project_batch(big, out=bout)
        torch.cuda.synchronize()

    for B in qs:
        c.project_batch(qs[B], out=outs[B])
    torch.cuda.synchronize()
    for B in qs:
        row = ["cold %.3f %.3f" % (mean10(B), mean10(B))]
        for pause in (0.0, 0.05, 0.25, 1.0, 3.0):
            load()
            time.sleep(pause)
            row.append("load+%.2fs %.3f %.3f" % (pause, mean10(B), mean10(B)))
        print("B=%6d  mean-of-10 ms  %s" % (B, " | ".join(row)), flush=True)


def sampler(argv):
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    B = 262144
    q = c.ambient_uniform_batch(0xC3, 0, B)
    out = torch.empty_like(q)
    a = timed(lambda: c.project_batch(q, out=out), reps=4)
    b = timed(lambda: c.sample_project_batch(0xC3, 0, B), reps=4)
    print("project_batch %.3f ms   sample_project_batch (fused sampler + wrap) %.3f ms" % (a, b))


def soak(argv):
    ctx = Context(0)
    for obj, B in (("Wine_Bottle", 262144), ("stefan", 98304), ("dumbbell", 40000), ("Wine_Bottle", 20000)):
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        q = c.ambient_uniform_batch(0x50A, 0, B)
        ref = c.project_batch(q)
        bad = sum(not all(torch.equal(a, b) for a, b in zip(c.project_batch(q), ref)) for _ in range(25))
        print(obj, B, "repeats differing from the first run:", bad, flush=True)
    # the analytic mode (lane-pair kernel + hand-over to the latency kernel) and the proxy clearance, the same way
    from closed_chain_motion_planner_amd.scene import ProxyValidityChecker

    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    c.setJacobianMode(1)
    q = c.ambient_uniform_batch(0x50B, 0, 200000)
    ref = c.project_batch(q)
    bad = sum(not all(torch.equal(a, b) for a, b in zip(c.project_batch(q), ref)) for _ in range(25))
    print("Wine_Bottle analytic 200000 (lane-pair kernel + latency kernel) repeats differing:", bad, flush=True)
    for obj, B in (("stefan", 262144), ("Wine_Bottle", 2097152), ("dumbbell", 5000)):  # the hand-over's order is decided by atomics; the latency kernel alone
        ca = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        ca.setJacobianMode(1)
        qa = ca.ambient_uniform_batch(0x50E, 0, B)
        ra = ca.project_batch(qa)
        bad = sum(not all(torch.equal(a, b) for a, b in zip(ca.project_batch(qa), ra)) for _ in range(25))
        print(obj, "analytic", B, "repeats differing:", bad, flush=True)
    # the extend step beyond the resident blocks: ticket queue + FP32 scout order (atomics decide who takes which edge,
    # the sort's ties fall as they fall) — counts, flags, Newton counts and every listed state, the same 25 times over
    cg = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    frm, to = near_edges(cg, 16384)
    rg = cg.discrete_geodesic_batch(frm, to, 16)
    live = torch.arange(16, device=frm.device)[None, :] < rg[1].clamp(max=16)[:, None]
    bad = 0
    for _ in range(25):
        g = cg.discrete_geodesic_batch(frm, to, 16)
        bad += not (all(torch.equal(a, b) for a, b in zip(g[1:], rg[1:])) and torch.equal(g[0][live], rg[0][live]))
    print("Wine_Bottle extend step, 16384 edges, scout order: repeats differing:", bad, flush=True)
    sc = ProxyValidityChecker(c).scene
    refc = sc.clearance_batch(ref[0])
    bad = sum(not all(torch.equal(a, b) for a, b in zip(sc.clearance_batch(ref[0]), refc)) for _ in range(25))
    print("proxy clearance 200000 repeats differing:", bad, flush=True)


def soak_resident(argv):
    """the resident service kernel under a long run: N distinct states, each through project / function / isSatisfied / jointValid as
    single calls on the service, against the BATCHED kernels' results on the same states (both are bit-identical to the oracle, so
    they must be to each other) — with idle exits (the service leaves after 2 ms here and restarts), batch launches and workspace
    growth on the same context in between, and every 64th state a single checkMotion edge towards the next one"""
    import time

    N = int(argv[0]) if argv else 20000
    ctx = Context(0)
    for obj in ("Wine_Bottle", "stefan"):
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        far = c.ambient_uniform_batch(0x50C, 0, N)
        qp, okp, _ = c.project_batch(c.ambient_uniform_batch(0x50D, 0, 4 * N))
        near = qp[okp == 1][: N // 2] + 0.04 * (torch.rand((min(N // 2, int((okp == 1).sum())), 14), dtype=torch.float64, device=qp.device) - 0.5)
        xs = torch.cat([far, near])[torch.randperm(N + near.shape[0], device=far.device)]
        qb, okb, itb = c.project_batch(xs)
        fb, sb, jb = c.function_batch(xs), c.is_satisfied_batch(xs), c.joint_valid_batch(xs)
        torch.cuda.synchronize()
        xs_h, qb_h, okb_h = xs.cpu().numpy(), qb.cpu().numpy(), okb.cpu().numpy()
        fb_h, sb_h, jb_h = fb.cpu().numpy(), sb.cpu().numpy(), jb.cpu().numpy()
        # single checkMotion edges, launched (the service off) for the expected values
        edges = list(range(0, xs_h.shape[0] - 1, 64))
        frm_e, to_e = qb[edges], qb[[e + 1 for e in edges]]
        exp_e = c.discrete_geodesic_batch(frm_e, to_e, 16, check_target=True)
        torch.cuda.synchronize()
        frm_h, to_h = frm_e.cpu().numpy(), to_e.cpu().numpy()
        exp_st, exp_n, exp_ok = exp_e[0].cpu().numpy(), exp_e[1].cpu().numpy(), exp_e[2].cpu().numpy()
        L, dp = _lib.lib(), C.POINTER(C.c_double)
        ctx.set_option("resident_idle_ms", 2)
        ctx.set_option("resident", 1)
        bad = restarts = 0
        grow = 30000
        t0 = time.perf_counter()
        for k in range(xs_h.shape[0]):
            x = xs_h[k]
            y = x.copy()
            ok = c.project(y)
            f = c.function(x)
            same = (np.array_equal(np.isnan(y), np.isnan(qb_h[k])) and np.array_equal(y[~np.isnan(y)].view(np.uint64), qb_h[k][~np.isnan(y)].view(np.uint64))
                    and bool(ok) == bool(okb_h[k]) and np.array_equal(np.isnan(f), np.isnan(fb_h[k])) and np.array_equal(f[~np.isnan(f)].view(np.uint64), fb_h[k][~np.isnan(f)].view(np.uint64))
                    and bool(c.isSatisfied(x)) == bool(sb_h[k]) and bool(c.jointValid(x)) == bool(jb_h[k]))
            if k % 64 == 0 and k // 64 < len(edges):
                e = k // 64
                # one edge from host buffers, as the unchanged planner asks for it: through the service
                st, n1, ok1 = np.zeros((16, 14)), (C.c_int32 * 1)(), (C.c_uint8 * 1)()
                rc = L.ccmp_geodesic_host_ex(ctx.handle, C.byref(c.problem), frm_h[e].ctypes.data_as(dp), to_h[e].ctypes.data_as(dp), 1, 16,
                                             st.ctypes.data_as(dp), n1, ok1, None, None, 0, 1)
                n = min(int(exp_n[e]), 16)
                same = same and rc == 0 and int(n1[0]) == int(exp_n[e]) and int(ok1[0]) == int(exp_ok[e]) and st[:n].tobytes() == exp_st[e, :n].tobytes()
            bad += not same
            if k % 997 == 0:
                time.sleep(0.004)  # idle exit: the next call starts the service again
                restarts += 1
            if k % 4001 == 0:  # a batch and a growing workspace on the same context under the live service
                grow += 7000
                c.project_batch(c.ambient_uniform_batch(0x50E, k, grow))
        dt = time.perf_counter() - t0
        gave_up = ctx.get_option("resident_gave_up")
        ctx.set_option("resident", 0)
        print("%s: %d states x (project, function, isSatisfied, jointValid) + %d single checkMotion edges through the resident service, %d idle exits, "
              "%d batch calls in between: differing from the batched kernels %d; gave up %d; %.1f s" %
              (obj, xs_h.shape[0], len(edges), restarts, xs_h.shape[0] // 4001 + 1, bad, gave_up, dt), flush=True)
        assert bad == 0 and gave_up == 0


def scout(argv):
    ctx = Context(0)
    L = _lib.lib()
    L.ccmp_ctx_debug_lpt_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for obj in ("Wine_Bottle", "stefan"):
        c = KinematicChainConstraint.from_yaml(CFG % obj, ctx=ctx)
        B = 262144
        q = c.ambient_uniform_batch(0xC3, 0, B)
        ctx.set_lpt(1, 0)
        _, _, it = c.project_batch(q)
        pred = np.zeros(B, dtype=np.uint16)
        assert L.ccmp_ctx_debug_lpt_pred(ctx.handle, pred.ctypes.data, B) == 0
        it, p = it.cpu().numpy().astype(np.int32), pred.astype(np.int32)
        big = it > 80
        print("%s scout vs FD iterations: equal %.3f  |d|<=2 %.3f  corr %.4f; of the %d samples with > 80 iterations the scout says > 60 for %.3f"
              % (obj, (p == it).mean(), (np.abs(p - it) <= 2).mean(), np.corrcoef(p, it)[0, 1], big.sum(), (p[big] > 60).mean()))


def clearance(argv):
    """proxy clearance: run time against batch size for both kernels (one block per state / 64-state tiles) + single-state call"""
    from closed_chain_motion_planner_amd.scene import ProxyValidityChecker

    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % (argv[0] if argv else "Wine_Bottle"), ctx=ctx)
    sc = ProxyValidityChecker(c).scene
    print("spheres %d pairs %d" % (len(sc.spheres), sc.num_pairs))
    for B in (1, 64, 1024, 4096, 8192, 16384, 32768, 65536, 262144, 1048576):
        q = c.ambient_uniform_batch(0xC3, 0, B)
        row = []
        for name, lim in (("per_state", 1 << 30), ("tiles", 0)):
            if name == "per_state" and B > 65536:
                row.append(float("nan"))
                continue
            ctx.set_option("clearance_per_state_max", lim)
            row.append(timed(lambda: sc.clearance_batch(q, 0.0), 10))
        print("B %8d  per-state %8.4f ms   tiles %8.4f ms" % (B, row[0], row[1]))
    ctx.set_option("clearance_per_state_max", 8192)
    x = c.ambient_uniform_batch(0xC1, 0, 8).cpu().numpy()
    ts = []
    for i in range(200):
        t0 = time.perf_counter()
        sc.clearance(x[i % 8])
        ts.append(time.perf_counter() - t0)
    print("single-state host call: median %.1f us" % (np.median(ts[20:]) * 1e6))


def run(argv):
    """fixed workloads for the profiler: one kernel family each, `reps` launches"""
    what = argv[0]
    reps = int(argv[1]) if len(argv) > 1 else 12
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % ("stefan" if what.startswith("stefan") else "Wine_Bottle"), ctx=ctx)
    if what == "stefan_tight":  # BASELINE configs[3] read as "tighter loop tol": half the reference's tolerances
        c.setTolerance(5e-4, 2.5e-3)
    if what == "calibrated":  # C3 on calibrated arms (PandaModel::initModel(dh), offsets differing per arm): the general instantiations
        for arm in (0, 1):
            dh = (C.c_double * 28)(*[(1e-3 if arm == 0 else -7e-4) * ((5 * i + 3 * arm) % 7 - 3) for i in range(28)])
            assert _lib.lib().ccmp_set_calibration(C.byref(c.problem), arm, dh) == 0
    if what in ("c3", "stefan", "stefan_tight", "calibrated"):
        q = c.ambient_uniform_batch(0xC3, 0, 262144)
        fn = lambda: c.project_batch(q)
    elif what.startswith("flat") and what[4:].isdigit() and int(what[4:]) > 1:  # flat4096 (C2), flat14336 ...: the latency kernel alone
        q = c.ambient_uniform_batch(0xC2, 0, int(what[4:]))
        fn = lambda: c.project_batch(q)
    elif what.startswith("mid"):  # mid32768, mid65536 ...: scout + throughput kernel + hand-over to the latency kernel
        q = c.ambient_uniform_batch(0xC3, 0, int(what[3:]))
        fn = lambda: c.project_batch(q)
    elif what == "flat1":
        x = c.ambient_uniform_batch(0xC1, 0, 64).cpu().numpy()
        fn = lambda: [c.project(x[i].copy()) for i in range(64)]
    elif what.startswith("geodesic"):  # geodesic (16384 edges: bench.py's first pass), geodesic65536 (a bulk call) ...
        frm, to = near_edges(c, int(what[8:] or 16384))
        fn = lambda: c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)  # bench.py's first pass
    elif what.startswith("analytic"):  # analytic (C3's batch), analytic4096, analytic2097152 ...
        c.setJacobianMode(1)
        q = c.ambient_uniform_batch(0xC3, 0, int(what[8:] or 262144))
        fn = lambda: c.project_batch(q)
    elif what == "clearance":
        from closed_chain_motion_planner_amd.scene import ProxyValidityChecker

        sc = ProxyValidityChecker(c).scene
        q = c.ambient_uniform_batch(0xC3, 0, 262144)
        fn = lambda: sc.clearance_batch(q, 0.0)
    else:
        raise SystemExit(__doc__)
    import os

    for kv in filter(None, os.environ.get("CCMP_OPTS", "").split(",")):  # CCMP_OPTS=name=value,...: options of the context for this run
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    # >= 3 warm-up calls (the profile's summary drops their dispatches), then `reps` calls, each bracketed by HIP events on the
    # launch stream (a call that forks to the context's side stream joins it back before it returns, so the second event is behind
    # everything) and followed by a synchronise — the idle gap that tells two calls apart in the kernel trace.  The per-call times
    # are printed as JSON: under `--kernel-trace` alone they are the call's wall time with its kernels overlapping as they do in
    # production; under `--pmc` they show what counter collection does to that overlap.
    import json

    warm = int(argv[2]) if len(argv) > 2 else 3
    for _ in range(warm):
        fn()
        torch.cuda.synchronize()
    calls = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        calls.append(e0.elapsed_time(e1))
    print("CALL_MS " + json.dumps({"workload": what, "warmup_calls": warm, "call_ms": calls}))


if __name__ == "__main__":
    cmds = {f.__name__: f for f in (sizes, single, geodesic, analytic, host, sharded, afterload, sampler, soak, soak_resident, scout, clearance, run)}
    if len(sys.argv) < 2 or sys.argv[1] not in cmds:
        raise SystemExit(__doc__)
    cmds[sys.argv[1]](sys.argv[2:])
