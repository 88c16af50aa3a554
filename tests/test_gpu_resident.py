"""The opt-in resident service kernel (option "resident"; csrc/ccmp_resident.h): single-state calls without a launch — the same
bits as the launched kernels, interleaved with batched launches and a buffer-growing call on the same context, restarting after
its idle exit, and never able to hang a device-wide synchronise."""
import ctypes as C
import time

import numpy as np
import pytest

from conftest import config_path

pytestmark = pytest.mark.gpu


def _constraint(obj="Wine_Bottle"):
    from closed_chain_motion_planner_amd import Context, KinematicChainConstraint

    ctx = Context(0)
    return KinematicChainConstraint.from_yaml(config_path(obj), ctx=ctx), ctx


def _single_calls(c, xs):
    out = []
    for x in xs:
        y = x.copy()
        ok = c.project(y)
        out.append((y, ok, c.function(x).copy(), c.isSatisfied(x), c.jointValid(x), c.isSatisfied(y), c.jointValid(y)))
    return out


def _bits(a, b):
    """bit for bit — except that a NaN need only be a NaN on both sides (which payload an invalid operation leaves is not part of
    the arithmetic the oracle pins)"""
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint64), b[~nb].view(np.uint64))


def _same(a, b):
    bad = [(k, u, v) for k, (u, v) in enumerate(zip(a, b)) if not (_bits(u[0], v[0]) and u[1] == v[1] and _bits(u[2], v[2]) and u[3:] == v[3:])]
    if bad:
        print("first difference (state %d):\n  launched %r\n  resident %r" % bad[0])
    return not bad


@pytest.mark.parametrize("obj", ["Wine_Bottle", "stefan"])
def test_resident_calls_are_bitwise_the_launched_ones(obj):
    import torch

    c, ctx = _constraint(obj)
    far = c.ambient_uniform_batch(0x4E5, 0, 48).cpu().numpy()
    q, ok, _ = c.project_batch(c.ambient_uniform_batch(0x4E6, 0, 512))
    near = q[ok == 1][:24].cpu().numpy() + np.random.default_rng(5).uniform(-0.05, 0.05, (24, 14))
    xs = np.concatenate([far, near, np.full((1, 14), np.nan)])
    want = _single_calls(c, xs)
    assert ctx.get_option("resident") == 0
    c.setResident(True)
    assert ctx.get_option("resident") == 1
    c.setResident(False)
    assert ctx.get_option("resident") == 0
    ctx.set_option("resident", 1)
    assert ctx.get_option("resident") == 1
    got = _single_calls(c, xs)
    assert _same(want, got)
    # iteration counts too (the C ABI's iters)
    from closed_chain_motion_planner_amd import _lib

    L, dp = _lib.lib(), C.POINTER(C.c_double)
    for x in xs[:8]:
        its = []
        for on in (1, 0):
            ctx.set_option("resident", on)
            y, okb, it = np.zeros(14), (C.c_uint8 * 1)(), (C.c_uint16 * 1)()
            assert L.ccmp_project_host(ctx.handle, C.byref(c.problem), x.ctypes.data_as(dp), y.ctypes.data_as(dp), okb, it, 1) == 0
            its.append((int(it[0]), int(okb[0]), y.tobytes()))
        assert its[0] == its[1]
    ctx.set_option("resident", 1)
    # another problem through the same service: tolerances, then calibrated arms (the general instantiation of the kernel)
    c.setTolerance(5e-4, 2.5e-3)
    got2 = _single_calls(c, xs[:16])
    ctx.set_option("resident", 0)
    assert _same(_single_calls(c, xs[:16]), got2)
    ctx.set_option("resident", 1)
    ctx.set_option("stock_kernels", 0)
    got3 = _single_calls(c, xs[:16])
    ctx.set_option("resident", 0)
    assert _same(_single_calls(c, xs[:16]), got3) and _same(got2, got3)
    torch.cuda.synchronize()


def test_resident_service_interleaved_with_batches_growth_and_synchronisation():
    """what must never hang: batched launches and a workspace-growing call on the SAME context between resident calls (the library
    stops the service before its hipFree), a device-wide synchronise while the service is alive (it leaves by itself after
    resident_idle_ms), and calls after that exit (one launch restarts it)"""
    import torch

    c, ctx = _constraint()
    ctx.set_option("resident_idle_ms", 20)
    ctx.set_option("resident", 1)
    xs = c.ambient_uniform_batch(0x4E7, 0, 8).cpu().numpy()
    ctx.set_option("resident", 0)
    want = _single_calls(c, xs)
    ctx.set_option("resident", 1)
    for B in (4096, 20000, 70000):  # each larger than the last: the scout's buffers, the pool and the staging grow under a live service
        got = _single_calls(c, xs[:3])
        assert _same(want[:3], got)
        q = c.ambient_uniform_batch(0x4E8, 0, B)
        qo, ok, it = c.project_batch(q)
        h = c.project_host(q[:3000].cpu().numpy())  # staged through the context's own stream and staging buffer
        torch.cuda.synchronize()
        assert np.array_equal(h[0].view(np.uint64), qo[:3000].cpu().numpy().view(np.uint64))
    got = _single_calls(c, xs)
    assert _same(want, got)
    # a device-wide synchronise right behind a resident call returns once the service has idled out — bounded, not for ever
    y = xs[0].copy()
    c.project(y)
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    assert waited < 2.0, waited
    # ... and the service is back with the next call
    assert _same(want, _single_calls(c, xs))
    time.sleep(0.1)  # idle exit on its own
    assert _same(want, _single_calls(c, xs))
    ctx.set_option("resident", 0)
    torch.cuda.synchronize()


def test_resident_latency_is_reported():
    """median host-to-host latency of the reference-signature calls, launched against resident (printed; the bench carries the
    same figures as single_project_resident_us / single_is_satisfied_resident_us)"""
    import torch

    c, ctx = _constraint()
    q, ok, _ = c.project_batch(c.ambient_uniform_batch(0x4E6, 0, 1024))
    near = q[ok == 1][:64].cpu().numpy() + np.random.default_rng(7).uniform(-0.05, 0.05, (64, 14))
    far = c.ambient_uniform_batch(0xC1, 0, 64).cpu().numpy()
    torch.cuda.synchronize()
    res = {}
    for on in (0, 1, 0, 1):
        ctx.set_option("resident", on)
        for name, xs, fn in (("project(uniform)", far, c.project), ("project(near)", near, c.project), ("isSatisfied", near, c.isSatisfied),
                             ("function", near, c.function)):
            ts = []
            for x in xs:
                y = x.copy()
                t0 = time.perf_counter()
                fn(y)
                ts.append(time.perf_counter() - t0)
            res.setdefault((name, on), []).append(float(np.median(ts[8:]) * 1e6))
    ctx.set_option("resident", 0)
    for name in ("project(uniform)", "project(near)", "isSatisfied", "function"):
        print("%-18s launched %6.1f us   resident %6.1f us" % (name, min(res[(name, 0)]), min(res[(name, 1)])))
    assert min(res[("isSatisfied", 1)]) < min(res[("isSatisfied", 0)])


def test_two_contexts_with_a_service_each():
    """two constraints, two contexts, both resident, called in turn (a planner with a second constraint object): each service has
    its own low-priority stream; either they get hardware queues of their own — then no call waits for the other's idle exit — or
    a call that finds its kernel queued behind the other's takes the launch path, abandons that kernel and tries again behind a
    back-off ("resident_gave_up" counts the occasions; the option stays on).  Either way: the same bits, and no call anywhere near
    the idle time."""
    import torch

    (ca, xa), (cb, xb) = _constraint(), _constraint("stefan")
    for ctx in (xa, xb):
        ctx.set_option("resident_idle_ms", 200)  # long enough that waiting for the other's idle exit would show
    xs = ca.ambient_uniform_batch(0x4E9, 0, 12).cpu().numpy()
    want_a, want_b = _single_calls(ca, xs), _single_calls(cb, xs)
    xa.set_option("resident", 1)
    xb.set_option("resident", 1)
    worst = 0.0
    for k in range(6):
        for c, want in ((ca, want_a), (cb, want_b)):
            t0 = time.perf_counter()
            got = _single_calls(c, xs[2 * k: 2 * k + 2])
            worst = max(worst, time.perf_counter() - t0)
            assert _same(want[2 * k: 2 * k + 2], got)
    print("two services: worst batch of 14 single-state calls %.2f ms; gave up: %d %d" % (worst * 1e3, xa.get_option("resident_gave_up"), xb.get_option("resident_gave_up")))
    assert worst < 0.05, worst  # 14 calls of ~0.1 ms; a wait for the other service's idle exit would be 200 ms
    assert xa.get_option("resident") == 1 and xb.get_option("resident") == 1  # giving a start up is for one call, not for the context's life
    xa.set_option("resident", 1)
    assert xa.get_option("resident_gave_up") == 0  # ... and asking again clears the count and the back-off
    xa.set_option("resident", 0)
    xb.set_option("resident", 0)
    torch.cuda.synchronize()


def test_single_edges_through_the_service_are_bitwise_the_launched_ones():
    """discreteGeodesic / checkMotion of ONE pair — how the unchanged planner asks for them (src/planner/stefanBiPRM.cpp:315-318,
    397-398; the adapter's ccmp_geodesic_host_ex with E == 1) — through the resident service kernel: the per-edge body of
    geodesic_flat_kernel, included into it.  States, counts, flags and carries bit for bit the launched kernel's: arriving edges,
    edges that give up, a target that fails isSatisfied, a list too short (n = max_states + 1), a round budget spent (ok = 2)."""
    import torch

    from closed_chain_motion_planner_amd import _lib

    c, ctx = _constraint()
    L, dp = _lib.lib(), C.POINTER(C.c_double)
    q, ok, _ = c.project_batch(c.ambient_uniform_batch(0x4EA, 0, 2048))
    good = q[ok == 1].cpu().numpy()
    frm = good[:24]
    to = np.array([c.sample_near_project_batch(0x4EB, 0, torch.as_tensor(frm).cuda(), 0.6, len(frm), want_iters=False)[0].cpu().numpy()][0])
    to[3] = good[100]           # far apart: many states or a give-up
    to[4] = to[4] + 0.3         # off the manifold: checkMotion's isSatisfied(to) fails
    torch.cuda.synchronize()
    cases = [(64, 0, 0), (64, 0, 1), (3, 0, 0), (16, 8, 0)]  # (max_states, round_budget, check_target)

    def run(on):
        ctx.set_option("resident", on)
        out = []
        for ms, budget, chk in cases:
            for a, b in zip(frm, to):
                st, n, okb, carry = np.full((ms, 14), 7.0), (C.c_int32 * 1)(), (C.c_uint8 * 1)(), np.zeros(2)
                rc = L.ccmp_geodesic_host_ex(ctx.handle, C.byref(c.problem), a.ctypes.data_as(dp), b.ctypes.data_as(dp), 1, ms, st.ctypes.data_as(dp), n, okb, None,
                                             carry.ctypes.data_as(dp), budget, chk)
                assert rc == 0
                rows = min(int(n[0]), ms)
                out.append((int(n[0]), int(okb[0]), st[:rows].tobytes(), carry.tobytes()))
        return out

    want = run(0)
    got = run(1)
    ctx.set_option("resident", 0)
    assert [w[:2] for w in want] == [g[:2] for g in got]
    assert want == got
    flags = {w[1] for w in want}
    assert {0, 1, 2} <= flags and any(w[0] == 4 for w in want[2 * len(frm): 3 * len(frm)])  # every kind of ending was in the sample
    # latency of one pair, launched against resident
    res = {}
    for on in (0, 1, 0, 1):
        ctx.set_option("resident", on)
        ts = []
        st, n, okb, carry = np.zeros((64, 14)), (C.c_int32 * 1)(), (C.c_uint8 * 1)(), np.zeros(2)
        for a, b in list(zip(frm, to)) * 3:
            t0 = time.perf_counter()
            L.ccmp_geodesic_host_ex(ctx.handle, C.byref(c.problem), a.ctypes.data_as(dp), b.ctypes.data_as(dp), 1, 64, st.ctypes.data_as(dp), n, okb, None,
                                    carry.ctypes.data_as(dp), 0, 1)
            ts.append(time.perf_counter() - t0)
        res.setdefault(on, []).append(float(np.median(ts[8:]) * 1e6))
    ctx.set_option("resident", 0)
    print("checkMotion of one pair: launched %.1f us   resident %.1f us" % (min(res[0]), min(res[1])))
    assert min(res[1]) < min(res[0])
    torch.cuda.synchronize()
