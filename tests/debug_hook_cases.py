"""The GPU tests that need a hook of include/ccmp_debug.h — the device probe of ccmp_detmath.h, fault injection.  NOT collected by
the suite directly (the default library exports no such symbol): tests/test_gpu_debug_hooks.py runs this file once, in a process
of its own, with CCMP_LIBRARY = lib/libccmp_debug.so (the same objects, ccmp_api.cpp / ccmp_policy.cpp built with
-DCCMP_DEBUG_HOOKS, plus the probe kernel)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import NCPU  # noqa: F401
from test_gpu_parity import _constraint

from closed_chain_motion_planner_amd import _lib

pytestmark = pytest.mark.gpu


def test_this_process_runs_on_the_debug_library():
    assert os.path.basename(_lib.LIBPATH) == "libccmp_debug.so" and all(hasattr(_lib.lib(), n) for n in _lib.DEBUG_EXPORTS)
    assert "fail_after_fork" in [o["name"] for o in _lib.option_table()]


def test_device_arithmetic_is_bitwise_host(gpu_ctx, oracle_det):
    """sincos / atan2 / sqrt / divide on gfx950 == the same source on the host, bit for bit."""
    import torch
    from closed_chain_motion_planner_amd import _lib

    rng = np.random.default_rng(7)
    n = 200000
    x = np.concatenate([rng.uniform(-8, 8, n // 2), rng.standard_normal(n // 4) * 1e3, rng.uniform(-1e-3, 1e-3, n // 4)])
    y = np.concatenate([rng.uniform(-8, 8, n // 2), rng.standard_normal(n // 4), rng.uniform(-1e3, 1e3, n // 4)])
    x[:6] = [0.0, -0.0, np.pi / 2, 1e6, 2e6, np.inf]
    y[:6] = [0.0, 1.0, 0.0, 3.0, 1.0, 1.0]
    # square roots over the whole exponent range: 2^-766 .. 2^1023 goes through the wave-uniform fast path of
    # ccmp_sqrt (no argument of the wavefront below 2^-767), the block after it — down to subnormals — through the
    # compiler's scaled expansion; both must be the correctly rounded root
    k = 4096
    x[1024:1024 + k] = np.ldexp(rng.uniform(1, 2, k), rng.integers(-766, 1023, k))
    x[1024 + k:1024 + 2 * k] = np.ldexp(rng.uniform(1, 2, k), rng.integers(-1074, -766, k))
    x[1024 + 2 * k:1024 + 2 * k + 3] = [2.0 ** -767, np.nextafter(2.0 ** -767, 0), 5e-324]
    xd, yd = torch.as_tensor(x).cuda(), torch.as_tensor(y).cuda()
    out = torch.empty((n, 5), dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().ccmp_detmath_probe(gpu_ctx.handle, xd.data_ptr(), yd.data_ptr(), out.data_ptr(), n, None),
               "probe")
    torch.cuda.synchronize()
    import torch as _t
    _lib.lib()  # keep loaded
    got = out.cpu().numpy()
    L = oracle_det.lib
    exp = np.empty_like(got)
    s, c = C.c_double(), C.c_double()
    for i in range(n):
        L.orc_sincos(x[i], C.byref(s), C.byref(c))
        exp[i, 0], exp[i, 1] = s.value, c.value
        with np.errstate(all="ignore"):
            exp[i, 2] = L.orc_atan2_nn(abs(x[i]), abs(y[i]))
    exp[:, 3] = np.sqrt(np.abs(x))
    with np.errstate(all="ignore"):
        exp[:, 4] = x / y
    same = (got.view(np.uint64) == exp.view(np.uint64)) | (np.isnan(got) & np.isnan(exp))
    assert same.all(), "device/host arithmetic differs at %s" % np.argwhere(~same)[:5]
    # and against an implementation that shares no source with the product (VERDICT r2, weak #6b: the det oracle compiles
    # the product's ccmp_detmath.h, so the comparison above cannot see a wrong ccmp_sincos): the DEVICE's sines, cosines
    # and arc tangents against glibc's, in units of the last place, over the range the reduction is exact for
    fin = np.isfinite(x) & (np.abs(x) <= 1e5)  # the range tests/test_detmath.py establishes for the host build
    with np.errstate(all="ignore"):
        ref = np.stack([np.sin(x), np.cos(x), np.arctan2(np.abs(x), np.abs(y))], axis=1)
    ulp = np.abs(got[:, :3] - ref) / np.spacing(np.abs(ref))
    ok3 = fin & np.isfinite(y) & ~((x == 0) & (y == 0))
    assert ulp[fin, 0].max() <= 1.0 and ulp[fin, 1].max() <= 1.0 and ulp[ok3, 2].max() <= 2.0, ulp[fin].max(axis=0)



@pytest.mark.parametrize("where", [1, 2])
def test_a_failure_behind_the_fork_still_joins_the_side_stream(gpu_ctx, where):
    """VERDICT r4 #8 / ADVICE r4: a HIP failure after work had been queued on the context's side stream returned at once — the
    caller's stream was never ordered behind the side stream, whose kernels went on writing the caller's buffers.  With the debug
    option "fail_after_fork" the projector's split launch and the extend step's bulk form report a failure in front of (1) /
    behind (2) their side-stream part: the call returns CCMP_EHIP, and once the CALLER'S stream is idle the side stream is too."""
    import torch

    from closed_chain_motion_planner_amd import CcmpError

    c = _constraint("Wine_Bottle", gpu_ctx)
    q = c.ambient_uniform_batch(0x77, 0, 20000)
    qs, oks, _, _ = c.sample_project_batch(0x6F5, 0, 8 * 16384, want_iters=False)  # growTree-shaped edges, as the extend-step tests make them
    frm = qs[oks == 1][:16384].contiguous()
    to, _, _, _ = c.sample_near_project_batch(0x6F6, 0, frm, 0.6, 16384, want_iters=False)
    c.project_batch(q)  # (workspaces grown, nothing pending)
    c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)
    torch.cuda.synchronize()
    gpu_ctx.set_option("fail_after_fork", where)
    try:
        for call in (lambda: c.project_batch(q), lambda: c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)):
            with pytest.raises(CcmpError) as e:
                call()
            assert e.value.code == -2 and "fail_after_fork" in str(e.value)
            torch.cuda.current_stream().synchronize()  # the caller's stream ONLY
            assert gpu_ctx.get_option("side_stream_busy") == 0
    finally:
        gpu_ctx.set_option("fail_after_fork", 0)
    # and the context is usable again
    a = c.project_batch(q)
    gpu_ctx.set_option("fd_split", 0)
    b = c.project_batch(q)
    gpu_ctx.set_option("fd_split", 1)
    torch.cuda.synchronize()
    assert all(torch.equal(u, v) for u, v in zip(a, b))




def test_fault_injection_reaches_every_compute_entry(gpu_ctx):
    """ccmp_debug_fail_calls(ctx, n): the next n compute entry points return CCMP_EHIP before anything is launched — what the
    adapter's exception-safety test (tests/test_cpp_adapter.py) injects from a second thread"""
    from closed_chain_motion_planner_amd import CcmpError

    c = _constraint("Wine_Bottle", gpu_ctx)
    q = c.ambient_uniform_batch(0x77, 0, 64)
    ref = c.project_batch(q)
    assert _lib.lib().ccmp_debug_fail_calls(gpu_ctx.handle, 2) == 0
    for _ in range(2):
        with pytest.raises(CcmpError) as e:
            c.project_batch(q)
        assert e.value.code == -2
    again = c.project_batch(q)
    import torch

    assert all(torch.equal(a, b) for a, b in zip(ref, again))
