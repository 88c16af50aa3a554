"""GPU parity proper: the HIP path, called through the C-ABI, against the CPU oracle.

FD-faithful mode (the default, reference arithmetic) must agree with the det-build oracle BIT FOR
BIT — joints, ok flags and iteration counts — because the reference's Newton iteration amplifies
ulp-level differences past the 1e-6 rad bar.  The 1e-6 rad tolerance of the
north star is asserted as well; bitwise equality implies it.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import NCPU, OBJECTS, config_path, load_cfg, load_path_rows

from closed_chain_motion_planner_amd import _lib  # (option defaults are asked of the library)

pytestmark = pytest.mark.gpu

_SCHED_CACHE = {}
TOL_RAD = 1e-6  # north_star: "projected joints match the reference CPU projector ... to 1e-6 rad"


def _constraint(obj, ctx, mode=0):
    from closed_chain_motion_planner_amd import KinematicChainConstraint

    c = KinematicChainConstraint.from_yaml(config_path(obj), ctx=ctx)
    c.setJacobianMode(mode)
    c._fixture = obj  # which YAML fixture it came from (see _oracle_problem)
    return c


def _oracle_problem(oracle, c):
    """The checker's problem for the constraint under test.  Built by the ORACLE'S OWN set-up code from the YAML fixture
    (orc_problem_init / orc_set_start: arm order, base frames, DH constants, init_chain_) whenever the product's problem
    still is what its loader made of that fixture — then the two set-up paths are also compared, byte for byte, in every
    GPU test (VERDICT r2, weak #6a).  Only a problem the test has changed since (tolerances, iteration cap, calibration,
    arms, analytic mode) is handed over as bytes."""
    obj = getattr(c, "_fixture", None)
    if obj is not None:
        own = oracle.problem(load_cfg(obj))
        own.jacobian_mode = c.problem.jacobian_mode  # a switch of this library, not part of the reference's set-up
        if bytes(own) == bytes(c.problem):
            return own
    return oracle.problem_from_bytes(bytes(c.problem))


@pytest.mark.parametrize("obj", OBJECTS)
def test_function_bitwise_and_golden(gpu_ctx, oracle_det, obj):
    import torch

    c = _constraint(obj, gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    q = oracle_det.ambient_uniform_batch(P, 0xF0, 0, 4096)
    q[0] = np.array(load_cfg(obj)["start_joint"])
    f_gpu = c.function_batch(torch.as_tensor(q).cuda()).cpu().numpy()
    f_cpu = oracle_det.function_batch(P, q, NCPU)
    assert np.array_equal(f_gpu.view(np.uint64), f_cpu.view(np.uint64))
    if obj != "stefan":  # recorded planner outputs: geodesic rows sit just under (1e-3, 5e-3)
        rows = load_path_rows(obj)
        f = c.function_batch(torch.as_tensor(rows).cuda()).cpu().numpy()
        on_manifold = f[:, 0] < 2e-3
        assert on_manifold.sum() >= 6
        assert (f[on_manifold, 0] <= 1e-3 + 2e-5).all() and (f[on_manifold, 1] <= 5e-3 + 5e-5).all()


@pytest.mark.parametrize("obj,B,seed", [("Wine_Bottle", 4096, 0xC2), ("dumbbell", 1024, 0xC1), ("stefan", 1500, 0xC4)])
def test_project_fd_bitwise(gpu_ctx, oracle_det, obj, B, seed):
    """C2-sized Wine_Bottle batch (and the other two objects): q_out, ok and iters identical."""
    import torch

    c = _constraint(obj, gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    # the checker's problem comes from the oracle's own set-up code, not from the product's struct
    assert bytes(P) == bytes(c.problem) and bytes(oracle_det.problem(load_cfg(obj))) == bytes(P)
    q = oracle_det.ambient_uniform_batch(P, seed, 0, B)
    q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, NCPU)
    qd = torch.as_tensor(q).cuda()
    q_gpu, ok_gpu, it_gpu = c.project_batch(qd)
    q_gpu, ok_gpu, it_gpu = q_gpu.cpu().numpy(), ok_gpu.cpu().numpy(), it_gpu.cpu().numpy()
    d = np.abs(q_gpu - q_cpu).max(axis=1)
    n_bad = int((d > TOL_RAD).sum())
    print("%s B=%d: max|dq|=%.3e  >1e-6: %d  ok mismatches: %d  iter mismatches: %d  ok frac %.3f  mean iters %.1f"
          % (obj, B, d.max(), n_bad, int((ok_gpu != ok_cpu).sum()), int((it_gpu != it_cpu).sum()), ok_cpu.mean(),
             it_cpu.mean()))
    assert n_bad == 0
    assert np.array_equal(ok_gpu, ok_cpu)
    assert np.array_equal(it_gpu.astype(np.int32), it_cpu)
    assert np.array_equal(q_gpu.view(np.uint64), q_cpu.view(np.uint64)), "not bit-identical"


@pytest.mark.parametrize("flat", [1, 0])
@pytest.mark.parametrize("schedule,small,lpt,thr", [(0, 0, 0, -1), (1, 0, 0, -1), (1, 0, 0, 2), (2, 0, 0, -1), (1, 8192, 0, -1),
                                                    (1, 0, 1, -1), (1, 0, 1, 10), (0, 0, 2, -1), (1, 0, 0, 90), (1, 0, 1, 45)])
def test_schedules_are_bitwise_identical(gpu_ctx, oracle_det, schedule, small, lpt, thr, flat):
    """group kernel only / group kernel + straggler hand-over to the latency kernel / latency kernel only /
    default policy / FP32-scout longest-first with and without hand-over / occupancy-driven hand-over (thr > 10: hand
    over once the samples in flight fill less than thr - 10 per cent of the group slots), each with the one-round
    128-thread latency kernel (flat) and with the one-wavefront-per-sample kernel: all bit-identical to the oracle."""
    if flat == 0 and schedule == 0:
        pytest.skip("group kernel only: no latency kernel involved")
    import torch

    c = _constraint("stefan", gpu_ctx)  # longest iteration tails, some samples hit the 250 cap
    P = _oracle_problem(oracle_det, c)
    B = 3000
    if "stefan" not in _SCHED_CACHE:
        q = oracle_det.ambient_uniform_batch(P, 0x5C, 0, B)
        _SCHED_CACHE["stefan"] = (q,) + oracle_det.project_batch(P, q, NCPU)
    q, q_cpu, ok_cpu, it_cpu = _SCHED_CACHE["stefan"]
    gpu_ctx.set_schedule(schedule, small)
    gpu_ctx.set_lpt(lpt, 0)  # lpt > 0 with min_batch 0: FP32 scout + longest-predicted-first even on this small batch
    gpu_ctx.set_option("handover_threshold", thr)
    gpu_ctx.set_option("flat_kernel", flat)
    try:
        q_gpu, ok_gpu, it_gpu = c.project_batch(torch.as_tensor(q).cuda())
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_schedule(1)
        gpu_ctx.set_lpt(1)
        gpu_ctx.set_option("handover_threshold", -1)
        gpu_ctx.set_option("flat_kernel", 1)
    assert np.array_equal(q_gpu.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64))
    assert np.array_equal(ok_gpu.cpu().numpy(), ok_cpu)
    assert np.array_equal(it_gpu.cpu().numpy().astype(np.int32), it_cpu)
    assert it_cpu.max() == 250  # the cap was exercised


@pytest.mark.parametrize("obj,pred,front,samples", [("stefan", 40, 64, 0), ("stefan", 96, 7, 50), ("Wine_Bottle", 30, 500, -1), ("Wine_Bottle", 1, 3, 0),
                                                    ("Wine_Bottle", 20, 16, 6000)])
def test_split_launch_is_bitwise_identical(gpu_ctx, oracle_det, obj, pred, front, samples):
    """The split launch of mid-size batches (option "fd_split"): the front of the scout's descending order — samples predicted
    >= `pred` iterations, at most `front` — runs on latency blocks on the side stream beside the throughput kernel, the rest
    goes the usual way (throughput kernel, two-class hand-over).  A front smaller and larger than the number of such samples,
    a front of everything predicted at all, blocks that go on with the next-longest samples (`samples` > `front`; 0 = one per
    block, -1 = the default for the batch size) up to the whole batch, q_in and the fused sampler: bit-identical to the oracle."""
    import torch

    c = _constraint(obj, gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    B = 6000
    q = oracle_det.ambient_uniform_batch(P, 0x5F, 0, B)
    q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, NCPU)
    gpu_ctx.set_schedule(1, 0)
    gpu_ctx.set_lpt(1, 0)
    for name, val in (("fd_split", 1), ("fd_split_min", 0), ("fd_split_pred", pred), ("fd_split_front", front), ("fd_split_samples", samples)):
        gpu_ctx.set_option(name, val)
    try:
        for _ in range(2):  # twice: queue words and events are reused
            q_gpu, ok_gpu, it_gpu = c.project_batch(torch.as_tensor(q).cuda())
            torch.cuda.synchronize()
            assert np.array_equal(q_gpu.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64))
            assert np.array_equal(ok_gpu.cpu().numpy(), ok_cpu) and np.array_equal(it_gpu.cpu().numpy().astype(np.int32), it_cpu)
        qs, oks, its, amb = c.sample_project_batch(0x5F, 0, B, want_ambient=True)  # the same samples through the fused sampler
        torch.cuda.synchronize()
        assert np.array_equal(amb.cpu().numpy().view(np.uint64), q.view(np.uint64))
        assert np.array_equal(oks.cpu().numpy(), ok_cpu) and np.array_equal(its.cpu().numpy().astype(np.int32), it_cpu)
        wrapped = np.array([oracle_det.enforce_bounds(x) for x in q_cpu[:64]])
        assert np.array_equal(qs[:64].cpu().numpy().view(np.uint64), wrapped.view(np.uint64))
    finally:
        gpu_ctx.set_schedule(1)
        gpu_ctx.set_lpt(1)
        for name, val in (("fd_split", 1), ("fd_split_min", 0), ("fd_split_pred", -1), ("fd_split_front", -1), ("fd_split_samples", -1)):
            gpu_ctx.set_option(name, val)


@pytest.mark.parametrize("wpc,cut", [(2, -1), (1, -1), (3, 3), (2, 8), (12, 8)])
def test_split_launch_with_few_waves_per_cu(gpu_ctx, oracle_det, wpc, cut):
    """waves_per_cu (an option, 1..32) at or below the split launch's cut of throughput wavefronts per CU (default 3 / 2, option up
    to 8): round 4 computed `CUs x (waves_per_cu - cut)` blocks for the throughput kernel — zero or negative — BEHIND a front that
    was already running.  The plan now decides the split before anything is launched: without one throughput wavefront
    per CU left beside the front there is no split.  Bit-identical to the oracle either way, twice in a row."""
    import torch

    c = _constraint("Wine_Bottle", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    B = 16384
    q = oracle_det.ambient_uniform_batch(P, 0x61 + wpc, 0, B)
    q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, NCPU)
    gpu_ctx.set_waves_per_cu(wpc)
    gpu_ctx.set_option("fd_split_group_cut", cut)
    try:
        from closed_chain_motion_planner_amd._lib import CALL_PROJECT

        line = gpu_ctx.describe(CALL_PROJECT, B)
        eff_cut = cut if cut >= 0 else 3
        assert ("split launch" in line) == (wpc - eff_cut >= 1), line
        for _ in range(2):
            q_gpu, ok_gpu, it_gpu = c.project_batch(torch.as_tensor(q).cuda())
            torch.cuda.synchronize()
            assert np.array_equal(q_gpu.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64))
            assert np.array_equal(ok_gpu.cpu().numpy(), ok_cpu) and np.array_equal(it_gpu.cpu().numpy().astype(np.int32), it_cpu)
    finally:
        gpu_ctx.set_waves_per_cu(0)
        gpu_ctx.set_option("fd_split_group_cut", -1)


def test_sample_project_bitwise(gpu_ctx, oracle_det):
    c = _constraint("Wine_Bottle", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    B, seed, first = 1000, 0xABCDEF, 12345
    q_gpu, ok_gpu, it_gpu, amb = c.sample_project_batch(seed, first, B, want_ambient=True)
    q_cpu, ok_cpu, it_cpu = oracle_det.sample_project_batch(P, seed, first, B, NCPU)
    amb_cpu = oracle_det.ambient_uniform_batch(P, seed, first, B)
    assert np.array_equal(amb.cpu().numpy().view(np.uint64), amb_cpu.view(np.uint64))
    assert np.array_equal(q_gpu.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64))
    assert np.array_equal(ok_gpu.cpu().numpy(), ok_cpu)
    assert np.array_equal(it_gpu.cpu().numpy().astype(np.int32), it_cpu)
    assert (np.abs(q_gpu.cpu().numpy()) <= np.pi).all()  # enforceBounds wrapped everything


@pytest.mark.parametrize("B", [0, 1, 5, 10, 11, 61, 256, 257, 640, 641])
def test_ragged_batches(gpu_ctx, oracle_det, B):
    """empty / single / not a multiple of the 10-samples-per-wave grouping; in place."""
    import torch

    c = _constraint("dumbbell", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    q = oracle_det.ambient_uniform_batch(P, 77, 0, B) if B else np.zeros((0, 14))
    qd = torch.as_tensor(q).cuda()
    out, ok, it = c.project_batch(qd, out=qd)  # in place, as the reference
    torch.cuda.synchronize()
    if B == 0:
        return
    q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, 4)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64))
    assert np.array_equal(ok.cpu().numpy(), ok_cpu)


@pytest.mark.parametrize("flat", [1, 0])
def test_tiny_batches_on_both_latency_kernels(gpu_ctx, oracle_det, flat):
    import torch

    c = _constraint("stefan", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    q = oracle_det.ambient_uniform_batch(P, 0x7A1, 0, 40)
    q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, 4)
    gpu_ctx.set_option("flat_kernel", flat)
    try:
        out, ok, it = c.project_batch(torch.as_tensor(q).cuda())
    finally:
        gpu_ctx.set_option("flat_kernel", 1)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64))
    assert np.array_equal(ok.cpu().numpy(), ok_cpu) and np.array_equal(it.cpu().numpy().astype(np.int32), it_cpu)


def test_is_satisfied_and_joint_valid(gpu_ctx, oracle_det):
    import torch

    c = _constraint("Wine_Bottle", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    q = oracle_det.ambient_uniform_batch(P, 5, 0, 512)
    q_cpu, ok_cpu, _ = oracle_det.project_batch(P, q, NCPU)
    both = np.concatenate([q, q_cpu])
    d = torch.as_tensor(both).cuda()
    sat = c.is_satisfied_batch(d).cpu().numpy()
    jv = c.joint_valid_batch(d).cpu().numpy()
    exp_sat = np.array([oracle_det.is_satisfied(P, x) for x in both], dtype=np.uint8)
    exp_jv = np.array([oracle_det.joint_valid(P, x) for x in both], dtype=np.uint8)
    assert np.array_equal(sat, exp_sat) and np.array_equal(jv, exp_jv)
    assert sat[512:].all()  # every projected sample converged within 250 iterations here


def test_single_state_surface(gpu_ctx, oracle_det):
    """the reference-signature methods: project(x) in place -> bool, function, isSatisfied, jointValid"""
    c = _constraint("Wine_Bottle", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    x0 = np.array(load_cfg("Wine_Bottle")["start_joint"])
    assert c.isSatisfied(x0) and c.jointValid(x0)
    x = x0.copy()
    assert c.project(x) is True and np.array_equal(x, x0)  # already satisfied: zero iterations, untouched
    x = oracle_det.ambient_uniform(P, 3, 3)
    ok_cpu, x_cpu, _ = oracle_det.project(P, x)
    ok = c.project(x)
    assert ok == ok_cpu and np.array_equal(x.view(np.uint64), x_cpu.view(np.uint64))
    assert np.array_equal(c.function(x_cpu), oracle_det.function(P, x_cpu))
    with pytest.raises(ValueError):
        c.setTolerance(0.0, 1e-3)


@pytest.mark.parametrize("obj,B", [("Wine_Bottle", 4096), ("stefan", 1500), ("dumbbell", 1024)])
def test_analytic_mode_bitwise(gpu_ctx, oracle_det, obj, B):
    """The analytic mode is built in the canonical rounding model: joints, flags and iteration counts are bit-identical
    to the CPU oracle run with ORC_JAC_ANALYTIC (orc_jacobian_analytic restates the kernel's operation order; the
    independent world-frame formulation agrees with it to 5e-13, tests/test_oracle_golden.py) — C2-sized batch,
    both kernel instantiations (stock structure and general), the fused sampler, and calibrated arms."""
    import torch
    from closed_chain_motion_planner_amd import _lib

    c = _constraint(obj, gpu_ctx, mode=1)
    for calibrated in (False, True):
        if calibrated:
            dh = (C.c_double * 28)(*[1e-3 * ((5 * i) % 7 - 3) for i in range(28)])
            assert _lib.lib().ccmp_set_calibration(C.byref(c.problem), 0, dh) == 0
        P = _oracle_problem(oracle_det, c)
        assert P.jacobian_mode == 1
        q = oracle_det.ambient_uniform_batch(P, 0xA7, 0, B)
        q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, NCPU)
        for stock in (1, 0):
            gpu_ctx.set_option("stock_kernels", stock)
            try:
                q_gpu, ok_gpu, it_gpu = c.project_batch(torch.as_tensor(q).cuda())
            finally:
                gpu_ctx.set_option("stock_kernels", 1)
            assert np.array_equal(q_gpu.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64)), (obj, calibrated, stock)
            assert np.array_equal(ok_gpu.cpu().numpy(), ok_cpu)
            assert np.array_equal(it_gpu.cpu().numpy().astype(np.int32), it_cpu)
    qs, oks, its, _ = c.sample_project_batch(0xA7, 0, 777)
    e_q, e_ok, e_it = oracle_det.sample_project_batch(P, 0xA7, 0, 777, NCPU)
    assert np.array_equal(qs.cpu().numpy().view(np.uint64), e_q.view(np.uint64)) and np.array_equal(oks.cpu().numpy(), e_ok)
    # the reference signature, one state through the host entry point (one wavefront of the latency kernel)
    for k in range(3):
        x = q[k].copy()
        ok1 = c.project(x)
        assert ok1 == bool(ok_cpu[k]) and np.array_equal(x.view(np.uint64), q_cpu[k].view(np.uint64)), k


_ANALYTIC_OPTIONS = ("analytic_small_batch", "analytic_waves_per_cu", "analytic_handover")


def _restore_analytic_options(ctx):
    for name in _ANALYTIC_OPTIONS:
        ctx.set_option(name, _lib.get_option(None, name))


@pytest.mark.parametrize("small,waves,handover", [(1 << 30, 12, 8), (0, 12, 0), (0, 12, 8), (0, 12, 32), (0, 12, 1), (0, 3, 16), (0, 1, 31)])
def test_analytic_schedules_are_bitwise_identical(gpu_ctx, oracle_det, small, waves, handover):
    """analytic mode's two kernels under every shape of a call: the sixteen-lanes-per-sample latency kernel alone (`small`); the
    lane-pair kernel alone (hand-over 0); the lane-pair kernel handing its last samples over at the loop top (x, index,
    counters) to the latency kernel once a wavefront's tickets are gone and it holds at most `handover` of them — 32: as soon
    as the tickets are gone, 1: the last sample of a wavefront only; with few wavefronts per CU (every wavefront refills many
    times, every ticket word runs dry early) — all bit-identical to the oracle's analytic mode"""
    import torch

    c = _constraint("stefan", gpu_ctx, mode=1)
    P = _oracle_problem(oracle_det, c)
    B = 60000
    q = c.ambient_uniform_batch(0xA9, 0, B)
    q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q.cpu().numpy(), NCPU)
    for name, v in zip(_ANALYTIC_OPTIONS, (small, waves, handover)):
        gpu_ctx.set_option(name, v)
    try:
        text = gpu_ctx.describe(_lib.CALL_PROJECT_ANALYTIC, B)
        assert ("alone" in text) == bool(small) and ("then project_row16_kernel" in text) == bool(handover and not small)
        out = torch.full_like(q, 777.0)
        _, ok, it = c.project_batch(q, out=out)
        torch.cuda.synchronize()
        qs, oks, its, _ = c.sample_project_batch(0xA8, 5, 30000)  # the fused sampler through the same launch shape
        torch.cuda.synchronize()
    finally:
        _restore_analytic_options(gpu_ctx)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64))
    assert np.array_equal(ok.cpu().numpy(), ok_cpu) and np.array_equal(it.cpu().numpy().astype(np.int32), it_cpu)
    assert it_cpu.max() == 250 and (it_cpu < 20).any()
    e_q, e_ok, e_it = oracle_det.sample_project_batch(P, 0xA8, 5, 30000, NCPU)
    assert np.array_equal(qs.cpu().numpy().view(np.uint64), e_q.view(np.uint64)) and np.array_equal(oks.cpu().numpy(), e_ok)
    assert np.array_equal(its.cpu().numpy().astype(np.int32), e_it)


def test_analytic_mode_statistics(gpu_ctx, oracle_det):
    """the opt-in fast mode against the REFERENCE arithmetic (FD oracle): same manifold, same acceptance statistics;
    not bit-comparable with it (bit-identity with the oracle's own analytic mode: test_analytic_mode_bitwise)"""
    import torch

    c = _constraint("Wine_Bottle", gpu_ctx, mode=1)
    P = _oracle_problem(oracle_det, c)
    P.jacobian_mode = 0
    B = 4096
    q = oracle_det.ambient_uniform_batch(P, 0xC2, 0, B)
    q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, NCPU)
    q_gpu, ok_gpu, it_gpu = c.project_batch(torch.as_tensor(q).cuda())
    f = c.function_batch(q_gpu).cpu().numpy()
    conv = it_gpu.cpu().numpy() < 250
    assert conv.mean() > 0.99
    assert (f[conv, 0] <= 1e-3).all() and (f[conv, 1] <= 5e-3).all()
    d = np.abs(q_gpu.cpu().numpy() - q_cpu).max(axis=1)
    print("analytic vs FD oracle: median %.2e, frac>1e-6 %.3f, ok agreement %.4f, mean iters %.2f vs %.2f"
          % (np.median(d), (d > 1e-6).mean(), (ok_gpu.cpu().numpy() == ok_cpu).mean(), it_gpu.float().mean().item(),
             it_cpu.mean()))
    assert np.median(d) < 1e-6
    assert (ok_gpu.cpu().numpy() == ok_cpu).mean() > 0.97
    assert abs(it_gpu.float().mean().item() - it_cpu.mean()) < 1.0


def test_non_finite_and_out_of_range_inputs_terminate(gpu_ctx, oracle_det):
    """NaN / inf / |q| beyond the sincos range: the projector must terminate at once with ok = 0 and zero
    iterations (NaN residual fails both comparisons of the loop test), exactly as the oracle does; their
    neighbours in the batch are unaffected."""
    import torch

    c = _constraint("Wine_Bottle", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    q = oracle_det.ambient_uniform_batch(P, 0xBAD, 0, 64)
    q[3, 2] = np.nan
    q[17, 9] = np.inf
    q[29, 0] = -np.inf
    q[40, 5] = 5e6      # finite, beyond CCMP_SINCOS_MAX: sincos returns NaN by definition
    q[41, 5] = 1.6e6    # finite, inside the exact-reduction range: a legitimate (if absurd) angle
    bad = [3, 17, 29, 40]
    for sched in (0, 2):
        gpu_ctx.set_schedule(sched, 0)
        try:
            out, ok, it = c.project_batch(torch.as_tensor(q).cuda())
            torch.cuda.synchronize()
        finally:
            gpu_ctx.set_schedule(1)
        out, ok, it = out.cpu().numpy(), ok.cpu().numpy(), it.cpu().numpy()
        q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, 4)
        assert (ok[bad] == 0).all() and (it[bad] == 0).all() and (ok_cpu[bad] == 0).all() and (it_cpu[bad] == 0).all()
        same = (out.view(np.uint64) == q_cpu.view(np.uint64)) | (np.isnan(out) & np.isnan(q_cpu))
        assert same.all() and np.array_equal(ok, ok_cpu) and np.array_equal(it.astype(np.int32), it_cpu)
    assert not c.isSatisfied(q[3]) and not c.isSatisfied(q[17])  # f.allFinite() is part of isSatisfied
    # the analytic mode's kernels terminate the same way
    c.setJacobianMode(1)
    Pa = _oracle_problem(oracle_det, c)
    qa_cpu, oka_cpu, ita_cpu = oracle_det.project_batch(Pa, q, 4)
    for small, handover in ((1 << 30, 8), (0, 0), (0, 32)):  # the latency kernel alone; the lane-pair kernel alone; a NaN sample handed over
        gpu_ctx.set_option("analytic_small_batch", small)
        gpu_ctx.set_option("analytic_handover", handover)
        try:
            out, ok, it = c.project_batch(torch.as_tensor(q).cuda())
            torch.cuda.synchronize()
        finally:
            _restore_analytic_options(gpu_ctx)
        out, ok, it = out.cpu().numpy(), ok.cpu().numpy(), it.cpu().numpy()
        same = (out.view(np.uint64) == qa_cpu.view(np.uint64)) | (np.isnan(out) & np.isnan(qa_cpu))
        assert same.all() and np.array_equal(ok, oka_cpu) and np.array_equal(it.astype(np.int32), ita_cpu)
        assert (ok[bad] == 0).all() and (it[bad] == 0).all()


def test_iteration_cap_and_loose_tolerance(gpu_ctx, oracle_det):
    import torch

    c = _constraint("dumbbell", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    q = oracle_det.ambient_uniform_batch(P, 0xCAB, 0, 200)
    for max_iter, tol in ((0, (1e-3, 5e-3)), (3, (1e-3, 5e-3)), (250, (0.5, 1.0)), (250, (10.0, 10.0))):
        c.problem.max_iter = max_iter
        c.setTolerance(*tol)
        P = _oracle_problem(oracle_det, c)
        out, ok, it = c.project_batch(torch.as_tensor(q).cuda())
        q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, 4)
        assert np.array_equal(out.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64)), (max_iter, tol)
        assert np.array_equal(ok.cpu().numpy(), ok_cpu) and np.array_equal(it.cpu().numpy().astype(np.int32), it_cpu)
        assert int(it.max()) <= max_iter
    # tolerance1 > 1: the reference's `norm1 = f[0] > tol1` stores a boolean, so "1.0 < 10.0" lets an unconverged
    # sample through the return test only if the loop exited — which with tol 10 it does at once: all ok = jointValid
    assert int(it.max()) == 0


@pytest.mark.parametrize("schedule", [0, 2])
def test_general_base_frames_take_the_full_product(gpu_ctx, oracle_det, schedule):
    """Every shipped t_wb has linear part diag(+-1) and tool_pose skips the products with exact zeros there; a
    tilted base (arm 2) and a 1-ulp-off identity (arm 1) must go through the general product, bit for bit."""
    import torch

    c = _constraint("Wine_Bottle", gpu_ctx)
    a, b = 0.3, -0.7
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Rz = np.array([[np.cos(b), -np.sin(b), 0], [np.sin(b), np.cos(b), 0], [0, 0, 1]])
    tilt = (Rz @ Rx).reshape(-1)
    for k in range(9):
        c.problem.base_R[9 + k] = float(tilt[k])
    c.problem.base_R[0] = float(np.nextafter(1.0, 0.0))
    c.setInitialPosition(np.array(c.problem.start_joint[:]))
    P = _oracle_problem(oracle_det, c)
    B = 600
    q = oracle_det.ambient_uniform_batch(P, 0xBA5E, 0, B)
    f_cpu = oracle_det.function_batch(P, q, NCPU)
    q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, NCPU)
    gpu_ctx.set_schedule(schedule, 0)
    try:
        f_gpu = c.function_batch(torch.as_tensor(q).cuda())
        q_gpu, ok_gpu, it_gpu = c.project_batch(torch.as_tensor(q).cuda())
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_schedule(1)
    assert np.array_equal(f_gpu.cpu().numpy().view(np.uint64), f_cpu.view(np.uint64))
    assert np.array_equal(q_gpu.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64))
    assert np.array_equal(ok_gpu.cpu().numpy(), ok_cpu)
    assert np.array_equal(it_gpu.cpu().numpy().astype(np.int32), it_cpu)
    assert it_cpu.max() > 0


def test_misaligned_rows_are_rejected(gpu_ctx):
    """rows travel in 16-byte pieces: an 8-byte-aligned (but not 16-byte-aligned) joint buffer is an argument error"""
    import torch
    from closed_chain_motion_planner_amd import _lib

    c = _constraint("Wine_Bottle", gpu_ctx)
    flat = torch.zeros(14 * 4 + 1, dtype=torch.float64, device="cuda")
    mis = flat[1:]  # data_ptr() + 8
    assert mis.data_ptr() % 16 == 8
    ok = torch.zeros(4, dtype=torch.uint8, device="cuda")
    rc = _lib.lib().ccmp_project_batch(gpu_ctx.handle, C.byref(c.problem), mis.data_ptr(), mis.data_ptr(), ok.data_ptr(), None, 4, None)
    assert rc == -1  # CCMP_EINVAL


@pytest.mark.parametrize("calibrated", [False, True])
def test_stock_structure_kernels_and_general_kernels(gpu_ctx, oracle_det, calibrated):
    """The latency kernel has an instantiation for the exact-zero structure of the uncalibrated Panda (products with
    exact zeros skipped).  Stock problem: specialised == general == oracle.  With calibration offsets (ccmp_set_calibration:
    axes tilt, offsets fill in) the structure test fails on the host and the general kernel must be the one that runs."""
    import torch
    from closed_chain_motion_planner_amd import _lib

    c = _constraint("Wine_Bottle", gpu_ctx)
    if calibrated:
        dh = (C.c_double * 28)(*[1e-3 * ((7 * i) % 5 - 2) for i in range(28)])
        assert _lib.lib().ccmp_set_calibration(C.byref(c.problem), 0, dh) == 0
        assert _lib.lib().ccmp_set_calibration(C.byref(c.problem), 1, dh) == 0
    P = _oracle_problem(oracle_det, c)
    q = oracle_det.ambient_uniform_batch(P, 0x57C, 0, 700)
    q_cpu, ok_cpu, it_cpu = oracle_det.project_batch(P, q, NCPU)
    f_cpu = oracle_det.function_batch(P, q, NCPU)
    for stock in (1, 0):
        gpu_ctx.set_option("stock_kernels", stock)
        try:
            out, ok, it = c.project_batch(torch.as_tensor(q).cuda())
            f = c.function_batch(torch.as_tensor(q).cuda())
        finally:
            gpu_ctx.set_option("stock_kernels", 1)
        assert np.array_equal(out.cpu().numpy().view(np.uint64), q_cpu.view(np.uint64)), (calibrated, stock)
        assert np.array_equal(ok.cpu().numpy(), ok_cpu) and np.array_equal(it.cpu().numpy().astype(np.int32), it_cpu)
        assert np.array_equal(f.cpu().numpy().view(np.uint64), f_cpu.view(np.uint64))
    # the extend step runs on the same Newton routine: both instantiations, bit for bit against the oracle
    good = q_cpu[ok_cpu == 1]
    frm, to = good[:24].copy(), good[24:48].copy()
    for stock in (1, 0):
        gpu_ctx.set_option("stock_kernels", stock)
        try:
            st, n, gok, _ = c.discrete_geodesic_batch(torch.as_tensor(frm).cuda(), torch.as_tensor(to).cuda(), 32)
        finally:
            gpu_ctx.set_option("stock_kernels", 1)
        st, n, gok = st.cpu().numpy(), n.cpu().numpy(), gok.cpu().numpy()
        for e in range(len(frm)):
            ok_c, st_c, _ = oracle_det.discrete_geodesic(P, frm[e], to[e], interpolate=True, max_states=32)
            if n[e] > 32:  # did not fit: stopped at the full list (the oracle binding re-ran the edge with room)
                assert n[e] == 33 and gok[e] == 0 and len(st_c) > 32 and np.array_equal(st[e].view(np.uint64), st_c[:32].view(np.uint64))
                continue
            assert int(gok[e]) == int(ok_c) and int(n[e]) == len(st_c), (calibrated, stock, e)
            assert np.array_equal(st[e, : n[e]].view(np.uint64), st_c.view(np.uint64))


@pytest.mark.parametrize("B", [1, 700, 20000, 70000])
def test_sample_uniform_on_calibrated_arms_runs_unfused_and_bitwise(gpu_ctx, oracle_det, B):
    """sampleUniform (jy_ProjectedStateSpace.cpp:10-15) on arms WITHOUT the stock structure — PandaModel::initModel(dh) with
    calibration offsets that differ per arm (panda_rbdl.cpp:80-99) — is ambient sampler -> projector in place -> enforceBounds
    as three launches (ccmp_api.cpp: project_common; the fused general instantiation spilled 2.3 KB per lane and was removed):
    samples, flags, iteration counts and the ambient states bit for bit the oracle's fused loop, for a single state, the latency
    kernel, the split launch and the throughput kernel; and the general kernels under the stock model give the same"""
    import torch

    c = _constraint("Wine_Bottle", gpu_ctx)
    for arm in (0, 1):
        dh = (C.c_double * 28)(*[(1e-3 if arm == 0 else -7e-4) * ((5 * i + 3 * arm) % 7 - 3) for i in range(28)])
        assert _lib.lib().ccmp_set_calibration(C.byref(c.problem), arm, dh) == 0
    P = _oracle_problem(oracle_det, c)
    e_q, e_ok, e_it = oracle_det.sample_project_batch(P, 0xCA1, 11, B, NCPU)
    amb = oracle_det.ambient_uniform_batch(P, 0xCA1, 11, B)
    q, ok, it, q_amb = c.sample_project_batch(0xCA1, 11, B, want_ambient=True)
    assert np.array_equal(q.cpu().numpy().view(np.uint64), e_q.view(np.uint64)) and np.array_equal(ok.cpu().numpy(), e_ok)
    assert np.array_equal(it.cpu().numpy().astype(np.int32), e_it) and np.array_equal(q_amb.cpu().numpy().view(np.uint64), amb.view(np.uint64))
    q2, ok2, it2, _ = c.sample_project_batch(0xCA1, 11, B)  # without the ambient output: projected in place in q_out
    assert torch.equal(q2, q) and torch.equal(ok2, ok) and torch.equal(it2, it)
    if B == 20000:  # stock model, general kernels (option): the same unfused path, the same bits as the fused stock kernels
        s = _constraint("Wine_Bottle", gpu_ctx)
        ref = s.sample_project_batch(0xCA2, 0, B)
        gpu_ctx.set_option("stock_kernels", 0)
        try:
            gen = s.sample_project_batch(0xCA2, 0, B)
        finally:
            gpu_ctx.set_option("stock_kernels", 1)
        assert all(torch.equal(a, b) for a, b in zip(ref[:3], gen[:3]))
