"""Proxy-geometry clearance on the GPU (ccmp_clearance_batch / _host through the C ABI) against the CPU checker
oracle/ccmp_oracle.c:orc_clearance: distances, the pair that attains them and the flags, bit for bit."""
import numpy as np
import pytest

from conftest import NCPU, OBJECTS, load_cfg, load_path_rows
from test_gpu_parity import _constraint, _oracle_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["per_state", "tiles", "default"])
def kernel_choice(request, gpu_ctx):
    """both kernels on every case: one block per state (small batches by default) and 64-state tiles (large ones)"""
    gpu_ctx.set_option("clearance_per_state_max", {"per_state": 1 << 30, "tiles": 0, "default": 8192}[request.param])
    yield request.param
    gpu_ctx.set_option("clearance_per_state_max", 8192)


def _bits_equal(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def _states(oracle, P, n, seed):
    """half ambient samples, half their projections (on or near the manifold: what the planner feeds the checker)"""
    q = oracle.ambient_uniform_batch(P, seed, 0, n)
    proj, _, _ = oracle.project_batch(P, q[: n // 2], NCPU)
    return np.concatenate([q[n // 2:], proj])


@pytest.mark.parametrize("obj", OBJECTS)
def test_default_scene_bitwise(gpu_ctx, oracle_det, obj, kernel_choice):
    import torch
    from closed_chain_motion_planner_amd import scene as S

    c = _constraint(obj, gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    chk = S.ProxyValidityChecker(c)
    sc = chk.scene
    q = _states(oracle_det, P, 4096, 0x5CE0)
    clr, pair, free = sc.clearance_batch(torch.as_tensor(q).cuda(), margin=0.0)
    clr_o, pair_o = oracle_det.clearance_batch(P, sc.spheres, sc.boxes, sc.allowed, q)
    assert _bits_equal(clr.cpu().numpy(), clr_o).all()
    assert np.array_equal(pair.cpu().numpy(), pair_o)
    assert np.array_equal(free.cpu().numpy(), (clr_o > 0.0).astype(np.uint8))
    n_tested = oracle_det.clearance(P, sc.spheres, sc.boxes, sc.allowed, q[0])[2]
    assert sc.num_pairs == n_tested
    print("%s: %d spheres, %d pairs; %.1f %% of states keep a positive clearance" % (obj, len(sc.spheres), n_tested,
                                                                                      100.0 * (clr_o > 0).mean()))


def test_full_scene_ragged_batches_and_flags(gpu_ctx, oracle_det, kernel_choice):
    """64 spheres on every kind of frame (the > 64 KB LDS configuration), 8 boxes (turned ones among them), a random
    allowed-pair matrix; batch sizes around the 64-state tile; ok_in and margin; a non-finite state"""
    import torch
    from closed_chain_motion_planner_amd import scene as S

    c = _constraint("stefan", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    rng = np.random.default_rng(0x5CE1)
    frames = [S.FRAME_WORLD] + list(range(18))
    sph = [(int(rng.choice(frames)), int(rng.integers(0, 32)), tuple(rng.uniform(-0.15, 0.15, 3)), float(rng.uniform(0.0, 0.08)))
           for _ in range(64)]
    sph[5] = (S.FRAME_WORLD, 3, (0.4, 0.0, 1.5), 0.2)
    boxes = []
    for b in range(8):
        A = rng.standard_normal((3, 3))
        Q, _ = np.linalg.qr(A)
        R = np.eye(3) if b < 2 else Q
        boxes.append((int(rng.integers(0, 32)), tuple(rng.uniform([-0.2, -0.8, 0.6], [0.9, 0.8, 1.8])), R, tuple(rng.uniform(0.0, 0.3, 3))))
    allowed = [int(v) for v in rng.integers(0, 2 ** 32, 32, dtype=np.uint64) & rng.integers(0, 2 ** 32, 32, dtype=np.uint64)]
    sc = S.ProxyScene(c, sph, boxes, allowed)
    q_all = _states(oracle_det, P, 2048, 0x5CE2)
    q_all[7, 3] = np.nan
    q_all[8, 12] = np.inf
    for B in (1, 63, 64, 65, 1000, 2048):
        q = q_all[:B].copy()
        ok = (rng.integers(0, 2, B)).astype(np.uint8)
        clr, pair, free = sc.clearance_batch(torch.as_tensor(q).cuda(), margin=0.02, ok=torch.as_tensor(ok).cuda())
        clr_o, pair_o = oracle_det.clearance_batch(P, sph, boxes, allowed, q)
        assert _bits_equal(clr.cpu().numpy(), clr_o).all(), B
        assert np.array_equal(pair.cpu().numpy(), pair_o), B
        with np.errstate(invalid="ignore"):
            assert np.array_equal(free.cpu().numpy(), ((ok != 0) & (clr_o > 0.02)).astype(np.uint8)), B
    assert np.isnan(clr_o[7]) and np.isnan(clr_o[8]) and pair_o[7] == -1
    assert len(set(pair_o.tolist())) > 20  # many different pairs attain the minimum: the pair bookkeeping is exercised
    assert (pair_o >> 8 >= 64).any() and (pair_o >> 8 < 64).any()  # boxes and spheres both


def test_degenerate_scenes(gpu_ctx, oracle_det, kernel_choice):
    import torch
    from closed_chain_motion_planner_amd import CcmpError
    from closed_chain_motion_planner_amd import scene as S

    c = _constraint("Wine_Bottle", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    q = _states(oracle_det, P, 128, 3)
    qd = torch.as_tensor(q).cuda()
    # nothing to test: +inf, pair -1, free
    for sph, boxes in (([], []), ([(S.frame(0, 2), 0, (0, 0, 0), 0.1)], []), ([], [S.ProxyValidityChecker.SUB_TABLE])):
        sc = S.ProxyScene(c, sph, boxes, None)
        clr, pair, free = sc.clearance_batch(qd)
        assert sc.num_pairs == 0 and torch.isinf(clr).all() and (pair == -1).all() and (free == 1).all()
    # one sphere against the table: only box pairs
    sc = S.ProxyScene(c, [(S.frame(1, 7), 0, (0, 0, 0.05), 0.03)], [S.ProxyValidityChecker.SUB_TABLE], None)
    clr, pair, _ = sc.clearance_batch(qd)
    clr_o, pair_o = oracle_det.clearance_batch(P, sc.spheres, sc.boxes, None, q)
    assert _bits_equal(clr.cpu().numpy(), clr_o).all() and np.array_equal(pair.cpu().numpy(), pair_o)
    # argument errors as in the header
    for bad in ([(18, 0, (0, 0, 0), 0.1)], [(-2, 0, (0, 0, 0), 0.1)], [(0, 32, (0, 0, 0), 0.1)], [(0, 0, (0, 0, 0), -0.1)],
                [(0, 0, (np.nan, 0, 0), 0.1)], [(0, 0, (0, 0, 0), 0.1)] * 65):
        with pytest.raises(CcmpError):
            S.ProxyScene(c, bad, [], None)
    with pytest.raises(CcmpError):
        S.ProxyScene(c, [], [(0, (0, 0, 0), np.eye(3), (0.1, -0.1, 0.1))], None)


def test_host_entry_and_checker_mirror(gpu_ctx, oracle_det, kernel_choice):
    """one state through the host entry point (what a StateValidityChecker wrapper calls), the reference-shaped mirror,
    and the pipeline project -> pre-filter -> compact on the device"""
    import torch
    from closed_chain_motion_planner_amd import scene as S

    c = _constraint("Wine_Bottle", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    chk = S.ProxyValidityChecker(c)
    rows = load_path_rows("Wine_Bottle")
    for x in rows[:6]:
        clr, pair, free = chk.scene.clearance(x)
        clr_o, pair_o, _ = oracle_det.clearance(P, chk.scene.spheres, chk.scene.boxes, chk.scene.allowed, x)
        assert clr == clr_o and pair == pair_o and free == (clr_o > 0)
        assert chk.isValid(x)  # states of the reference's own solution path
    asked = []
    chk2 = S.ProxyValidityChecker(c, inner=lambda x: asked.append(1) or False)
    assert chk2.isValid(rows[0]) is False and len(asked) == 1  # free of proxy contact: the exact checker decides
    folded = rows[0].copy(); folded[7:] = folded[:7]           # arm 1 mirrors arm 0 into the same volume
    clr, _, _ = chk.scene.clearance(folded)
    # a batch of host states
    clr_b, pair_b, free_b = chk.scene.clearance(rows[:20])
    clr_o, pair_o = oracle_det.clearance_batch(P, chk.scene.spheres, chk.scene.boxes, chk.scene.allowed, rows[:20])
    assert _bits_equal(clr_b, clr_o).all() and np.array_equal(pair_b, pair_o)
    # grasped object as spheres on arm 0's hand: allowed against both hands, tested against everything else
    chk.attachObject([[0, 0, 0], [0, 0, 0.1], [0, 0, -0.1]], 0.03)
    sc = chk.scene
    assert sum(1 for s in sc.spheres if s[1] == S.GROUP_OBJECT) == 3
    q = _states(oracle_det, P, 1024, 11)
    clr, pair, free = sc.clearance_batch(torch.as_tensor(q).cuda())
    clr_o, pair_o = oracle_det.clearance_batch(P, sc.spheres, sc.boxes, sc.allowed, q)
    assert _bits_equal(clr.cpu().numpy(), clr_o).all() and np.array_equal(pair.cpu().numpy(), pair_o)
    # pipeline: sampleUniform -> project -> pre-filter (flags chained through ok_in) -> compaction
    qp, ok, _, _ = c.sample_project_batch(0xF17, 0, 8192, want_iters=False)
    flags = chk.filter_batch(qp, ok)
    kept, cnt = c.compact_valid(qp, flags)
    torch.cuda.synchronize()
    n = int(cnt.item())
    qh, okh = qp.cpu().numpy(), ok.cpu().numpy()
    clr_o, _ = oracle_det.clearance_batch(P, sc.spheres, sc.boxes, sc.allowed, qh)
    # the default skeleton spheres are not inscribed in the links: the checker refuses only overlaps deeper than 3 cm
    # (a caller's own proxies keep the strict threshold 0)
    assert chk.margin == S.DEFAULT_SKELETON_MARGIN == -0.03 and S.ProxyValidityChecker(c, spheres=sc.spheres).margin == 0.0
    expect = qh[(okh != 0) & (clr_o > chk.margin)]
    assert n == len(expect) and np.array_equal(kept[:n].cpu().numpy().view(np.uint64), expect.view(np.uint64))
    print("pipeline: %d sampled, %d on the manifold and within limits, %d also clear of proxy contact" % (len(qh), int(okh.sum()), n))


def test_full_size_batch(gpu_ctx, oracle_det):
    """BASELINE's batch (262 144 states): spot-checked against the checker, flags consistent with the distances"""
    import time

    import torch
    from closed_chain_motion_planner_amd import scene as S

    c = _constraint("Wine_Bottle", gpu_ctx)
    P = _oracle_problem(oracle_det, c)
    sc = S.ProxyValidityChecker(c).scene
    B = 262144
    q = c.ambient_uniform_batch(0xB16, 0, B)
    clr, pair, free = sc.clearance_batch(q, margin=0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        sc.clearance_batch(q, margin=0.01)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    clr_h, free_h = clr.cpu().numpy(), free.cpu().numpy()
    assert np.array_equal(free_h, (clr_h > 0.01).astype(np.uint8))
    idx = np.random.default_rng(2).integers(0, B, 2000)
    idx[:3] = [0, B - 1, B - 64]
    clr_o, pair_o = oracle_det.clearance_batch(P, sc.spheres, sc.boxes, sc.allowed, q.cpu().numpy()[idx])
    assert _bits_equal(clr_h[idx], clr_o).all() and np.array_equal(pair.cpu().numpy()[idx], pair_o)
    print("clearance: %d states x %d pairs in %.3f ms = %.1f M states/s" % (B, sc.num_pairs, dt * 1e3, B / dt / 1e6))
