"""The CPU oracle against everything that pins it: the reference's recorded planner outputs
(tests/golden/paths, from debug/*_path.txt), the survey's cross-check constants, an independent
50-digit mpmath formulation (tests/golden/mp_vectors.json) and the reference's documented quirks.
No GPU needed."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, OBJECTS, load_cfg, load_path_rows

# SURVEY.md Appendix A (computed at survey time by an independent numpy restatement)
APPENDIX_A = {
    "Wine_Bottle": dict(
        init=[[-0.998943209742, -0.034359631195, 0.030526700683, -0.088290011624],
              [-0.035138089435, 0.999061127784, -0.025341223765, -0.005935844107],
              [-0.029627324909, -0.026387093345, -0.999212661511, 0.215053361662]],
        tcp1=(0.455391283, 0.115767522, 1.398953451), tcp2=(0.450292705, -0.101468615, 1.316123660)),
    "dumbbell": dict(
        init=[[-0.998368700754, -0.054400280475, -0.017336286774, 0.593535975762],
              [0.054650289716, -0.998403310082, -0.014289025522, -0.013928001854],
              [-0.016531279103, -0.015213148940, 0.999747606604, 0.004071953333]],
        tcp1=(0.353088529, 0.301572832, 1.380679233), tcp2=(0.946793469, 0.304356053, 1.382170216)),
    "stefan": dict(
        init=[[-0.974161128992, 0.225725015241, -0.007636246219, 0.292056451288],
              [-0.225565471038, -0.974068662083, -0.017619870151, 0.381692128649],
              [-0.011415473597, -0.015442119124, 0.999815596958, 0.007945445027]],
        tcp1=(0.315231377, 0.223412581, 1.593750666), tcp2=(0.777877451, 0.100250166, 1.636629704)),
}
# rows of the recorded paths that are TRAC-IK vertices (off the manifold by ~1e-2)
IK_ROWS = {"Wine_Bottle": {6, 7, 13, 14, 21, 22, 29}, "dumbbell": {4, 5, 8}}


@pytest.fixture(scope="module", params=["det", "libm"])
def orc(request, oracle_det, oracle_libm):
    return oracle_det if request.param == "det" else oracle_libm


@pytest.mark.parametrize("obj", OBJECTS)
def test_init_chain_and_tcp_match_survey(orc, obj):
    cfg = load_cfg(obj)
    P = orc.problem(cfg)
    A = np.array(APPENDIX_A[obj]["init"])
    assert np.allclose(np.array(P.init_R).reshape(3, 3), A[:, :3], atol=2e-12)
    assert np.allclose(np.array(P.init_p), A[:, 3], atol=2e-12)
    _, p1 = orc.fk(P, 0, cfg["start_joint"][:7])
    _, p2 = orc.fk(P, 1, cfg["start_joint"][7:])
    assert np.allclose(p1, APPENDIX_A[obj]["tcp1"], atol=1e-9) and np.allclose(p2, APPENDIX_A[obj]["tcp2"], atol=1e-9)
    f = orc.function(P, cfg["start_joint"])
    assert f[0] < 1e-12 and f[1] < 1e-7  # the start state defines the manifold


def test_wine_bottle_pose_comments(orc):
    """config/Wine_Bottle.yaml:21-22 (MoveIt targets used to pick start_joint): ~5 mm agreement."""
    cfg = load_cfg("Wine_Bottle")
    P = orc.problem(cfg)
    _, p1 = orc.fk(P, 0, cfg["start_joint"][:7])
    _, p2 = orc.fk(P, 1, cfg["start_joint"][7:])
    assert np.abs(p1 - [0.45, 0.11, 1.40]).max() < 7e-3 and np.abs(p2 - [0.45, -0.10, 1.32]).max() < 7e-3


@pytest.mark.parametrize("obj", ["Wine_Bottle", "dumbbell"])
def test_recorded_paths_sit_just_under_tolerance(orc, obj):
    """Known-answer test for FK + residual + tolerances: every geodesic row of the reference's
    recorded output has f0 in [7e-4, 1e-3], f1 <= 5e-3 — the signature of the 0.30-step Newton
    stopping the first time both tolerances hold.  (Rows carry 6 significant digits: +-2e-5.)"""
    P = orc.problem(load_cfg(obj))
    rows = load_path_rows(obj)
    f = np.array([orc.function(P, r) for r in rows])
    assert f[0, 0] < 1e-5 and f[0, 1] < 1e-5  # row 1 = start_joint, rounding only
    geo = [i for i in range(2, len(rows)) if i not in IK_ROWS[obj]]
    assert len(geo) >= 4
    assert (f[geo, 0] <= 1e-3 + 2e-5).all() and (f[geo, 0] >= 7e-4).all()
    assert (f[geo, 1] <= 5e-3 + 5e-5).all()
    ik = sorted(IK_ROWS[obj])
    assert (f[ik, 0] > 5e-3).all()  # IK vertices are NOT on the manifold to tolerance
    if obj == "Wine_Bottle":  # consecutive geodesic states are delta = 0.25 apart (ConstrainedPlanningCommon.cpp:118)
        steps = [np.linalg.norm(rows[i + 1] - rows[i]) for i in (2, 3, 4)]
        assert all(0.245 < s < 0.255 for s in steps)


def test_against_mpmath_formulation(orc):
    """Independent DH-matrix / trace-angle formulation at 50 digits (tests/golden/make_mp_vectors.py)."""
    V = json.load(open(os.path.join(GOLDEN, "mp_vectors.json")))
    for obj, d in V.items():
        P = orc.problem(load_cfg(obj))
        assert list(P.arm_index) == d["arm_index"]
        assert np.abs(np.array(P.init_R) - d["init_R"]).max() < 5e-15
        for c in d["cases"]:
            R1, p1 = orc.fk(P, 0, c["x"][:7])
            R2, p2 = orc.fk(P, 1, c["x"][7:])
            assert np.abs(R1.ravel() - c["R1"]).max() < 5e-15 and np.abs(p1 - c["p1"]).max() < 5e-15
            assert np.abs(R2.ravel() - c["R2"]).max() < 5e-15 and np.abs(p2 - c["p2"]).max() < 5e-15
            assert np.abs(orc.function(P, c["x"]) - c["f"]).max() < 1e-13


def test_arm_order_is_alphabetical(oracle_det):
    """std::map iteration order (ConstrainedPlanningCommon.cpp:13-14,89-91): swapping arm1/arm2 in
    the YAML must not change the problem."""
    cfg = load_cfg("dumbbell")
    P1 = oracle_det.problem(cfg)
    swapped = dict(cfg, arm1=cfg["arm2"], arm2=cfg["arm1"])
    P2 = oracle_det.problem(swapped)
    assert bytes(P1) == bytes(P2) and list(P1.arm_index) == [0, 2]


def test_fd_jacobian_matches_analytic(oracle_det):
    P = oracle_det.problem(load_cfg("Wine_Bottle"))
    worst = 0.0
    for i in range(40):
        x = oracle_det.ambient_uniform(P, 11, i)
        Jf, Ja, Jw = oracle_det.jacobian(P, x), oracle_det.jacobian(P, x, analytic=True), oracle_det.jacobian(P, x, analytic="world")
        worst = max(worst, np.abs(Jf - Ja).max())
        assert np.abs(Ja - Jw).max() < 5e-13  # base-frame (kernel order) and world-frame formulations of the same derivative
    assert worst < 5e-7  # FD noise ~ eps/h


def test_minnorm_solve_is_pseudoinverse(oracle_det):
    rng = np.random.default_rng(3)
    for _ in range(50):
        J, f = rng.standard_normal((2, 14)), rng.standard_normal(2)
        assert np.allclose(oracle_det.solve_minnorm(J, f), np.linalg.pinv(J) @ f, rtol=1e-11, atol=1e-13)
    # rank deficient: parallel rows -> minimum-norm least squares, as Eigen's thresholded SVD solve
    r = rng.standard_normal(14)
    J = np.vstack([r, 2 * r])
    f = np.array([1.0, 1.0])
    assert np.allclose(oracle_det.solve_minnorm(J, f), np.linalg.pinv(J) @ f, rtol=1e-9, atol=1e-12)
    assert np.array_equal(oracle_det.solve_minnorm(np.zeros((2, 14)), f), np.zeros(14))


def test_project_semantics_and_quirks(oracle_det):
    cfg = load_cfg("Wine_Bottle")
    P = oracle_det.problem(cfg)
    x0 = np.array(cfg["start_joint"])
    ok, x, it = oracle_det.project(P, x0)
    assert ok and it == 0 and np.array_equal(x, x0)  # already satisfied: zero iterations, untouched
    stats = []
    for i in range(64):
        q = oracle_det.ambient_uniform(P, 0xC2, i)
        ok, x, it = oracle_det.project(P, q)
        f = oracle_det.function(P, x)
        stats.append((ok, it))
        assert it <= 250
        if it < 250:  # exited through the residual test
            assert f[0] <= P.tol_pos and f[1] <= P.tol_rot
            assert ok == (oracle_det.joint_valid(P, x) and f[1] < P.tol_rot)  # strict < in the return (:75)
        else:
            assert not ok or (f[0] <= P.tol_pos and f[1] < P.tol_rot)
        assert oracle_det.is_satisfied(P, x) == (f[0] <= P.tol_pos and f[1] <= P.tol_rot)
    its = np.array([s[1] for s in stats])
    assert 25 < its.mean() < 42  # mean ~33 for Wine_Bottle
    # the iteration cap: with max_iter = 3 the iterate after exactly 3 updates is returned, ok false
    P3 = oracle_det.problem(cfg)
    P3.max_iter = 3
    ok, x, it = oracle_det.project(P3, oracle_det.ambient_uniform(P, 0xC2, 0))
    assert (not ok) and it == 3


def test_joint_valid_margin(oracle_det):
    P = oracle_det.problem(load_cfg("Wine_Bottle"))
    mid = np.array([(P.lb[j % 7] + P.ub[j % 7]) / 2 for j in range(14)])
    assert oracle_det.joint_valid(P, mid)
    for j in (0, 3, 5, 7 + 6):
        x = mid.copy(); x[j] = P.lb[j % 7] + 0.0009
        assert not oracle_det.joint_valid(P, x)
        x[j] = P.lb[j % 7] + 0.0011
        assert oracle_det.joint_valid(P, x)
        x[j] = P.ub[j % 7] - 0.0009
        assert not oracle_det.joint_valid(P, x)


def test_enforce_bounds_wraps_not_clamps(oracle_det):
    x = np.zeros(14)
    x[5] = 3.5          # inside joint 6's range (-0.0175, 3.7525) but > pi: wrapped to a negative angle
    x[0] = -3.5
    x[1] = np.pi        # v >= pi -> v - 2pi
    x[2] = 7.0
    y = oracle_det.enforce_bounds(x)
    assert y[5] == 3.5 - 2 * np.pi and y[0] == -3.5 + 2 * np.pi and y[1] == np.pi - 2 * np.pi
    assert y[2] == np.fmod(7.0, 2 * np.pi) and (np.abs(y) <= np.pi).all()


def test_interpolate_and_distance(oracle_det):
    a, b = np.zeros(14), np.zeros(14)
    a[0], b[0] = 3.0, -3.0     # |diff| > pi: shortest arc through +-pi
    a[1], b[1] = 0.5, 1.5
    m = oracle_det.interpolate(a, b, 0.5)
    assert m[1] == 1.0
    assert abs(abs(m[0]) - np.pi) < 0.15 + 1e-12  # half-way along the short arc (length 2pi-6)
    assert oracle_det.distance(a, b) == np.sqrt(36.0 + 1.0)  # plain Euclidean, not wrapped


def test_discrete_geodesic_properties(oracle_det):
    cfg = load_cfg("Wine_Bottle")
    P = oracle_det.problem(cfg)
    rows = load_path_rows("Wine_Bottle")
    a, b = np.array(cfg["start_joint"]), rows[5]
    ok, states, its = oracle_det.discrete_geodesic(P, a, b, interpolate=True)
    assert np.array_equal(states[0], a)
    for s in states[1:]:
        assert oracle_det.is_satisfied(P, s)
    steps = [oracle_det.distance(states[i], states[i + 1]) for i in range(len(states) - 1)]
    assert all(s <= P.lambda_ * P.delta for s in steps)
    d = [oracle_det.distance(s, b) for s in states]
    assert all(d[i + 1] < d[i] for i in range(len(d) - 1))  # strictly closer each accepted step
    if ok:
        assert d[-1] <= P.delta
    # from == to (or within delta): true immediately, only `from` in the list
    ok2, st2, _ = oracle_det.discrete_geodesic(P, a, a + 0.01)
    assert ok2 and len(st2) == 1


@pytest.mark.parametrize("obj", ["Wine_Bottle", "dumbbell"])
def test_recorded_roadmap_edges_are_traversable(oracle_det, obj):
    """The reference's dumped roadmap (debug/<obj>_node_info.graphml): node 0 is start_joint; the other milestones are
    IK vertices (off the manifold by ~1e-2, like the IK rows of the recorded paths) or projector outputs (just under
    tolerance); every recorded connection was accepted by the reference's checkMotion (discreteGeodesic + collision),
    so discreteGeodesic without the collision test must reach the target at least one way round."""
    from conftest import load_roadmap

    cfg = load_cfg(obj)
    P = oracle_det.problem(cfg)
    nodes, edges = load_roadmap(obj)
    f = np.array([oracle_det.function(P, n) for n in nodes])
    assert np.abs(nodes[0] - np.array(cfg["start_joint"])).max() < 1e-5 and f[0].max() < 1e-5
    on = [i for i in range(1, len(nodes)) if oracle_det.is_satisfied(P, nodes[i])]
    off = [i for i in range(1, len(nodes)) if i not in on]
    for i in on:   # stopped by the Newton loop the first time both tolerances held
        assert 7e-4 <= f[i, 0] <= 1e-3 + 2e-5 and f[i, 1] <= 5e-3 + 5e-5
    for i in off:  # IK milestones
        assert 5e-3 < f[i, 0] < 2.5e-2 and f[i, 1] < 8e-2
    assert len(on) == (2 if obj == "Wine_Bottle" else 0)
    pairs = sorted({(min(a, b), max(a, b)) for a, b in edges})
    assert len(edges) == 2 * len(pairs)  # stored as two directed edges per connection
    both = 0
    for a, b in pairs:
        ok_ab, st_ab, _ = oracle_det.discrete_geodesic(P, nodes[a], nodes[b], interpolate=True)
        ok_ba, st_ba, _ = oracle_det.discrete_geodesic(P, nodes[b], nodes[a], interpolate=True)
        assert ok_ab or ok_ba, (a, b)
        both += int(ok_ab and ok_ba)
        for st in (st_ab, st_ba):  # consecutive states at most lambda*delta apart, about dist/delta of them
            assert all(oracle_det.distance(st[i], st[i + 1]) <= P.lambda_ * P.delta for i in range(len(st) - 1))
    print("%s: %d connections, %d traversable both ways" % (obj, len(pairs), both))
    assert both >= len(pairs) - 1


def test_sampler_is_counter_based(oracle_det):
    P = oracle_det.problem(load_cfg("stefan"))
    a = oracle_det.ambient_uniform_batch(P, 9, 100, 8)
    b = oracle_det.ambient_uniform_batch(P, 9, 104, 4)
    assert np.array_equal(a[4:], b)  # sample i depends on (seed, i) only: shards reproduce the whole
    lb, ub = np.array(P.lb[:]), np.array(P.ub[:])
    assert (a[:, :7] >= lb).all() and (a[:, :7] <= ub).all() and (a[:, 7:] >= lb).all() and (a[:, 7:] <= ub).all()
    assert oracle_det.lib.orc_splitmix64(0) == 0xE220A8397B1DCDAF  # published SplitMix64 first output for seed 0


def test_libm_and_det_builds_agree_where_the_algorithm_is_well_conditioned(oracle_det, oracle_libm):
    """Same algorithm, two rounding models: residuals agree to ~1e-15; the Newton iterates do NOT
    stay within 1e-6 rad (the iteration amplifies ulp differences) — the reason GPU parity is
    defined bitwise against the det build."""
    cfg = load_cfg("Wine_Bottle")
    Pd, Pl = oracle_det.problem(cfg), oracle_libm.problem(cfg)
    q = oracle_det.ambient_uniform_batch(Pd, 0xC2, 0, 192)
    fd, fl = oracle_det.function_batch(Pd, q, 4), oracle_libm.function_batch(Pl, q, 4)
    assert np.abs(fd - fl).max() < 1e-13
    qd, okd, itd = oracle_det.project_batch(Pd, q, 8)
    ql, okl, itl = oracle_libm.project_batch(Pl, q, 8)
    d = np.abs(qd - ql).max(axis=1)
    assert np.median(d) < 1e-6          # most samples track each other...
    assert (d > 1e-6).mean() > 0.03     # ...but a sizeable fraction does not, by construction of the reference
    assert abs(okd.mean() - okl.mean()) < 0.05


def test_reference_iteration_amplifies_one_ulp(oracle_libm):
    """What "matches the reference to 1e-6 rad" can and cannot mean.  The glibc
    build of the restatement — the closest thing to a stock build of the reference — against ITSELF, its 4 096 uniform
    Wine_Bottle inputs moved by ONE ulp: about a fifth of the projections end more than 1e-6 rad away, some by
    milliradians, and thousands stop at another iteration.  The reference's FD-Newton from a uniform sample
    (ConstraintFunction.h:68-72: sqrt(eps) stencil, norm-valued residuals, ~33 iterations) cannot be reproduced to
    1e-6 rad by anything that does not reproduce it bit for bit — the reason GPU parity is defined bitwise against the
    det build.  The exact Jacobian in the same loop is far better conditioned (a twentieth instead of a fifth)."""
    cfg = load_cfg("Wine_Bottle")
    P = oracle_libm.problem(cfg)
    q = oracle_libm.ambient_uniform_batch(P, 0xC3, 0, 4096)
    q1 = np.nextafter(q, np.inf)  # one ulp, every joint
    nthr = min(8, os.cpu_count() or 1)
    a, oka, ita = oracle_libm.project_batch(P, q, nthr)
    b, okb, itb = oracle_libm.project_batch(P, q1, nthr)
    d = np.abs(a - b).max(axis=1)
    frac = float((d > 1e-6).mean())
    assert 0.15 < frac < 0.27, frac                      # measured 0.207
    assert d.max() > 1e-3                                # milliradians on the worst samples (measured 6.9e-3)
    assert np.median(d) < 1e-6                           # the typical sample does track
    assert (np.abs(ita - itb) >= 1).mean() > 0.05        # the discrete stopping rule flips
    assert abs(oka.mean() - okb.mean()) < 0.02           # the statistics are the same: both answers are "the" projection
    Pa = oracle_libm.problem(cfg)
    Pa.jacobian_mode = 1
    aa, _, _ = oracle_libm.project_batch(Pa, q, nthr)
    ba, _, _ = oracle_libm.project_batch(Pa, q1, nthr)
    frac_a = float((np.abs(aa - ba).max(axis=1) > 1e-6).mean())
    assert 0.02 < frac_a < 0.08 and frac_a < 0.4 * frac, (frac_a, frac)  # measured 0.044


# ---- trajectory-level pin: the reference's recorded paths ARE outputs of its discreteGeodesic ------------------------
# `path.interpolate()` (src/base/constraints/ConstrainedPlanningCommon.cpp:217) replaces every solution segment by the
# states of jy_ProjectedStateSpace::discreteGeodesic(s1, s2, interpolate = true) (OMPL's ConstrainedSpaceInformation::
# getMotionStates), and `printAsMatrix` (:219-222) dumps them with 6 significant digits.  Between two repeated rows
# (s1 == `from`, the first geodesic state) the file therefore records — digit for digit — what the real RBDL / Eigen /
# OMPL build of project() + discreteGeodesic produced: interpolate, FD Jacobian, SVD solve, 0.30 step, stop rule, the
# four break tests.  A stop one Newton iteration early or late would move a state by ~1e-3.
RECORDED_SEGMENTS = {
    # obj: (delta the file was recorded with, [(a, b)]: rows a..b-1 are the geodesic from rows[a] towards rows[b])
    "Wine_Bottle": (0.25, [(1, 6), (7, 13), (14, 21), (22, 29)]),
    # the dumbbell run used delta = 0.5 (consecutive rows are 0.50 apart; with 0.25 the state COUNT is already wrong:
    # 5 and 6 states instead of the recorded 3 and 3); every other delta in 0.4..1.0 misses by > 6e-2
    "dumbbell": (0.5, [(1, 4), (5, 8)]),
}


def print_precision(v):
    """half a unit of the 6th significant digit of the C++ default stream format"""
    v = np.abs(np.asarray(v, dtype=np.float64))
    e = np.floor(np.log10(np.where(v > 0, v, 1.0)))
    return 0.5 * 10.0 ** (e - 5)


@pytest.mark.parametrize("obj", sorted(RECORDED_SEGMENTS))
def test_recorded_paths_are_reproduced(orc, obj):
    """Both oracle builds return exactly the recorded number of states for every recorded segment, and every state
    equals the recorded row to the print precision of the file (Wine_Bottle: <= 1.2e-5 rad, measured 4e-6..8.4e-6)."""
    P = orc.problem(load_cfg(obj))
    delta, segs = RECORDED_SEGMENTS[obj]
    P.delta = delta
    rows = load_path_rows(obj)
    rng = np.random.default_rng(17)
    worst = 0.0
    for a, b in segs:
        ok, st, its = orc.discrete_geodesic(P, rows[a], rows[b], interpolate=True)
        assert len(st) == b - a, (obj, a, b, len(st))
        err = np.abs(st - rows[a:b]).max(axis=1)
        worst = max(worst, float(err.max()))
        if obj == "Wine_Bottle":
            assert err.max() <= 1.2e-5, (a, b, err)
        else:
            # delta = 0.5 starts each projection further from the manifold (~22 Newton iterations per state instead
            # of ~8), and the iteration multiplies the +-5e-6 rounding of the printed endpoints by 1.4-2 per
            # iteration in the tangent direction: compare with the spread that endpoints consistent
            # with the printed digits produce
            spread = np.zeros(len(st))
            for _ in range(24):
                fa = rows[a] + rng.uniform(-1, 1, 14) * print_precision(rows[a])
                fb = rows[b] + rng.uniform(-1, 1, 14) * print_precision(rows[b])
                _, st2, _ = orc.discrete_geodesic(P, fa, fb, interpolate=True)
                assert len(st2) == len(st)
                spread = np.maximum(spread, np.abs(st2 - st).max(axis=1))
            assert (err <= 1.5 * spread + 1e-5).all() and err.max() <= 3e-4, (a, b, err, spread)
        # the segment ends where the reference's loop ended: the next step would be the target itself
        assert orc.distance(st[-1], rows[b]) <= 2 * delta
    print("%s (%s build): recorded geodesic rows reproduced, max |dq| = %.2e rad" % (obj, orc.kind, worst))


def test_recorded_path_rejects_a_wrong_projector(oracle_det):
    """The pin has teeth: a Newton step of 0.25 or 0.35 instead of the reference's 0.30, a looser tolerance, or one
    iteration more or less, all leave the recorded rows by far more than the print precision."""
    cfg = load_cfg("Wine_Bottle")
    rows = load_path_rows("Wine_Bottle")

    def worst(mut):
        P = oracle_det.problem(cfg)
        mut(P)
        w = 0.0
        for a, b in RECORDED_SEGMENTS["Wine_Bottle"][1]:
            _, st, _ = oracle_det.discrete_geodesic(P, rows[a], rows[b], interpolate=True)
            m = min(len(st), b - a)
            w = max(w, float(np.abs(st[:m] - rows[a:a + m]).max()), 1.0 if len(st) != b - a else 0.0)
        return w

    assert worst(lambda P: None) <= 1.2e-5
    assert worst(lambda P: setattr(P, "step", 0.25)) > 1e-4
    assert worst(lambda P: setattr(P, "step", 0.35)) > 1e-4
    assert worst(lambda P: setattr(P, "tol_pos", 1.2e-3)) > 1e-4
    assert worst(lambda P: setattr(P, "tol_rot", 2.5e-3)) > 1e-4
    assert worst(lambda P: setattr(P, "delta", 0.26)) > 1e-3


# What the recorded paths can and cannot tell apart.  The reference calls OMPL's default
# finite-difference `jacobian` and Eigen's `JacobiSVD(...).solve` / `angularDistance` (ConstraintFunction.h:70-71,94-97);
# the oracle restates the upstream algorithms.  Each variant below replaces one of those third-party choices by another
# mathematically consistent one; the table says whether 6 printed digits of ~8-iteration projections notice.
RESOLVING_POWER = [
    # (name, how to switch it on, expected on Wine_Bottle): "rejected" = a state count differs or a state leaves the
    # recorded row by more than the print-precision bound 1.2e-5 rad; "indistinguishable" = neither
    ("restatement (7-point stencil, min-norm SVD solve, 2 atan2)", None, "indistinguishable"),
    ("exact (analytic) Jacobian instead of the FD stencil", ("mode", 1), "indistinguishable"),
    ("3-point central difference instead of OMPL's 7-point stencil", ("stencil", 1), "indistinguishable"),
    ("h = 1e-6 instead of sqrt(DBL_EPSILON)", ("h", 1), "indistinguishable"),
    ("normal equations J^T (J J^T)^-1 f instead of the thresholded SVD", ("solve", 1), "indistinguishable"),
    ("Eigen 3.2 angularDistance 2 acos|a.b| instead of 2 atan2(|vec|, |w|)", ("angle", 1), "indistinguishable"),
    ("return f0 <= tol1 && f1 <= tol2 instead of the norm1 / norm2 quirk", ("return", 1), "indistinguishable"),
    ("damped least squares (J J^T + 1e-4 I) — north_star's wording, not the reference's code", ("solve", 2), "rejected"),
    ("Newton step 0.25 instead of 0.30", ("attr", "step", 0.25), "rejected"),
    ("tolerance 1.2e-3 m instead of 1e-3", ("attr", "tol_pos", 1.2e-3), "rejected"),
]


def recorded_path_error(orc, obj, how):
    """(max |state - recorded row| over every recorded segment, all state counts equal the recorded ones)"""
    import contextlib

    P = orc.problem(load_cfg(obj))
    delta, segs = RECORDED_SEGMENTS[obj]
    P.delta = delta
    rows = load_path_rows(obj)
    cm = contextlib.nullcontext()
    if how is not None:
        if how[0] == "mode":
            P.jacobian_mode = how[1]
        elif how[0] == "attr":
            setattr(P, how[1], how[2])
        else:
            cm = orc.variant(how[0], how[1])
    worst, counts = 0.0, True
    with cm:
        for a, b in segs:
            _, st, _ = orc.discrete_geodesic(P, rows[a], rows[b], interpolate=True)
            m = min(len(st), b - a)
            counts = counts and len(st) == b - a
            worst = max(worst, float(np.abs(st[:m] - rows[a:a + m]).max()))
    return worst, counts


def test_recorded_paths_resolving_power(orc):
    """The recorded paths pin FK, residual, tolerances, the 0.30 step, the stop rule, `interpolate`, the break tests and
    delta — NOT the choice of Jacobian (FD stencil vs exact), of linear solve or of angle formula: those variants
    reproduce every recorded row exactly as well as the restatement does (8.37e-6 rad on Wine_Bottle, the rounding of the
    printed digits).  a3 / a4 therefore rest on restating upstream OMPL `Constraint::jacobian` and Eigen
    `JacobiSVD::solve`, not on a reference artefact (DESIGN.md §3 carries this table with the measured figures)."""
    print()
    for name, how, expect in RESOLVING_POWER:
        w, counts = recorded_path_error(orc, "Wine_Bottle", how)
        wd, cd = recorded_path_error(orc, "dumbbell", how)
        got = "indistinguishable" if (counts and w <= 1.2e-5) else "rejected"
        print("%-90s Wine_Bottle %.3e%s  dumbbell %.3e%s  -> %s" % (name, w, "" if counts else " (count!)", wd, "" if cd else " (count!)", got))
        assert got == expect, (name, w, counts)
        if expect == "indistinguishable":
            # and not by a margin either: within 1 % of the restatement's own distance from the printed rows
            assert abs(w - 8.3667e-6) < 1e-7 and cd and wd < 3e-4
    assert all(orc.lib.orc_get_variant(k) == 0 for k in range(5))  # every variant switched off again


def ulp_sensitivity(orc, P, frm, to, states, n, maxs, nthreads):
    """How far the oracle's OWN geodesics move when the endpoints are perturbed in their last bit (4 draws of
    +-4e-16 relative): the conditioning of each edge, independent of any second implementation"""
    r = np.random.default_rng(1)
    out = np.zeros(len(frm))
    for _ in range(4):
        f2 = frm * (1 + r.uniform(-1, 1, frm.shape) * 4e-16)
        t2 = to * (1 + r.uniform(-1, 1, to.shape) * 4e-16)
        s2, n2, _, _ = orc.discrete_geodesic_batch(P, f2, t2, maxs, nthreads)
        for j in range(len(frm)):
            out[j] = max(out[j], np.abs(s2[j, : n[j]] - states[j, : n[j]]).max() if n2[j] == n[j] else 1.0)
    return out


@pytest.mark.parametrize("obj", ["Wine_Bottle", "dumbbell"])
def test_det_and_libm_builds_on_the_recorded_extend_workload(oracle_det, oracle_libm, obj):
    """Every directed edge of the reference's dumped roadmap through both rounding models.  Wine_Bottle — near-manifold
    projections of ~8 Newton iterations per state — stays together: same counts, flags and iteration totals, states
    within 1e-6 rad (measured 7e-9), so north_star's bar is meetable against a glibc projector on that extend workload.
    dumbbell (arms stretched 0.59 m apart, ~20 iterations per state) does not: its edges differ by 2e-5..4e-4 rad
    between the two libms and every one of them is ill-conditioned — a last-bit change of the endpoints moves the det
    build's OWN result by more than 1e-7 rad — which is why GPU parity is defined bitwise."""
    from conftest import NCPU, load_roadmap

    cfg = load_cfg(obj)
    Pd, Pl = oracle_det.problem(cfg), oracle_libm.problem(cfg)
    nodes, edges = load_roadmap(obj)
    frm = np.array([nodes[a] for a, _ in edges])
    to = np.array([nodes[b] for _, b in edges])
    sd, nd, okd, itd = oracle_det.discrete_geodesic_batch(Pd, frm, to, 64, NCPU)
    sl, nl, okl, itl = oracle_libm.discrete_geodesic_batch(Pl, frm, to, 64, NCPU)
    assert np.array_equal(nd, nl) and np.array_equal(okd, okl)
    d = np.array([np.abs(sd[e, : nd[e]] - sl[e, : nd[e]]).max() for e in range(len(edges))])
    if obj == "Wine_Bottle":
        assert np.array_equal(itd, itl) and d.max() <= 1e-6
    else:
        out = np.where(d > 1e-6)[0]
        sens = ulp_sensitivity(oracle_det, Pd, frm[out], to[out], sd[out], nd[out], 64, NCPU)
        assert (sens > 1e-7).all() and d.max() < 2e-3
    print("%s: %d directed edges, %d states, %.1f Newton iterations per projected state, det vs libm max |dq| = %.1e rad"
          % (obj, len(edges), int(nd.sum()), itd.sum() / max(1, (nd - 1).sum()), d.max()))


def test_geodesic_reports_a_list_that_does_not_fit(oracle_det):
    """a traversal that needs more than max_states entries stops there and reports max_states + 1 with false (a cut list
    must never look complete, and a creeping edge must not run on unbounded); the binding re-runs with room"""
    cfg = load_cfg("Wine_Bottle")
    P = oracle_det.problem(cfg)
    rows = load_path_rows("Wine_Bottle")
    full_ok, full, _ = oracle_det.discrete_geodesic(P, rows[14], rows[21], interpolate=True, max_states=64)
    st, n, ok, _ = oracle_det.discrete_geodesic_batch(P, rows[14:15], rows[21:22], 3, 1)
    assert len(full) == 7 and n[0] == 4 and ok[0] == 0 and np.array_equal(st[0, :3], full[:3])
    st, n, ok, _ = oracle_det.discrete_geodesic_batch(P, rows[14:15], rows[21:22], 7, 1)  # exactly enough room
    assert n[0] == 7 and bool(ok[0]) == full_ok and np.array_equal(st[0], full)
    ok2, st2, _ = oracle_det.discrete_geodesic(P, rows[14], rows[21], interpolate=True, max_states=2)  # binding re-runs
    assert ok2 == full_ok and np.array_equal(st2, full)


def test_a_continued_geodesic_is_the_uninterrupted_one(oracle_det):
    """the resumable form (ccmp_geodesic_batch_ex / orc_discrete_geodesic_ex): a traversal cut into lists of 2, 3 or 5
    states and continued from each list's last state gives the states, flag and Newton count of the uninterrupted
    traversal, bit for bit — on every recorded segment and on every directed edge of the dumped Wine_Bottle roadmap"""
    from conftest import load_roadmap

    cfg = load_cfg("Wine_Bottle")
    P = oracle_det.problem(cfg)
    rows = load_path_rows("Wine_Bottle")
    nodes, edges = load_roadmap("Wine_Bottle")
    pairs = [(rows[a], rows[b]) for a, b in RECORDED_SEGMENTS["Wine_Bottle"][1]] + [(nodes[a], nodes[b]) for a, b in edges]
    cuts = 0
    for frm, to in pairs:
        full_ok, full, full_its = oracle_det.discrete_geodesic(P, frm, to, interpolate=True)
        for cap in (2, 3, 5):
            ok, st, n, its, carry = oracle_det.discrete_geodesic_ex(P, frm, to, cap)
            parts, total_its = [st], its
            while n > cap:
                cuts += 1
                ok, st, n, its, carry = oracle_det.discrete_geodesic_ex(P, parts[-1][-1], to, cap, carry_in=carry)
                parts.append(st[1:])
                total_its += its
            got = np.concatenate(parts, axis=0)
            assert got.shape == full.shape and np.array_equal(got.view(np.uint64), full.view(np.uint64))
            assert ok == full_ok and total_its == full_its
    assert cuts > 50


def test_gram_step_is_the_minimum_norm_solution(oracle_det, oracle_libm):
    """orc_solve_gram — the analytic mode's step, SURVEY.md §7.3: dx = J^T (J J^T)^-1 f through the 2x2 Gram matrix in closed form —
    against numpy's minimum-norm least-squares solution and against the SVD-equivalent routine of the reference arithmetic
    (orc_solve_minnorm = Eigen's JacobiSVD.solve, ConstraintFunction.h:71) on Jacobians of the real problem: the same solution to
    1e-10 relative wherever the two rows are not nearly parallel; and where they are (sin^2 of their angle <= 2^-20), or the matrix
    is rank-deficient, or something is NaN, it IS orc_solve_minnorm, bit for bit — rank handling stays Eigen's."""
    for O in (oracle_det, oracle_libm):
        P = O.problem(load_cfg("Wine_Bottle"))
        rng = np.random.default_rng(11)
        worst = 0.0
        for k in range(200):
            x = O.ambient_uniform(P, 0x6A, k)
            J = O.jacobian(P, x, analytic=True)
            f = O.function(P, x)
            dx = O.solve_gram(J, f)
            ref = np.linalg.lstsq(J, f, rcond=None)[0]
            worst = max(worst, np.abs(dx - ref).max() / np.abs(ref).max())
            assert np.abs(dx - O.solve_minnorm(J, f)).max() <= 1e-10 * np.abs(ref).max()
        assert worst < 1e-10, worst
        # nearly parallel rows, exactly parallel rows, a zero row, a NaN: the fallback, bit for bit
        base = rng.standard_normal(14)
        for J in (np.stack([base, base * (1 + 1e-9) + 1e-9 * rng.standard_normal(14)]), np.stack([base, 2.0 * base]), np.stack([base, np.zeros(14)]),
                  np.stack([base, np.where(np.arange(14) == 3, np.nan, base)])):
            f = np.array([0.3, 0.2])
            a, b = O.solve_gram(J, f), O.solve_minnorm(J, f)
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64)) or (np.isnan(a).all() and np.isnan(b).all()), J
