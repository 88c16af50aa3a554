"""Host side of libccmp without a GPU: the library builds, loads, exports every symbol
include/ccmp.h declares, reads the reference's YAML, and its set-up arithmetic is bit-identical to
the oracle's.  No compute entry point is called successfully here (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import OBJECTS, ROOT, config_path, load_cfg


@pytest.fixture(scope="module")
def L(ccmp_built):
    from closed_chain_motion_planner_amd import _lib

    return _lib.lib()


def test_exports_every_declared_symbol(L, ccmp_built):
    hdr = open(os.path.join(ROOT, "include", "ccmp.h")).read()
    declared = set(re.findall(r"\b(ccmp_[a-z0-9_]+)\s*\(", hdr))
    from closed_chain_motion_planner_amd import _lib

    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.ccmp_version() == 600
    # the test / tool hooks live in include/ccmp_debug.h and in lib/libccmp_debug.so only: the product library exports none of them
    # (nor any other symbol that says debug / probe / experimental), and knows no debug option
    import subprocess

    dbg_hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ccmp_debug.h")).read(), flags=re.S)
    dbg_declared = set(re.findall(r"\b(ccmp_[a-z0-9_]+)\s*\(", dbg_hdr))
    assert dbg_declared == set(_lib.DEBUG_EXPORTS) and not (dbg_declared & declared)
    syms = lambda path: set(re.findall(r" T (\w+)", subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout))
    default_syms, debug_syms = syms(ccmp_built), syms(_lib.DEBUG_LIBPATH)
    assert not [n for n in default_syms if re.search(r"debug|probe|experimental", n)], default_syms
    assert dbg_declared <= debug_syms and declared <= debug_syms and declared <= default_syms
    assert all(o["name"] != "fail_after_fork" and not o["doc"].startswith("debug") for o in _lib.option_table())


def test_library_does_not_link_rccl(ccmp_built):
    """the collective entry points open librccl at run time: planners that never shard pay no dependency"""
    import subprocess

    out = subprocess.run(["ldd", ccmp_built], capture_output=True, text=True).stdout
    assert "rccl" not in out and "nccl" not in out and "libamdhip64" in out


def test_abi_has_no_torch_or_cxx_types(ccmp_built):
    hdr = open(os.path.join(ROOT, "include", "ccmp.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # signatures only, comments stripped
    assert "torch" not in code and "std::" not in code and "hipStream_t" not in code and "Eigen" not in code
    assert 'extern "C"' in hdr


@pytest.mark.parametrize("obj", OBJECTS)
def test_problem_from_yaml_is_bitwise_the_oracle(L, oracle_det, obj):
    from closed_chain_motion_planner_amd import load_config

    P = load_config(config_path(obj))
    Po = oracle_det.problem(load_cfg(obj))
    assert L.ccmp_problem_sizeof() == oracle_det.lib.orc_problem_sizeof()
    assert bytes(P) == bytes(Po)
    assert (P.tol_pos, P.tol_rot, P.step, P.max_iter, P.delta, P.lambda_, P.joint_eps) == (1e-3, 5e-3, 0.30, 250, 0.25, 2.0, 1e-3)


def test_problem_struct_layouts_agree(L):
    from closed_chain_motion_planner_amd._lib import CcmpProblem
    from oracle_binding import OrcProblem

    assert C.sizeof(CcmpProblem) == C.sizeof(OrcProblem)
    for (n1, _), (n2, _) in zip(CcmpProblem._fields_, OrcProblem._fields_):
        assert n1 == n2 and getattr(CcmpProblem, n1).offset == getattr(OrcProblem, n2).offset


def test_yaml_reader_edge_cases(L, tmp_path):
    from closed_chain_motion_planner_amd import CcmpError, load_config

    with pytest.raises(CcmpError) as e:
        load_config(str(tmp_path / "missing.yaml"))
    assert e.value.code == -3  # CCMP_EIO
    bad = tmp_path / "bad.yaml"
    bad.write_text("obj_name: x\nstart_joint: [0, 1, 2]\n")
    with pytest.raises(CcmpError) as e:
        load_config(str(bad))
    assert e.value.code == -4  # CCMP_EPARSE: wrong length / missing keys
    # comments, irregular spacing, list continued on the next line, arms given in reverse order
    cfg = load_cfg("dumbbell")
    txt = ("# header\nobj_name: dumbbell   # trailing comment\nstart_joint: [%s,\n   %s]\n"
           % (" , ".join(repr(v) for v in cfg["start_joint"][:5]), ",".join(repr(v) for v in cfg["start_joint"][5:])))
    txt += "t_wo_start_pos: [0.65, 0.3, 1.28]\nt_wo_start_quat: [0, 0, 0, 1]  # x, y, z, w\n"
    txt += "t_wo_goal_pos: [0.65, 0.0, 1.28]\nt_wo_goal_quat: [0, 0, 0, 1]\n\nmesh_file_: package://x/y.stl\n"
    txt += "arm1:\n  name: panda_top\n  index: 2\narm2:\n  name: panda_left\n  index: 0\n"
    ok = tmp_path / "ok.yaml"
    ok.write_text(txt)
    P = load_config(str(ok))
    assert bytes(P) == bytes(load_config(config_path("dumbbell")))


def test_set_tolerance_rejects_non_positive(L):
    from closed_chain_motion_planner_amd import load_config

    P = load_config(config_path("Wine_Bottle"))
    assert L.ccmp_set_tolerance(C.byref(P), 0.0, 1e-3) == -1
    assert L.ccmp_set_tolerance(C.byref(P), 1e-3, -1.0) == -1
    assert L.ccmp_set_tolerance(C.byref(P), 5e-4, 2.5e-3) == 0 and (P.tol_pos, P.tol_rot) == (5e-4, 2.5e-3)


def test_set_start_moves_the_manifold(L, oracle_det):
    from closed_chain_motion_planner_amd import load_config

    P = load_config(config_path("Wine_Bottle"))
    q0 = np.array(P.start_joint[:]) + 0.05
    assert L.ccmp_set_start(C.byref(P), q0.ctypes.data_as(C.POINTER(C.c_double))) == 0
    Po = oracle_det.problem(dict(load_cfg("Wine_Bottle"), start_joint=list(q0)))
    assert bytes(P) == bytes(Po)


def test_calibration_offsets_change_the_model(L):
    from closed_chain_motion_planner_amd import load_config

    P = load_config(config_path("Wine_Bottle"))
    before = bytes(P)
    dh = (C.c_double * 28)(*([0.0] * 28))
    assert L.ccmp_set_calibration(C.byref(P), 0, dh) == 0 and bytes(P) == before  # zero offsets = shipped model
    dh[2] = 1e-3  # theta offset of joint 1
    assert L.ccmp_set_calibration(C.byref(P), 0, dh) == 0 and bytes(P) != before
    assert L.ccmp_set_calibration(C.byref(P), 2, dh) == -1


def test_no_cpu_fallback_without_a_device(L):
    """On a box without a GPU the context cannot be created; nothing computes on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from closed_chain_motion_planner_amd import CcmpError, Context

    with pytest.raises(CcmpError) as e:
        Context(0)
    assert e.value.code == -5  # CCMP_ENODEV
    h = C.c_void_p()
    assert L.ccmp_project_batch(h, None, None, None, None, None, 4, None) == -1  # NULL ctx: EINVAL, not a host path


def test_product_does_not_touch_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline may include, import, link or call oracle/
    (comments may cite it)"""
    bad = re.compile(r'#\s*include\s*[<"][^>"]*oracle|^\s*(from|import)\s+\S*oracle|\borc_\w+\s*\(|libccmp_oracle', re.M)
    roots = [os.path.join(ROOT, "closed_chain_motion_planner_amd"), os.path.join(ROOT, "include")]
    for root in roots:
        for dirpath, _, files in os.walk(root):
            for fn in files:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    m = bad.search(open(os.path.join(dirpath, fn)).read())
                    assert m is None, (fn, m.group(0))


def test_set_arms_keeps_the_configured_problem(L, oracle_det):
    """setArmModels AFTER loadConfig (the reference's order, ConstrainedPlanningCommon.cpp:126): object poses, t_o7,
    tolerances, delta / lambda and mode survive; choosing other arms changes only what depends on them"""
    from closed_chain_motion_planner_amd import load_config
    from closed_chain_motion_planner_amd._lib import CcmpProblem

    P = load_config(config_path("stefan"))
    P.tol_pos, P.delta, P.jacobian_mode = 5e-4, 0.2, 1
    before = bytes(P)
    assert L.ccmp_set_arms(C.byref(P), b"panda_left", 0, b"panda_top", 2) == 0  # the arms the YAML chose, in its (std::map) order
    assert bytes(P) == before
    assert L.ccmp_set_arms(C.byref(P), b"panda_left", 0, b"panda_right", 1) == 0
    assert list(P.arm_index) == [0, 1] and (P.tol_pos, P.delta, P.jacobian_mode) == (5e-4, 0.2, 1)
    assert bytes(P.obj_start_R) == bytes(CcmpProblem.from_buffer_copy(before).obj_start_R)
    assert bytes(P.t_o7_R)[:72] == before[CcmpProblem.t_o7_R.offset:CcmpProblem.t_o7_R.offset + 72]  # the left arm did not move
    Po = oracle_det.problem(dict(load_cfg("stefan"), arm2={"name": "panda_right", "index": 1}))
    assert bytes(P.init_R) == bytes(Po.init_R) and bytes(P.init_p) == bytes(Po.init_p)  # init_chain_ follows the new arm
    assert L.ccmp_set_arms(C.byref(P), b"panda_left", 0, b"panda_top", 3) == -1


def test_set_arms_stores_the_order_given_and_base_frames_by_value(L, oracle_det):
    """KinematicChainConstraint::setArmModels pushes arm1, then arm2 (ConstraintFunction.h:122-126): no sorting — that is
    ConstrainedProblem::_setEnvironment's std::map (ConstrainedPlanningCommon.cpp:89-91), i.e. ccmp_problem_init /
    _from_yaml here.  ccmp_set_base_frame takes ArmModel::t_wb by value (panda_model.h:15) and reproduces the index
    table's problem byte for byte when given the table's frames."""
    from closed_chain_motion_planner_amd import load_config
    from closed_chain_motion_planner_amd._lib import CcmpProblem

    P = load_config(config_path("stefan"))  # YAML order: panda_left (0), panda_top (2)
    ref = CcmpProblem.from_buffer_copy(bytes(P))
    assert L.ccmp_set_arms(C.byref(P), b"panda_top", 2, b"panda_left", 0) == 0  # the other order is KEPT
    assert list(P.arm_index) == [2, 0]
    assert bytes(P.base_R)[:72] == bytes(ref.base_R)[72:] and bytes(P.base_p)[:24] == bytes(ref.base_p)[24:]
    assert bytes(P.init_p) != bytes(ref.init_p)  # init_chain_ = inverse(second) * first with the bases swapped (its rotation
    # happens to survive: top = diag(-1, -1, 1) is symmetric and left is the identity)
    # frames by value: left = trans(0, 0.3, 1.006), top = trans(1.35, 0.3, 1.006) * Rz(pi) (grasping_point.cpp:11-20)
    Q = load_config(config_path("Wine_Bottle"))  # left + right
    assert L.ccmp_set_arms(C.byref(Q), b"panda_left", 0, b"panda_right", 1) == 0
    I3 = np.eye(3)
    Rz = np.diag([-1.0, -1.0, 1.0])
    dp = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))
    assert L.ccmp_set_base_frame(C.byref(Q), 1, dp(Rz), dp([1.35, 0.3, 1.006])) == 0
    Q.arm_index[1] = 2  # the index is bookkeeping; the frame is what function() multiplies with
    S = load_config(config_path("Wine_Bottle"))
    assert L.ccmp_set_arms(C.byref(S), b"panda_left", 0, b"panda_top", 2) == 0
    assert bytes(Q) == bytes(S)
    assert L.ccmp_set_base_frame(C.byref(Q), 0, dp(I3), dp([0.0, 0.3, 1.006])) == 0 and bytes(Q) == bytes(S)
    # refused: not a rotation, a reflection, non-finite, bad slot
    assert L.ccmp_set_base_frame(C.byref(Q), 0, dp(2 * I3), dp([0, 0, 0])) == -1
    assert L.ccmp_set_base_frame(C.byref(Q), 0, dp(np.diag([1.0, 1.0, -1.0])), dp([0, 0, 0])) == -1
    assert L.ccmp_set_base_frame(C.byref(Q), 0, dp(I3), dp([0, np.nan, 0])) == -1
    assert L.ccmp_set_base_frame(C.byref(Q), 2, dp(I3), dp([0, 0, 0])) == -1
    assert bytes(Q) == bytes(S)


def test_every_sampler_gets_its_own_seed(monkeypatch):
    """each sampler owns an independently seeded stream, as each OMPL sampler owns its ompl::RNG
    (src/base/jy_ProjectedStateSpace.cpp:5-8): two samplers of one space never share a seed, and CCMP_SEED makes the
    sequence reproducible"""
    from closed_chain_motion_planner_amd import space as sp

    class _NoGpu:  # the seed plumbing does not touch the constraint
        ctx = None

    s = sp.jy_ProjectedStateSpace(_NoGpu(), seed=7)
    a, b, c = s.allocStateSampler(), s.allocDefaultStateSampler(), sp.jy_ProjectedStateSampler(_NoGpu())
    assert len({a.seed, b.seed, c.seed}) == 3
    assert a.seed == sp.splitmix64(7) and b.seed == sp.splitmix64(8)
    monkeypatch.setenv("CCMP_SEED", "0x5EED")
    monkeypatch.setattr(sp, "_process_seed", None)
    monkeypatch.setattr(sp, "_seed_counter", __import__("itertools").count())
    assert sp.next_sampler_seed() == sp.splitmix64(0x5EED) and sp.next_sampler_seed() == sp.splitmix64(0x5EEE)
    assert sp.splitmix64(0) == 0xE220A8397B1DCDAF  # published first output of SplitMix64 seeded with 0


# ---- the option table, its getter and the policy's own description (round 5: the header had drifted from the code) ----------------
def test_header_option_table_is_the_librarys(L):
    """the table of defaults and ranges in include/ccmp.h is generated from the library's own (csrc/ccmp_policy.cpp) — and is
    re-generated here: a default stated in the header IS the default in the code"""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_option_docs

    hdr = open(os.path.join(ROOT, "include", "ccmp.h")).read()
    a, b = hdr.index(gen_option_docs.BEGIN), hdr.index(gen_option_docs.END) + len(gen_option_docs.END) + 1
    assert hdr[a:b] == gen_option_docs.block(), "include/ccmp.h is stale: python tools/gen_option_docs.py --write"
    # and nothing outside the table states a number for the two defaults that had drifted
    rest = hdr[:a] + hdr[b:]
    assert "14336" not in rest and "26624" not in rest


def test_get_option_reports_the_defaults_without_a_device(L):
    from closed_chain_motion_planner_amd import get_option, option_table

    table = option_table()
    names = [o["name"] for o in table]
    assert len(names) == len(set(names))
    for o in table:
        assert o["lo"] <= o["default"] <= o["hi"], o
        assert get_option(None, o["name"]) == o["default"]
        assert o["doc"]
    # the defaults the previous header stated wrongly, as the code has them
    assert get_option(None, "small_batch") == 10240 and get_option(None, "lpt_min_batch") == 16384
    assert get_option(None, "num_cus") == 256 and get_option(None, "resident") == 0
    v = C.c_long()
    assert L.ccmp_ctx_get_option(None, b"no_such_option", C.byref(v)) == -1
    assert L.ccmp_ctx_get_option(None, b"small_batch", None) == -1
    assert L.ccmp_ctx_get_option(None, b"side_stream_busy", C.byref(v)) == -1  # a fact of a live context only
    assert L.ccmp_ctx_set_option(None, b"small_batch", 1) == -1
    assert L.ccmp_ctx_option_info(len(table), None, None, None, None, None) == -1 and L.ccmp_ctx_option_info(-1, None, None, None, None, None) == -1
    assert L.ccmp_ctx_option_info(0, None, None, None, None, None) == 0


def test_describe_names_the_regime_on_both_sides_of_every_boundary(L):
    """ccmp_ctx_describe (the built-in policy on a 256-CU device: no context needed) at every size boundary of the policy; the
    line comes from the functions the launches use (csrc/ccmp_policy.cpp)"""
    from closed_chain_motion_planner_amd import describe
    from closed_chain_motion_planner_amd._lib import CALL_GEODESIC, CALL_GEODESIC_BUDGET, CALL_PROJECT, CALL_PROJECT_ANALYTIC, CALL_SAMPLE_PROJECT

    d = lambda n, kind=CALL_PROJECT: describe(None, kind, n)
    # projector, reference arithmetic
    assert "latency kernel alone" in d(1) and "one per sample" in d(2048) and "scout" not in d(2048)
    assert "ticket queue" in d(2049) and "longest-predicted-first" in d(2049)                    # latency_order_min: as soon as blocks take tickets
    assert "latency kernel alone" in d(10240) and "split launch" in d(10241)                     # small_batch
    assert "predicted >= 40" in d(24576) and "512 project_fd_flat_kernel" in d(24576)            # wide split up to kSplitWideMax
    assert "predicted >= 56" in d(24577) and "256 project_fd_flat_kernel" in d(24577)
    assert "(<= 768)" in d(40959) and "(<= 1024)" in d(40960)                                    # samples of the front per CU: 3 -> 4
    assert "below 70 % occupancy" in d(53247) and "per wavefront at <= 10 busy groups" in d(53248)
    assert "split launch" in d(90112) and "split launch" not in d(90113)                         # fd_split_max
    assert "hand-over" in d(131071) and "no hand-over" not in d(131071) and "no hand-over" in d(131072)  # kNoHandoverFrom
    assert d(4096, CALL_SAMPLE_PROJECT).startswith("sample_project B=4096")
    # analytic mode
    a = lambda n: d(n, CALL_PROJECT_ANALYTIC)
    assert "project_row16_kernel (sixteen lanes per sample) alone x 2048 wavefronts" in a(8192) and "project_pair_kernel" not in a(8192)   # analytic_small_batch
    assert "project_pair_kernel (one sample per lane pair) x 257 wavefronts, each handing over" in a(8193) and "holds <= 8 samples, then project_row16_kernel (sixteen lanes per sample) x 514" in a(8193)
    assert "x 3072 wavefronts" in a(262144) and "then project_row16_kernel (sixteen lanes per sample) x 2048 wavefronts" in a(262144)
    # extend step
    g, gb = (lambda n: d(n, CALL_GEODESIC)), (lambda n: d(n, CALL_GEODESIC_BUDGET))
    assert "geodesic_flat_kernel_lat x 1024 blocks, one per edge" in g(1024) and "ticket queue" in g(1025)
    assert "geodesic_flat_kernel_lat" in gb(1024) and "geodesic_flat_kernel x 1025 blocks, one per edge" in gb(1025)
    assert "one per edge" in gb(2048) and "first" not in gb(2048)
    assert "ticket queue" in gb(2049) and "FP32 scout on lane pairs" in gb(2049)                 # geodesic_order_min = geodesic_scout_min: as soon as blocks take tickets
    assert "bulk form" not in gb(13311) and "bulk form" in gb(13312) and "bulk form" not in g(16384)     # geodesic_group_min
    assert "else >= 40" in gb(20479) and "else >= 48" in gb(20480)                               # kGeoGroupHighCut
    assert "else >= 48" in gb(65535) and "else >= 56" in gb(65536)                               # kGeoGroupHigherCut
    assert "hand-over below 50 %" in gb(32767) and "hand-over below 80 %" in gb(32768)           # kGeoGroupLateHandoverFrom
    assert "x 1639 wavefronts" in gb(16384) and "x 2048 wavefronts" in gb(65536)
    assert "lane pairs" in gb(131072) and "lane pairs" not in gb(131073)
    # truncation follows snprintf: the return value is the whole length, the buffer holds what fits
    buf = C.create_string_buffer(16)
    n = L.ccmp_ctx_describe(None, CALL_PROJECT, 4096, buf, 16)
    assert n > 16 and len(buf.value) == 15 and d(4096).startswith(buf.value.decode())
    assert L.ccmp_ctx_describe(None, 99, 4096, buf, 16) == -1


def test_stock_kernels_do_not_spill(ccmp_built):
    """build.py keeps the compiler's own account of every kernel (-Rpass-analysis=kernel-resource-usage -> build/*.resources.json)
    and FAILS the build when a kernel exceeds its scratch bound; here the record is read back: the stock instantiations of the
    projector kernels, of the extend step's kernels and of the resident service kernel use no scratch at all, the general ones spill no vector register."""
    from closed_chain_motion_planner_amd.build import check_resources, resource_report

    rep = resource_report()
    assert {"ccmp_kernels_fd.hip", "ccmp_kernels_flat.hip", "ccmp_kernels_geo.hip", "ccmp_kernels_geo_lat.hip", "ccmp_kernels_resident.hip"} <= set(rep)
    by_name = {k["name"]: k for ks in rep.values() for k in ks}
    for name in ("project_fd_kernel<0, true>", "project_fd_kernel<1, true>", "geodesic_group_kernel<true>", "project_fd_flat_kernel<0, true>",
                 "geodesic_flat_kernel<true>", "geodesic_flat_kernel_lat<true>", "resident_service_kernel<true>"):
        assert by_name[name]["scratch"] == 0 and by_name[name]["scratch_bound"] == 0, (name, by_name[name])
    assert by_name["project_fd_kernel<0, true>"]["vgprs"] <= 168 and by_name["geodesic_group_kernel<true>"]["vgprs"] <= 168  # three wavefronts per SIMD
    assert by_name["project_fd_flat_kernel<0, true>"]["vgprs"] <= 128 and by_name["geodesic_flat_kernel<true>"]["vgprs"] <= 128  # four
    # the analytic mode's lane-pair kernel: no scratch, no AGPR, three wavefronts per SIMD in EVERY instantiation (stock twin arms, stock, calibrated)
    for name in ("project_pair_kernel<true, true>", "project_pair_kernel<true, false>", "project_pair_kernel<false, false>"):
        k = by_name[name]
        assert k["scratch"] == 0 and k["scratch_bound"] == 0 and k["agprs"] == 0 and k["vgprs"] <= 168 and k["occupancy"] >= 3, (name, k)
    for name in ("project_row16_kernel<true>", "project_row16_kernel<false>"):  # its latency kernel: two per SIMD
        k = by_name[name]
        assert k["scratch"] == 0 and k["scratch_bound"] == 0 and k["agprs"] == 0 and k["occupancy"] >= 2, (name, k)
    # the general instantiations (calibrated arms, tilted bases) of the reference-arithmetic kernels: no VGPR goes to scratch; what two of
    # them declare as private segment (132 B) are stack slots of SGPR spills that live in VGPR lanes — their ISA holds no scratch
    # instruction (build.py: _SCRATCH_RULES).  The fused sampler's general instantiations (2.3 KB of spills) are gone.
    general = [n for n in by_name if n.endswith("false>") and n.split("<")[0] in ("project_fd_kernel", "project_fd_flat_kernel", "geodesic_flat_kernel",
                                                                                   "geodesic_flat_kernel_lat", "geodesic_group_kernel", "resident_service_kernel")]
    assert "project_fd_kernel<0, false>" in general and "project_fd_kernel<1, false>" not in by_name and "project_fd_flat_kernel<1, false>" not in by_name
    for name in general:
        assert by_name[name]["vgpr_spill"] == 0 and by_name[name]["scratch"] <= 136, (name, by_name[name])
    assert by_name["geodesic_flat_kernel<false>"]["scratch"] == 0 and by_name["project_fd_flat_kernel<0, false>"]["scratch"] == 0
    assert not {n: k["scratch"] for n, k in by_name.items() if k["scratch"] > 136}
    with pytest.raises(RuntimeError):  # and the check itself bites
        check_resources([{"name": "project_fd_kernel<0, true>", "scratch": 8, "vgprs": 168, "vgpr_spill": 2}], "ccmp_kernels_fd.hip.o")
