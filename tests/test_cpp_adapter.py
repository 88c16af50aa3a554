"""The C++ host side (include/ccmp_ompl_adapter.hpp, part 1) compiled with g++ against libccmp.so:
builds and links without a GPU; on the GPU box it runs and must match the oracle bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, config_path, load_cfg

EXE = os.path.join(ROOT, "tests", "cpp", "adapter_check")


def _build(ccmp_built):
    libdir = os.path.dirname(ccmp_built)
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_check.cpp"),
           "-L", libdir, "-lccmp", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.run(cmd, check=True)
    return EXE


def test_adapter_compiles_and_links_as_plain_cxx14(ccmp_built):
    """C++14 like the reference (CMakeLists.txt), no torch, no HIP headers on the include path."""
    exe = _build(ccmp_built)
    assert os.path.exists(exe)
    hdr = open(os.path.join(ROOT, "include", "ccmp_ompl_adapter.hpp")).read()
    assert "class KinematicChainConstraint : public ompl::base::Constraint" in hdr  # part 2 keeps the reference's name


@pytest.mark.gpu
def test_adapter_matches_oracle(ccmp_built, oracle_det, tmp_path):
    from closed_chain_motion_planner_amd import load_config

    exe = _build(ccmp_built)
    P = oracle_det.checker_problem(config_path("Wine_Bottle"), load_config(config_path("Wine_Bottle")))  # the oracle's own set-up
    q = oracle_det.ambient_uniform_batch(P, 0xAD, 0, 6)
    q[0] = np.array(P.start_joint[:])
    np.savetxt(tmp_path / "states.txt", q, fmt="%.17g")
    out = subprocess.run([exe, config_path("Wine_Bottle"), str(tmp_path / "states.txt")], check=True, capture_output=True,
                         text=True).stdout.splitlines()
    assert out[0] == "setTolerance_throws 1"
    k = 1
    for i in range(6):
        ok_cpu, x_cpu, it_cpu = oracle_det.project(P, q[i])
        hdr = out[k].split()
        x = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in out[k + 1].split()])
        f = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in out[k + 2].split()])
        k += 3
        assert int(hdr[3]) == int(ok_cpu)
        assert np.array_equal(x.view(np.uint64), x_cpu.view(np.uint64))
        assert np.array_equal(f, oracle_det.function(P, x_cpu))
        assert int(hdr[9]) == int(oracle_det.joint_valid(P, x_cpu))
        assert int(hdr[7]) == int(oracle_det.is_satisfied(P, x_cpu))
    proj = []
    for i in range(6):
        ok_cpu, x_cpu, it_cpu = oracle_det.project(P, q[i])
        proj.append(x_cpu)
        parts = out[k + i].split()
        assert int(parts[3]) == int(ok_cpu) and int(parts[5]) == it_cpu
    k += 6
    exp, _, _ = oracle_det.sample_project_batch(P, 42, 0, 7, 2)
    for i in range(7):
        assert out[k] == "sample %d same 1" % i
        x = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in out[k + 1].split()])
        assert np.array_equal(x.view(np.uint64), exp[i].view(np.uint64))
        k += 2
    ok_g, st_g, _ = oracle_det.discrete_geodesic(P, proj[0], proj[1], interpolate=True, max_states=64)
    hdr = out[k].split()
    assert int(hdr[2]) == int(ok_g) and int(hdr[4]) == len(st_g)
    for j in range(len(st_g)):
        x = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in out[k + 1 + j].split()])
        assert np.array_equal(x.view(np.uint64), st_g[j].view(np.uint64))
    k += 1 + len(st_g)
    hdr = out[k].split()
    assert int(hdr[4]) == min(1, len(st_g)) and (int(hdr[2]) == 0 or len(st_g) == 1)  # everything rejected: only `from` survives
    k += 1
    assert out[k] == "geodesic_small_buffer ok %d n %d" % (int(ok_g), len(st_g))  # re-run, not cut
    k += 1

    calls = 0
    for e, (i0, i1) in enumerate(((0, 1), (1, 0), (0, 0))):  # ccmp::discreteGeodesicBatch: three edges in one launch
        ok_e, st_e, _ = oracle_det.discrete_geodesic(P, proj[i0], proj[i1], interpolate=True, max_states=256)
        assert out[k] == "gbatch %d ok %d n %d" % (e, int(ok_e), len(st_e))
        x = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in out[k + 1].split()])
        assert np.array_equal(x.view(np.uint64), st_e[-1].view(np.uint64))
        calls += len(st_e) - 1
        k += 2
    assert out[k] == "gbatch checker_calls %d" % calls  # every state but `from`, edge by edge
    k += 1
    assert out[k] == "gbatch_big edges 1200 mismatches 0"  # the large-batch shape (short lists + round budget + continuation)
    k += 1

    def around(kind, salt, index, param):
        amb = oracle_det.ambient_ref_batch(P, kind, 42 ^ salt, index, proj[0], param, 1)
        _, x, _ = oracle_det.project(P, amb[0])
        return oracle_det.enforce_bounds(x)

    for i in range(6):
        assert out[k] == "near %d" % i
        x = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in out[k + 1].split()])
        assert np.array_equal(x.view(np.uint64), around("near", 0x4E454152, i, 0.2).view(np.uint64))
        k += 2
    x = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in out[k + 1].split()])
    assert out[k] == "gauss 0" and np.array_equal(x.view(np.uint64), around("gaussian", 0x47415553, 0, 0.05).view(np.uint64))
    k += 2
    # ccmp::ShardedProjector: the valid states of 500 sampleUniform draws through the RCCL all-gather, in sample order
    assert out[k] == "threads mismatches 0"  # four threads on one Projector: the mutex serialises the shared context
    k += 1
    # ccmp::ProxyScene: clearance and closest pair of the first states, bit for bit the checker's
    from closed_chain_motion_planner_amd import scene as S

    sph = [(S.frame(0, 7), 0, (0, 0, 0), 0.04), (S.frame(1, 7), 1, (0, 0, 0), 0.04), (S.frame(1, 3), 2, (0, 0, 0.1), 0.06)]
    boxes = [(3,) + S.ProxyValidityChecker.SUB_TABLE[1:]]
    allowed = S.allow([0] * 32, 0, 1)
    assert out[k] == "scene pairs %d" % oracle_det.clearance(P, sph, boxes, allowed, q[0])[2]
    k += 1
    for i in range(4):
        clr_o, pair_o, _ = oracle_det.clearance(P, sph, boxes, allowed, q[i])
        assert out[k] == "clearance %d pair %d" % (i, pair_o)
        got = struct.unpack(">d", bytes.fromhex(out[k + 1].split()[0]))[0]
        assert np.float64(got).view(np.uint64) == np.float64(clr_o).view(np.uint64)
        k += 2
    e_q, e_ok, _ = oracle_det.sample_project_batch(P, 42, 0, 500, 4)
    exp_valid = e_q[e_ok == 1]
    while not out[k].startswith("sharded"):  # RCCL prints its version banner on stdout when the communicator is created
        k += 1
    assert out[k] == "sharded n_valid %d counts 1 first %d" % (len(exp_valid), len(exp_valid))
    for i in range(3):
        x = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in out[k + 1 + i].split()])
        assert np.array_equal(x.view(np.uint64), exp_valid[i].view(np.uint64))


OMPL_EXE = os.path.join(ROOT, "tests", "cpp", "adapter_ompl_check")


def _build_part2(ccmp_built):
    libdir = os.path.dirname(ccmp_built)
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "mock_ompl"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "adapter_ompl_check.cpp"), "-L", libdir, "-lccmp", "-Wl,-rpath," + libdir,
           "-Wl,-rpath,/opt/rocm/lib", "-o", OMPL_EXE]
    subprocess.run(cmd, check=True)
    return OMPL_EXE


def test_part2_type_checks_against_the_interface_mock(ccmp_built):
    """Part 2 (the classes with the reference's names, deriving from OMPL's) compiled against tests/cpp/mock_ompl — an
    interface mock, not OMPL: override signatures, the State* overloads kept visible, state access."""
    assert os.path.exists(_build_part2(ccmp_built))


def _hex_row(line, tag):
    parts = line.split()
    assert parts[0] == tag, line
    return np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in parts[1:]])


@pytest.mark.gpu
def test_part2_control_flow_matches_oracle(ccmp_built, oracle_det):
    from closed_chain_motion_planner_amd import load_config

    exe = _build_part2(ccmp_built)
    P = oracle_det.checker_problem(config_path("Wine_Bottle"), load_config(config_path("Wine_Bottle")))  # the oracle's own set-up
    start = np.array(P.start_joint[:])
    from closed_chain_motion_planner_amd import splitmix64

    env = dict(os.environ, CCMP_SEED="0x5EED")  # the counterpart of ompl::RNG::setSeed: reproducible sampler streams
    out = subprocess.run([exe] + ["%.17g" % v for v in start], check=True, capture_output=True, text=True, env=env).stdout.splitlines()
    space_seed = splitmix64(0x5EED)  # first seed the process hands out
    seed1, seed2 = splitmix64(space_seed), splitmix64(space_seed + 1)  # first and second sampler of the space
    assert out[0] == "throws 1 codim 2"
    assert out[1] == "name ProjectedKinematicChainSpace"
    xa0 = start + 0.05 * ((np.arange(14) % 3) - 1)
    ok_a, xa, _ = oracle_det.project(P, xa0)
    assert out[2] == "project %d satisfied %d" % (int(ok_a), int(oracle_det.is_satisfied(P, xa)))
    assert np.array_equal(_hex_row(out[3], "xa").view(np.uint64), xa.view(np.uint64))
    assert np.array_equal(_hex_row(out[4], "fa"), oracle_det.function(P, xa))
    exp, _, _ = oracle_det.sample_project_batch(P, seed1, 0, 1, 1)  # sampleUniform: the sampler's own seed, running index 0
    assert np.array_equal(_hex_row(out[5], "uniform").view(np.uint64), exp[0].view(np.uint64))
    exp2, _, _ = oracle_det.sample_project_batch(P, seed2, 0, 1, 1)  # a second sampler of the same space: another stream
    assert np.array_equal(_hex_row(out[6], "uniform2").view(np.uint64), exp2[0].view(np.uint64))
    assert not np.array_equal(exp[0], exp2[0])

    def around(kind, salt, index, param):
        amb = oracle_det.ambient_ref_batch(P, kind, seed1 ^ salt, index, xa, param, 1)
        _, x, _ = oracle_det.project(P, amb[0])
        return oracle_det.enforce_bounds(x)

    assert np.array_equal(_hex_row(out[7], "gauss").view(np.uint64), around("gaussian", 0x47415553, 0, 0.05).view(np.uint64))
    assert np.array_equal(_hex_row(out[8], "near0").view(np.uint64), around("near", 0x4E454152, 0, 0.2).view(np.uint64))
    xn = around("near", 0x4E454152, 1, 0.2)  # second draw around the same state: next index of the look-ahead buffer
    assert out[9] == "near_satisfied %d" % int(oracle_det.is_satisfied(P, xn))
    assert np.array_equal(_hex_row(out[10], "near").view(np.uint64), xn.view(np.uint64))
    ok_g, st_g, _ = oracle_det.discrete_geodesic(P, xa, xn, interpolate=True, max_states=256)
    k = 11
    n_full = len(st_g)
    assert out[k] == "geodesic accept 1000000 ok %d n %d checker_calls %d" % (int(ok_g), n_full, n_full - 1)
    for j in range(n_full):
        assert np.array_equal(_hex_row(out[k + 1 + j], "g").view(np.uint64), st_g[j].view(np.uint64))
    k += 1 + n_full
    hdr = out[k].split()
    n_cut = min(n_full, 3)  # `from` + the two states the checker accepted
    assert hdr[:3] == ["geodesic", "accept", "2"] and int(hdr[6]) == n_cut and int(hdr[8]) == min(n_full - 1, 3)
    if n_full > 3:
        d = float(np.sqrt(((st_g[2] - xn) ** 2).sum()))
        assert int(hdr[4]) == int(d <= 0.25)
    k += 1 + n_cut
    assert out[k] == "geodesics ok %d 1 n %d 1 checker_calls %d" % (int(ok_g), n_full, n_full - 1)  # a->b, and b->b: only `from`
    assert np.array_equal(_hex_row(out[k + 1], "gl").view(np.uint64), st_g[-1].view(np.uint64))
    k += 2
    ok_back, _, _ = oracle_det.discrete_geodesic(P, xn, xa, interpolate=True, max_states=256)
    assert out[k] == "checkMotion %d %d" % (int(oracle_det.is_satisfied(P, xn) and ok_g), int(oracle_det.is_satisfied(P, xa) and ok_back))
    assert out[k + 1] == "geodesic_interpolate ok %d" % int(ok_g)
    # PrefilteredValidityChecker: fat fingertip spheres refuse the state without asking the exact checker, thin ones pass it on
    from closed_chain_motion_planner_amd import scene as S

    for line, radius in ((out[k + 2], 0.2), (out[k + 3], 0.01)):
        sph = [(S.frame(0, 7), 0, (0, 0, 0), radius), (S.frame(1, 7), 1, (0, 0, 0), radius)]
        clr, _, _ = oracle_det.clearance(P, sph, [], None, xa)
        free = clr > 0.0
        assert line == "prefilter radius %.2f valid %d exact_calls %d rejected %d" % (radius, int(free), int(free), int(not free))
    assert out[k + 2].split()[4] == "0" and out[k + 3].split()[4] == "1"
    # the reference's configuration call order on a second constraint: same problem, cap still 250, same projection
    assert out[k + 4] == "replay cap 250 delta 0.25 lambda 2.0 tol 0.001 0.005"
    assert out[k + 5] == "replay project %d" % int(ok_a)
    assert np.array_equal(_hex_row(out[k + 6], "xc").view(np.uint64), xa.view(np.uint64))
    # the sampler's two-argument constructor: third sampler of the space, its own stream
    exp3, _, _ = oracle_det.sample_project_batch(P, splitmix64(space_seed + 2), 0, 1, 1)
    assert np.array_equal(_hex_row(out[k + 7], "uniform3").view(np.uint64), exp3[0].view(np.uint64))
    # the adapter's opt-in setResident(true): project(State*) / isSatisfied / function through the resident service kernel
    assert out[k + 8] == "resident project %d satisfied %d" % (int(ok_a), int(oracle_det.is_satisfied(P, xa)))
    assert np.array_equal(_hex_row(out[k + 9], "xr").view(np.uint64), xa.view(np.uint64))
    assert np.array_equal(_hex_row(out[k + 10], "fr"), oracle_det.function(P, xa))


# ---- the drop-in recipe of INTEGRATION.md section 2, compiled: both replacement headers, two translation units --------------
OVERLAY = os.path.join(ROOT, "include", "reference_overlay")
DROPIN_EXE = os.path.join(ROOT, "tests", "cpp", "dropin_check")
REFERENCE = "/root/reference"


def _build_dropin(ccmp_built, tmp):
    libdir = os.path.dirname(ccmp_built)
    objs = []
    for tu in ("dropin_problem", "dropin_planner"):
        obj = os.path.join(str(tmp), tu + ".o")
        # include path order = the patched reference's: its own include/ (here: the overlay holding the two replaced headers),
        # third-party headers (here: the interface mock), $CCMP_ROOT/include
        subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", OVERLAY, "-I", os.path.join(ROOT, "tests", "cpp", "mock_ompl"),
                        "-I", os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "tests", "cpp", tu + ".cpp"), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run(["g++"] + objs + ["-L", libdir, "-lccmp", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", DROPIN_EXE],
                   check=True)
    return DROPIN_EXE, objs


def test_dropin_headers_compile_in_two_translation_units_and_link(ccmp_built, tmp_path):
    """the recipe replaces BOTH headers that define the path's classes — ConstraintFunction.h and
    jy_ProjectedStateSpace.h — and drops src/base/jy_ProjectedStateSpace.cpp from SOURCES.  Two translation units include
    both replacement headers (in either order: each must stand alone), use the classes as ConstrainedPlanningCommon.cpp
    and stefanBiPRM.cpp do, and link into one program: nothing is defined twice, everything in the adapter is inline."""
    exe, objs = _build_dropin(ccmp_built, tmp_path)
    assert os.path.exists(exe)
    # every adapter class really is emitted in both objects (weak, to be merged by the linker) — the test would be empty otherwise
    weak = []
    for obj in objs:
        syms = subprocess.run(["nm", "-C", obj], check=True, capture_output=True, text=True).stdout.splitlines()
        weak.append({l.split(" W ", 1)[1] for l in syms if " W " in l and ("jy_" in l or "ccmp::" in l or "KinematicChainConstraint" in l)})
    both = weak[0] & weak[1]
    assert any(n.startswith("jy_ProjectedStateSpace::traverse(") for n in both), sorted(both)[:5]
    assert len(both) >= 3, sorted(both)
    for hdr in ("closed_chain_motion_planner/base/constraints/ConstraintFunction.h", "closed_chain_motion_planner/base/jy_ProjectedStateSpace.h"):
        text = open(os.path.join(OVERLAY, hdr)).read()
        code = "\n".join(l.split("//")[0] for l in text.split("#pragma once")[1].splitlines())
        assert "#include <ccmp_ompl_adapter.hpp>" in code and "class " not in code  # no class body left behind


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout is not on this box")
def test_overlay_matches_the_reference_it_replaces(tmp_path):
    """With the reference at hand: (1) CMakeLists.diff applies to its CMakeLists.txt and takes jy_ProjectedStateSpace.cpp out
    of SOURCES; (2) each replacement header keeps every #include and the namespace / using line of the header it replaces —
    what the reference's other sources get through them; (3) the classes the adapter defines are exactly those the two
    original headers define, and the dropped source file defines members of no other class."""
    import shutil

    cm = tmp_path / "CMakeLists.txt"
    shutil.copy(os.path.join(REFERENCE, "CMakeLists.txt"), cm)
    subprocess.run(["patch", "-s", "-p1", str(cm)], stdin=open(os.path.join(OVERLAY, "CMakeLists.diff")), check=True, cwd=tmp_path)
    patched = cm.read_text()
    assert "src/base/jy_ProjectedStateSpace.cpp" not in patched and "${CCMP_LIB}" in patched and "CCMP_ROOT}/include" in patched
    import re

    inc = re.compile(r'^\s*#\s*include\s*[<"]([^>"]+)[>"]', re.M)
    for hdr, keep in (("closed_chain_motion_planner/base/constraints/ConstraintFunction.h", "using namespace std;"),
                      ("closed_chain_motion_planner/base/jy_ProjectedStateSpace.h", "namespace ob = ompl::base;")):
        orig = open(os.path.join(REFERENCE, "include", hdr)).read()
        ours = open(os.path.join(OVERLAY, hdr)).read()
        assert set(inc.findall(orig)) <= set(inc.findall(ours)), (hdr, set(inc.findall(orig)) - set(inc.findall(ours)))
        assert keep in orig and keep in ours
    cls = re.compile(r"^\s*class\s+(\w+)\s*:", re.M)
    originals = set()
    for hdr in ("closed_chain_motion_planner/base/constraints/ConstraintFunction.h", "closed_chain_motion_planner/base/jy_ProjectedStateSpace.h"):
        originals |= set(cls.findall(open(os.path.join(REFERENCE, "include", hdr)).read()))
    adapter = open(os.path.join(ROOT, "include", "ccmp_ompl_adapter.hpp")).read()
    part2 = adapter[adapter.index("#ifdef CCMP_WITH_OMPL"):]
    assert originals <= set(cls.findall(part2)), originals - set(cls.findall(part2))
    members = set(re.findall(r"^\w[\w:<> \*&]*?\b(\w+)::\w+\(", open(os.path.join(REFERENCE, "src", "base", "jy_ProjectedStateSpace.cpp")).read(), re.M))
    assert members <= {"jy_ProjectedStateSampler", "jy_ProjectedStateSpace"}, members


@pytest.mark.gpu
@pytest.mark.parametrize("obj,shift", [("Wine_Bottle", 0.0), ("stefan", 0.0), ("stefan", 0.0125)])
def test_dropin_program_runs_and_matches_the_oracle(ccmp_built, oracle_det, tmp_path, obj, shift):
    """The linked two-unit program on the GPU: ConstrainedProblem's constructor order, ArmModels carrying t_wb by value
    (stefan's top arm: linear part diag(-1, -1, 1); with `shift` its base moved 12.5 mm — an edit of grasping_point.cpp that
    must reach the projector without touching the library), growTree's loop call by call and as one launch."""
    exe, _ = _build_dropin(ccmp_built, tmp_path)
    cfg = load_cfg(obj)
    P = oracle_det.checker_problem(config_path(obj))  # the oracle's own set-up from the YAML fixture
    if shift:
        P.base_p[3] += shift  # second arm's base along x, then init_chain_ / t_o7 again by the oracle's own code
        oracle_det.set_start(P, np.array(cfg["start_joint"], dtype=np.float64))
    start = np.array(cfg["start_joint"], dtype=np.float64)
    # arm2 first on the command line for stefan: std::map order (panda_left < panda_top) must come from the replayed
    # _setEnvironment, not from the order of the arguments
    a1, a2 = cfg["arm1"], cfg["arm2"]
    out = subprocess.run([exe, a1["name"], str(a1["index"]), a2["name"], str(a2["index"]), repr(shift)] + ["%.17g" % v for v in start],
                         check=True, capture_output=True, text=True).stdout.splitlines()
    assert out[0] == "arms %d %d name ProjectedKinematicChainSpace cap 250 delta 0.25 lambda 2.0" % (a1["index"], a2["index"])
    assert np.array_equal(_hex_row(out[1], "base_p").view(np.uint64), np.array(P.base_p[:]).view(np.uint64))
    assert np.array_equal(_hex_row(out[2], "init_p").view(np.uint64), np.array(P.init_p[:]).view(np.uint64))
    i = np.arange(14)
    x0 = [start + 0.05 * ((i % 3) - 1), start - 0.04 * ((i % 4) - 1.5), start + 0.03 * ((i % 5) - 2)]
    proj = [oracle_det.project(P, x) for x in x0]
    assert out[3] == "project %d %d %d" % tuple(int(p[0]) for p in proj)
    for k, tag in enumerate(("xa", "xb", "xc")):
        assert np.array_equal(_hex_row(out[4 + k], tag).view(np.uint64), proj[k][1].view(np.uint64))
    assert np.array_equal(_hex_row(out[7], "fa"), oracle_det.function(P, proj[0][1]))
    geo = [oracle_det.discrete_geodesic(P, proj[e][1], proj[2][1], interpolate=True, max_states=256) for e in (0, 1)]
    n = [len(g[1]) for g in geo]
    connected = sum(int(g[0]) for g in geo)
    calls = sum(n) - 2  # every state but `from`, the checker accepts all
    assert out[8] == "grow connected %d %d n %d %d | %d %d checker_calls %d %d" % (connected, connected, n[0], n[1], n[0], n[1], calls, calls)
    k = 9
    for e in (0, 1):
        assert out[k] == "edge %d same 1" % e
        for j in range(n[e]):
            assert np.array_equal(_hex_row(out[k + 1 + j], "g").view(np.uint64), geo[e][1][j].view(np.uint64))
        k += 1 + n[e]
    assert out[k] == "checkMotion %d" % int(oracle_det.is_satisfied(P, proj[2][1]) and geo[0][0])
    assert out[k + 1].startswith("sampled_satisfied ")


# ---- exception safety of Part 2 -------------------------------------------------------------------------------
FAULT_EXE = os.path.join(ROOT, "tests", "cpp", "adapter_fault_check")


def _build_fault_check(ccmp_built):
    """tests/cpp/adapter_fault_check.cpp against the interface mock and lib/libccmp_debug.so (the fault injection of
    include/ccmp_debug.h is not in the product library)"""
    libdir = os.path.dirname(ccmp_built)
    assert os.path.exists(os.path.join(libdir, "libccmp_debug.so"))
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "tests", "cpp", "mock_ompl"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "adapter_fault_check.cpp"), "-L", libdir, "-lccmp_debug", "-Wl,-rpath," + libdir,
           "-Wl,-rpath,/opt/rocm/lib", "-o", FAULT_EXE]
    subprocess.run(cmd, check=True)
    return FAULT_EXE


def test_fault_check_compiles_against_the_mock_and_the_debug_library(ccmp_built):
    assert os.path.exists(_build_fault_check(ccmp_built))


@pytest.mark.gpu
def test_no_exception_escapes_the_reference_overrides(ccmp_built):
    """project / isSatisfied / jointValid / function / the three samplers / discreteGeodesic / checkMotion / discreteGeodesics
    with EVERY compute entry point of the context failing (CCMP_EHIP injected through ccmp_debug_fail_calls), called from a second
    thread as the reference's solution-checker thread calls them (src/planner/stefanBiPRM.cpp:848-849): nothing is thrown, "no" is
    answered, states stay as they were, the error is readable (lastError / lastErrorMessage) and sticky; setTolerance still throws
    ompl::Exception (ConstraintFunction.h:104-108); with the fault lifted the same objects work."""
    from closed_chain_motion_planner_amd import load_config

    exe = _build_fault_check(ccmp_built)
    start = np.array(load_config(config_path("Wine_Bottle")).start_joint[:])
    out = subprocess.run([exe] + ["%.17g" % v for v in start], check=True, capture_output=True, text=True, timeout=300).stdout.splitlines()
    assert out[0] == "fault escaped 0 wrong 0 lastError -2 message_names_the_call 1", out
    assert out[1] == "setTolerance throws 1"
    assert out[2].startswith("after project 1 satisfied 1 sampled 1 geodesic ") and out[2].endswith(" sticky -2"), out[2]
    assert out[2].split()[8] == "1" and int(out[2].split()[10]) >= 1  # reached (the two states are 0.15 rad apart: within delta), the list starts with `from`
    assert out[3] == "cleared 0"
