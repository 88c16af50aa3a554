"""Builds libccmp.so (HIP kernels for gfx950 + the C-ABI host code) in-tree with hipcc.

Translation units with deliberately different flags:
  ccmp_kernels_fd.hip    -ffp-contract=off -DCCMP_USE_FMA   canonical, bit-reproducible arithmetic (throughput kernel);
                         -DCCMP_LEAN_SQRT: ccmp_detmath.h's wave-uniform fast path of the IEEE square root (same bits, -1.9 %)
  ccmp_kernels_wave.hip  -ffp-contract=off -DCCMP_USE_FMA   same arithmetic, one-wave-per-sample kernels
  ccmp_kernels_flat.hip  -ffp-contract=off -DCCMP_USE_FMA   same arithmetic, one 128-thread block per sample (latency kernel)
  ccmp_kernels_geo.hip   -ffp-contract=off -DCCMP_USE_FMA   the extend step on the same Newton routine (ccmp_flat_newton.h), built twice:
                         like the flat unit (throughput flavour) and -DCCMP_GEO_LATENCY with machine LICM and a 256-register
                         budget (latency flavour)
  ccmp_problem.cpp       -ffp-contract=off -DCCMP_USE_FMA   host set-up (problem, constants) in the same rounding model
  ccmp_api.cpp           -ffp-contract=off -DCCMP_USE_FMA   context, launches            } compiled a second time with -DCCMP_DEBUG_HOOKS for
  ccmp_policy.cpp                                           option table, plans, describe } lib/libccmp_debug.so (include/ccmp_debug.h)
  ccmp_resident.cpp                                         opt-in resident service kernel for single-state calls (host side)
  ccmp_kernels_resident.hip -ffp-contract=off -DCCMP_USE_FMA  ... its device side, on the latency flavour's Newton routine
  ccmp_host_io.cpp                                          *_host conveniences (staging, pinned block, page-locked caller buffers), sharded host calls
  ccmp_comm.cpp                                             one process / several GPUs: RCCL communicator and sharded entry points
  ccmp_kernels_fast.hip  -ffp-contract=off -DCCMP_USE_FMA -DCCMP_LEAN_SQRT   analytic fast mode (one sample per lane pair), bit-identical to the oracle's analytic mode
  ccmp_kernels_scout.hip -ffast-math                        FP32 iteration-count predictor + ordering (never touches results)
  ccmp_kernels_scene.hip -ffp-contract=off -DCCMP_USE_FMA   proxy-geometry clearance (pre-filter ahead of the host's MoveIt test)
  ccmp_scene.cpp                                            proxy scenes: validation, pair list, launches
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# CCMP_LIBRARY: another build of the same library (tools/sanitize_cpu.py: host code under ASan/UBSan); never set in production
LIBPATH = os.environ.get("CCMP_LIBRARY") or os.path.join(LIBDIR, "libccmp.so")
# the same sources with the test / tool hooks of include/ccmp_debug.h (-DCCMP_DEBUG_HOOKS on the two host units that carry them)
DEBUG_LIBPATH = os.path.join(LIBDIR, "libccmp_debug.so")
_DEBUG_UNITS = ("ccmp_api.cpp", "ccmp_policy.cpp")
_DEBUG_ONLY_UNITS = [("ccmp_kernels_debug.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA", "-DCCMP_LEAN_SQRT", "-mllvm", "-disable-machine-licm"])]  # the device probe of ccmp_detmath.h
ARCH = "gfx950"

_UNITS = [
    # -disable-machine-licm: LLVM's machine LICM hoists ~35 FP64 polynomial/literal constants out of the Newton
    # loop into VGPR pairs and then spills them to scratch (168 VGPRs + 30 spilled dwords); without it the
    # throughput kernel needed 133 VGPRs and no scratch (measured +3.3 %, in-process A/B; today's kernel: 167, no scratch,
    # three wavefronts per SIMD).  The wave kernels are
    # 2.7 % slower with the option, hence their own unit.
    ("ccmp_kernels_fd.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA", "-DCCMP_LEAN_SQRT", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
     + os.environ.get("CCMP_FD_EXTRA_FLAGS", "").split()),
    ("ccmp_kernels_wave.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA"]),
    # max-ilp scheduling: -0.6 % (throughput kernel) ... -1.5 % (latency kernel, single state), in-process A/B
    ("ccmp_kernels_flat.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
    # the extend step, built twice from one source.  Throughput flavour (the projector's latency kernel's flags: eight blocks
    # per CU) for calls that bound the Newton rounds per edge; latency flavour (machine LICM on, 256-register budget, four
    # blocks per CU: the ~60 FP64 literals of a Newton round stay in registers instead of being re-materialised every
    # round) for calls whose end is one edge's serial chain: 16 384 edges without a round budget -16 %, with one +23 %.
    ("ccmp_kernels_geo.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
    ("ccmp_kernels_geo.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA", "-DCCMP_GEO_LATENCY", "-DCCMP_SUMS_IN_LANE", "-DCCMP_FLAT_MIN_WAVES=2", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
     "ccmp_kernels_geo_lat.hip.o"),
    # the resident service kernel: one block alone on its SIMDs — the latency flavour's flags (registers are free, fewest instructions per round)
    ("ccmp_kernels_resident.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA", "-DCCMP_SUMS_IN_LANE", "-DCCMP_FLAT_MIN_WAVES=2", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
    # the analytic mode's lane-pair kernel: without machine LICM (the ~35 FP64 literals of sincos / atan would be held in
    # registers across the Newton loop and spilled: 168 registers + 36 B of scratch instead of 142 and none)
    ("ccmp_kernels_fast.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA", "-DCCMP_LEAN_SQRT", "-mllvm", "-disable-machine-licm"]),
    ("ccmp_kernels_scout.hip", ["-O3", "-ffp-contract=fast", "-ffast-math", "-fno-slp-vectorize"]),  # SLP packs into v_pk_* and spills 310 dwords
    ("ccmp_problem.cpp", ["-O2", "-ffp-contract=off", "-DCCMP_USE_FMA", "-x", "hip"]),
    ("ccmp_api.cpp", ["-O2", "-ffp-contract=off", "-DCCMP_USE_FMA", "-x", "hip"]),
    ("ccmp_policy.cpp", ["-O2", "-x", "hip"]),
    ("ccmp_resident.cpp", ["-O2", "-x", "hip"]),
    ("ccmp_host_io.cpp", ["-O2", "-x", "hip"]),
    ("ccmp_comm.cpp", ["-O2", "-x", "hip"]),
    ("ccmp_kernels_scene.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA", "-DCCMP_LEAN_SQRT"]),
    ("ccmp_scene.cpp", ["-O2", "-ffp-contract=off", "-DCCMP_USE_FMA", "-x", "hip"]),
]
_HEADERS = ["ccmp_detmath.h", "ccmp_kin.h", "ccmp_solve.h", "ccmp_fd_common.h", "ccmp_flat_newton.h", "ccmp_host.h", "ccmp_ctx.h", "ccmp_policy.h", "ccmp_resident.h", "ccmp_split.h", "ccmp_fd_newton_phase1.inc", "ccmp_fd_newton_phase2.inc", "ccmp_geo_edge.h", "ccmp_geo_edge_body.inc", "ccmp_scene.h", os.path.join("..", "..", "include", "ccmp.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libccmp.so cannot be built (no CPU fallback exists)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


# ---- build-time check of the kernels' register allocation (round 5) ------------------------------------------------------------
# Every device unit is compiled with -Rpass-analysis=kernel-resource-usage; the remarks are kept beside the object
# (build/<unit>.resources.json) and checked against the bounds below — a build whose hot kernels have started to spill FAILS
# instead of shipping a silently slower library.  Scratch is in bytes per lane.
#   * the STOCK instantiations (template argument `true`: both arms carry the uncalibrated Panda's exact zeros — what the
#     reference ships, ConstrainedPlanningCommon.cpp:97 has the calibration commented out) of the projector kernels, of the extend
#     step's kernels (geodesic_flat_kernel both builds, geodesic_group_kernel) and of the resident service kernel: NO scratch.
#     (geodesic_group_kernel<true> had 64-76 B until round 5: ten loop-invariant per-lane values the allocator parked in scratch
#     under the bound of three wavefronts per SIMD; they are now recomputed where they are used or kept in LDS.  Without the bound
#     the kernel takes 181 registers, no scratch, and is 5 % SLOWER at 16 384 - 32 768 edges — profiles/r05_bulk_live_ab.log, B/r4
#     of version 1: the front's latency blocks lose the register space.)
#   * the general instantiations (calibrated arms, tilted bases): at most 136 B of PRIVATE SEGMENT and not one scratch
#     instruction — the 96-132 B that project_fd_kernel<0,false>, geodesic_group_kernel<false> and geodesic_flat_kernel<false>
#     declare are stack slots of SGPR spills that the allocator then kept in VGPR lanes (v_writelane / v_readlane; the ISA of
#     those kernels contains no scratch_load / scratch_store: tests/test_host_cabi.py checks the objects).  The fused-sampler
#     general instantiation project_fd_kernel<1,false> (2.3 KB of real spills) was removed in round 6: sampleUniform on
#     calibrated arms runs unfused (ccmp_api.cpp: project_common);
#   * everything else (scouts, small per-lane kernels, analytic mode, scene): at most 64 B unless listed.
_SCRATCH_RULES = [  # (regex on the demangled name, bound); first match wins
    (r"(project_fd_kernel|project_fd_flat_kernel|project_fd_wave_kernel|geodesic_flat_kernel(_lat)?|geodesic_group_kernel|resident_service_kernel)<(\d+, )?true>", 0),
    (r"(project_fd_kernel|project_fd_flat_kernel|project_fd_wave_kernel|geodesic_flat_kernel(_lat)?|geodesic_group_kernel|resident_service_kernel)<(\d+, )?false>", 136),
    (r"project_pair_kernel|project_row16_kernel", 0),  # analytic mode, every instantiation (stock twin arms, stock, calibrated)
    (r"scout_|clearance", 400),
    (r".", 64),
]


def _demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", shutil.which("c++filt")):
        if tool and os.path.exists(tool):
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
            return dict(zip(names, out))
    return {n: n for n in names}


def parse_resource_remarks(stderr_text):
    """[{name, vgprs, agprs, sgprs, scratch, occupancy, lds, vgpr_spill, sgpr_spill}] from -Rpass-analysis=kernel-resource-usage"""
    import re

    kernels, cur = [], None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
    for line in stderr_text.splitlines():
        m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"mangled": v}
            kernels.append(cur)
        elif cur is not None and k in keys:
            cur[keys[k]] = int(v)
    dm = _demangle([k["mangled"] for k in kernels])
    for k in kernels:
        k["name"] = dm[k["mangled"]].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    return kernels


def check_resources(kernels, unit):
    """raises RuntimeError when a kernel of `unit` exceeds its scratch bound"""
    import re

    bad = []
    for k in kernels:
        bound = next(b for rx, b in _SCRATCH_RULES if re.search(rx, k["name"]))
        k["scratch_bound"] = bound
        if k.get("scratch", 0) > bound:
            bad.append("%s: %d B/lane of scratch (bound %d), %d VGPRs, %d spilled" % (k["name"], k["scratch"], bound, k.get("vgprs", -1), k.get("vgpr_spill", -1)))
    if bad and not os.environ.get("CCMP_ALLOW_SPILLS"):
        raise RuntimeError("register allocation of %s regressed (closed_chain_motion_planner_amd/build.py: _SCRATCH_RULES; CCMP_ALLOW_SPILLS=1 "
                           "builds anyway):\n  " % unit + "\n  ".join(bad))


def resource_report():
    """{unit object: [kernel records]} of the last build (build/*.resources.json)"""
    import glob
    import json

    return {os.path.basename(f)[:-len(".resources.json")]: json.load(open(f)) for f in sorted(glob.glob(os.path.join(HERE, "build", "*.resources.json")))}


def build_library(force=False, verbose=False):
    """Compile (if stale) and return the path of libccmp.so."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = hipcc_path()
    headers = [os.path.join(CSRC, h) for h in _HEADERS]
    objs, dbg_objs, jobs = [], [], []
    relink = force
    for src, flags, *obj in _UNITS:  # optional third entry: object name (a source built twice)
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, obj[0] if obj else src + ".o")
        objs.append(op)
        variants = [(op, [])]
        if src in _DEBUG_UNITS:  # the debug library's copy of the unit
            dp = os.path.join(objdir, src + ".dbg.o")
            dbg_objs.append(dp)
            variants.append((dp, ["-DCCMP_DEBUG_HOOKS"]))
        else:
            dbg_objs.append(op)
        for target, extra in variants:
            if force or _stale(target, [sp] + headers):
                jobs.append([hipcc, "--offload-arch=" + ARCH, "-fPIC", "-std=c++17"] + flags + extra +
                            (["-Rpass-analysis=kernel-resource-usage"] if src.endswith(".hip") else []) + ["-c", sp, "-o", target])
    for src, flags in _DEBUG_ONLY_UNITS:
        sp, op = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        dbg_objs.append(op)
        if force or _stale(op, [sp] + headers):
            jobs.append([hipcc, "--offload-arch=" + ARCH, "-fPIC", "-std=c++17"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", sp, "-o", op])
    if jobs:  # the units are independent: compile them side by side (hipcc is one process per unit)
        import json
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            rest = "\n".join(ln for ln in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln)
            if rest.strip():
                print(rest)
            if r.returncode != 0:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            if "-Rpass-analysis=kernel-resource-usage" in cmd:
                kernels = parse_resource_remarks(r.stderr)
                try:
                    check_resources(kernels, os.path.basename(cmd[-1]))
                except RuntimeError:
                    os.remove(cmd[-1])  # a failed check must not leave an object that looks up to date
                    raise
                json.dump(kernels, open(cmd[-1][:-2] + ".resources.json", "w"), indent=1)

        workers = max(1, min(len(jobs), int(os.environ.get("CCMP_BUILD_JOBS", "0")) or min(4, os.cpu_count() or 1)))
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(run, jobs))
        relink = True
    default_lib = os.path.join(LIBDIR, "libccmp.so")
    for target, members in ((default_lib, objs), (DEBUG_LIBPATH, dbg_objs)):
        if relink or _stale(target, members):
            cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", target] + members
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
    return LIBPATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
