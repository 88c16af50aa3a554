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
  ccmp_api.cpp           -ffp-contract=off -DCCMP_USE_FMA   context, scheduling, launches
  ccmp_host_io.cpp                                          *_host conveniences (staging, pinned block, page-locked caller buffers), sharded host calls
  ccmp_comm.cpp                                             one process / several GPUs: RCCL communicator and sharded entry points
  ccmp_kernels_fast.hip  -ffp-contract=off -DCCMP_USE_FMA   analytic fast mode, bit-identical to the oracle's analytic mode
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
    ("ccmp_kernels_fast.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA"]),
    ("ccmp_kernels_scout.hip", ["-O3", "-ffp-contract=fast", "-ffast-math", "-fno-slp-vectorize"]),  # SLP packs into v_pk_* and spills 310 dwords
    ("ccmp_problem.cpp", ["-O2", "-ffp-contract=off", "-DCCMP_USE_FMA", "-x", "hip"]),
    ("ccmp_api.cpp", ["-O2", "-ffp-contract=off", "-DCCMP_USE_FMA", "-x", "hip"]),
    ("ccmp_host_io.cpp", ["-O2", "-x", "hip"]),
    ("ccmp_comm.cpp", ["-O2", "-x", "hip"]),
    ("ccmp_kernels_scene.hip", ["-O3", "-ffp-contract=off", "-DCCMP_USE_FMA", "-DCCMP_LEAN_SQRT"]),
    ("ccmp_scene.cpp", ["-O2", "-ffp-contract=off", "-DCCMP_USE_FMA", "-x", "hip"]),
]
_HEADERS = ["ccmp_detmath.h", "ccmp_kin.h", "ccmp_solve.h", "ccmp_fd_common.h", "ccmp_flat_newton.h", "ccmp_host.h", "ccmp_ctx.h", "ccmp_scene.h", os.path.join("..", "..", "include", "ccmp.h")]


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


def build_library(force=False, verbose=False):
    """Compile (if stale) and return the path of libccmp.so."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = hipcc_path()
    headers = [os.path.join(CSRC, h) for h in _HEADERS]
    objs, jobs = [], []
    relink = force
    for src, flags, *obj in _UNITS:  # optional third entry: object name (a source built twice)
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, obj[0] if obj else src + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            jobs.append([hipcc, "--offload-arch=" + ARCH, "-fPIC", "-std=c++17"] + flags + ["-c", sp, "-o", op])
    if jobs:  # the units are independent: compile them side by side (hipcc is one process per unit)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)

        workers = max(1, min(len(jobs), int(os.environ.get("CCMP_BUILD_JOBS", "0")) or min(4, os.cpu_count() or 1)))
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(run, jobs))
        relink = True
    if relink or _stale(LIBPATH, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIBPATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIBPATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
