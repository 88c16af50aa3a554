"""In-process A/B of two builds of libccmp (interleaved rounds on ONE device: boxes differ by ~10 %,
so numbers from different gpurun calls cannot be compared).

  python tools/ab.py build "<extra hipcc flags for the FD unit of variant B>"     (build container)
  python tools/ab.py run [obj] [B]                                                (GPU box)
"""
import ctypes as C
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBB = os.environ.get("AB_LIB", os.path.join(ROOT, "closed_chain_motion_planner_amd", "lib", "libccmp_B.so"))


def build(flags):
    from closed_chain_motion_planner_amd import build as b

    objdir = os.path.join(b.HERE, "build_B")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src, fl, *obj in b._UNITS:
        op = os.path.join(objdir, obj[0] if obj else src + ".o")
        # AB_UNIT names a source, or the object of a source that is built twice (ccmp_kernels_geo_lat.hip.o)
        hit = os.environ.get("AB_UNIT", "ccmp_kernels_fd.hip") in (src if not obj else None, os.path.basename(op))
        extra = flags.split() if hit else []
        if hit:
            # AB_DROP="-disable-machine-licm": flags of the unit that variant B is built WITHOUT (an -mllvm option goes with its "-mllvm")
            for d in os.environ.get("AB_DROP", "").split():
                while d in fl:
                    k = fl.index(d)
                    fl = fl[:k - 1] + fl[k + 1:] if k > 0 and fl[k - 1] == "-mllvm" else fl[:k] + fl[k + 1:]
        subprocess.run([b.hipcc_path(), "--offload-arch=" + b.ARCH, "-fPIC", "-std=c++17"] + fl + extra +
                       ["-c", os.path.join(b.CSRC, src), "-o", op], check=True)
        objs.append(op)
    subprocess.run([b.hipcc_path(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", LIBB] + objs, check=True)
    print("built", LIBB, "with", flags)


def run(obj, B):
    import torch
    from closed_chain_motion_planner_amd import _lib, load_config

    LA = _lib.lib()
    LB = C.CDLL(LIBB)
    P = load_config(os.path.join(ROOT, "tests", "golden", "config", obj + ".yaml"))
    P.jacobian_mode = int(os.environ.get("AB_MODE", "0"))  # 1: analytic mode (AB_UNIT=ccmp_kernels_fast.hip)
    libs = {}
    for name, L in (("A", LA), ("B", LB)):
        h = C.c_void_p()
        L.ccmp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        assert L.ccmp_ctx_create(0, C.byref(h)) == 0
        L.ccmp_project_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ccmp_ambient_uniform_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]
        libs[name] = (L, h)
    q = torch.empty((B, 14), dtype=torch.float64, device="cuda")
    LA.ccmp_ambient_uniform_batch(libs["A"][1], C.byref(P), 0xC3, 0, q.data_ptr(), B, None)
    outs = {n: torch.empty_like(q) for n in libs}
    ok = torch.empty(B, dtype=torch.uint8, device="cuda")
    it = torch.empty(B, dtype=torch.int16, device="cuda")

    def once(n):
        L, h = libs[n]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.ccmp_project_batch(h, C.byref(P), q.data_ptr(), outs[n].data_ptr(), ok.data_ptr(), it.data_ptr(), B, None)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
        return e0.elapsed_time(e1)

    for n in libs:
        once(n)
    t = {n: [] for n in libs}
    for _ in range(7):
        for n in ("A", "B"):
            t[n].append(once(n))
    same = torch.equal(outs["A"], outs["B"])
    for n in ("A", "B"):
        print("%s %s B=%d: median %.3f ms  min %.3f ms   %.3e proj/s" % (n, obj, B, statistics.median(t[n]), min(t[n]),
                                                                      B / statistics.median(t[n]) * 1e3))
    print("B/A time ratio (median): %.4f   outputs bit-identical: %s" % (statistics.median(t["B"]) / statistics.median(t["A"]), same))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2] if len(sys.argv) > 2 else "")
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "Wine_Bottle", int(sys.argv[3]) if len(sys.argv) > 3 else 262144)
