"""Cycles per phase of the flat latency kernel (thread 0 of block 0), from a -DCCMP_FLAT_TIMING build:
   AB_UNIT=ccmp_kernels_flat.hip python tools/ab.py build "-DCCMP_FLAT_TIMING"   (build container)
   python tools/time_phases.py                                                     (GPU box)"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401,E402  (HIP runtime first)
from closed_chain_motion_planner_amd import load_config  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "closed_chain_motion_planner_amd", "lib", "libccmp_B.so"))
P = load_config(os.path.join(ROOT, "tests", "golden", "config", "Wine_Bottle.yaml"))
h = C.c_void_p()
L.ccmp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
assert L.ccmp_ctx_create(0, C.byref(h)) == 0
L.ccmp_project_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
L.ccmp_debug_flat_timing.argtypes = [C.c_void_p, C.c_int]
rng = np.random.default_rng(5)
lb = np.array(P.lb[:]); ub = np.array(P.ub[:])
names = ["A angles+sincos", "B chain+pose", "C residual", "D test+combine", "E solve+update"]
tot = np.zeros(8); iters = 0
for rep in range(20):
    q = np.concatenate([lb + rng.random(7) * (ub - lb), lb + rng.random(7) * (ub - lb)])
    out = np.empty(14); ok = np.zeros(1, np.uint8); it = np.zeros(1, np.uint16)
    t = (C.c_ulonglong * 8)()
    L.ccmp_debug_flat_timing(t, 1)
    assert L.ccmp_project_host(h, C.byref(P), q.ctypes.data, out.ctypes.data, ok.ctypes.data, it.ctypes.data, 1) == 0
    L.ccmp_debug_flat_timing(t, 1)
    if rep >= 2:
        tot += np.array(list(t), dtype=np.float64); iters += int(it[0]) + 1
print("iterations timed:", iters)
for k, n in enumerate(names):
    print("%-18s %8.0f cycles per iteration" % (n, tot[k] / iters))
print("%-18s %8.0f cycles per iteration (counter clock)" % ("sum", tot[:5].sum() / iters))
