"""PCIe-inclusive rate of the host-buffer entry point (ccmp_project_host) at C3: pageable and pinned host memory."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib
ctx = Context(0); L = _lib.lib()
c = KinematicChainConstraint.from_yaml("tests/golden/config/Wine_Bottle.yaml", ctx=ctx)
B = 262144
q = c.ambient_uniform_batch(0xC3, 0, B).cpu()
for name, pin in (("pageable", False), ("pinned", True)):
    qi = q.clone().pin_memory() if pin else q.clone()
    qo = torch.empty_like(qi).pin_memory() if pin else torch.empty_like(qi)
    ok = torch.empty(B, dtype=torch.uint8).pin_memory() if pin else torch.empty(B, dtype=torch.uint8)
    it = torch.empty(B, dtype=torch.int16).pin_memory() if pin else torch.empty(B, dtype=torch.int16)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        rc = L.ccmp_project_host(ctx.handle, C.byref(c.problem), C.cast(qi.data_ptr(), C.POINTER(C.c_double)), C.cast(qo.data_ptr(), C.POINTER(C.c_double)),
                                 C.cast(ok.data_ptr(), C.POINTER(C.c_uint8)), C.cast(it.data_ptr(), C.POINTER(C.c_uint16)), B)
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    ms = np.median(ts[1:]) * 1e3
    print("%-9s host buffers: %.2f ms per 262144 -> %.2e projections/s (device-resident: see bench.py)" % (name, ms, B / ms * 1e3))
