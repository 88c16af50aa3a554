import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib
from tools.time_kernels import timed
ctx = Context(0); L = _lib.lib()
L.ccmp_ctx_debug_lpt_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for obj in ("Wine_Bottle", "stefan"):
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    B = 262144
    q = c.ambient_uniform_batch(0xC3, 0, B)
    ctx.set_lpt(1, 0)
    out, ok, it = c.project_batch(q)
    pred = np.zeros(B, dtype=np.uint16)
    assert L.ccmp_ctx_debug_lpt_pred(ctx.handle, pred.ctypes.data, B) == 0
    it = it.cpu().numpy().astype(np.int32); p = pred.astype(np.int32)
    print(obj, "scout vs FD iters: equal %.3f  |d|<=2 %.3f  corr %.4f  pred mean %.2f max %d  fd mean %.2f" % ((p == it).mean(), (np.abs(p - it) <= 2).mean(), np.corrcoef(p, it)[0, 1], p.mean(), p.max(), it.mean()))
    big = it > 80
    print("   of the %d samples with >80 FD iterations, scout says >60 for %.3f" % (big.sum(), (p[big] > 60).mean()))
    # cost of the scout alone: time lpt modes at a batch so large that ordering cannot help
