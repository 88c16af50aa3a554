"""How wrong can the scout be on the samples it schedules last?  (decides whether large ordered batches need hand-over)"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib
ctx = Context(0); L = _lib.lib()
L.ccmp_ctx_debug_lpt_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for obj in ("Wine_Bottle", "stefan", "dumbbell"):
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    for seed in (0xC3, 0xD4, 0xE5):
        B = 262144
        q = c.ambient_uniform_batch(seed, 0, B)
        ctx.set_lpt(1, 0)
        out, ok, it = c.project_batch(q)
        pred = np.zeros(B, dtype=np.uint16)
        assert L.ccmp_ctx_debug_lpt_pred(ctx.handle, pred.ctypes.data, B) == 0
        it = it.cpu().numpy().astype(np.int32); p = pred.astype(np.int32)
        order = np.argsort(-p, kind="stable")
        last = order[-30720:]           # the final fill of the throughput kernel
        print("%-12s seed %x: last fill predicted <= %d; true iterations there: max %d, p99.9 %.0f, mean %.1f; overall worst under-prediction %d"
              % (obj, seed, p[last].max(), it[last].max(), np.quantile(it[last], 0.999), it[last].mean(), (it - p).max()), flush=True)
