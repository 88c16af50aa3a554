"""Upper bound of longest-predicted-first scheduling: order from the TRUE iteration counts of a previous run."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402
from tools.time_kernels import timed  # noqa: E402
ctx = Context(0)
L = _lib.lib()
L.ccmp_ctx_set_order_experimental.argtypes = [C.c_void_p, C.c_void_p]
for obj in ("Wine_Bottle", "stefan"):
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    B = 262144
    q = c.ambient_uniform_batch(0xC3, 0, B)
    out = torch.empty_like(q)
    _, ok0, it0 = c.project_batch(q, out=out)
    ref = out.clone()
    c.setJacobianMode(1)
    _, _, ita = c.project_batch(q)
    c.setJacobianMode(0)
    print(obj, "analytic-vs-FD iteration counts: equal %.3f, |diff|<=2 %.3f, corr %.4f" % (
        (ita == it0).float().mean().item(), ((ita - it0).abs() <= 2).float().mean().item(),
        torch.corrcoef(torch.stack([ita.float(), it0.float()]))[0, 1].item()))
    for name, key in (("perfect", it0), ("analytic-predicted", ita)):
        order = torch.argsort(key.to(torch.int32), descending=True).to(torch.int32).contiguous()
        for sched in (0, 1):
            ctx.set_schedule(sched, 0)
            L.ccmp_ctx_set_order_experimental(ctx.handle, None)
            base = timed(lambda: c.project_batch(q, out=out), reps=3)
            L.ccmp_ctx_set_order_experimental(ctx.handle, order.data_ptr())
            lpt = timed(lambda: c.project_batch(q, out=out), reps=3)
            same = torch.equal(out, ref)
            L.ccmp_ctx_set_order_experimental(ctx.handle, None)
            print("%-12s %-18s schedule=%d  in-order %.3f ms   longest-first %.3f ms  (%.3fx)  identical=%s"
                  % (obj, name, sched, base, lpt, base / lpt, same), flush=True)
