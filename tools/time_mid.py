"""Mid-size batches (8k..128k): which schedule wins where.  python tools/time_mid.py [obj]"""
import sys
import torch
sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402
from tools.time_kernels import timed  # noqa: E402

obj = sys.argv[1] if len(sys.argv) > 1 else "Wine_Bottle"
ctx = Context(0)
c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
variants = [("default", None)]
variants.append(("wave", dict(sched=(2, 0), lpt=(0, 0), thr=-1, wpc=0)))
for lpt in (0, 1):
    for thr in (2, 4, 6, 10):
        variants.append(("g/lpt%d/thr%d" % (lpt, thr), dict(sched=(1, 0), lpt=(lpt, 0), thr=thr, wpc=0)))
for wpc in (4, 8):
    variants.append(("g/lpt1/thr4/wpc%d" % wpc, dict(sched=(1, 0), lpt=(1, 0), thr=4, wpc=wpc)))
for B in (8192, 12288, 16384, 24576, 32768, 49152, 65536, 98304, 131072):
    q = c.ambient_uniform_batch(0xC3, 0, B)
    out = torch.empty_like(q)
    res = []
    for name, v in variants:
        if v is None:
            ctx.set_schedule(1); ctx.set_lpt(1); ctx.set_option("handover_threshold", -1); ctx.set_waves_per_cu(0)
        else:
            ctx.set_schedule(*v["sched"]); ctx.set_lpt(*v["lpt"]); ctx.set_option("handover_threshold", v["thr"]); ctx.set_waves_per_cu(v["wpc"])
        ms = timed(lambda: c.project_batch(q, out=out), reps=3)
        res.append((ms, name))
    best = min(res)
    print("B=%-7d " % B + "  ".join("%s %.2f" % (n, m) for m, n in res) + "   BEST %s %.2f ms %.2e/s" % (best[1], best[0], B / best[0] * 1e3), flush=True)
