"""Quick timing sweep of the projector kernels (development aid; bench.py is the contract)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else "Wine_Bottle"
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    print("CUs", ctx.num_cus)
    for mode, name in ((0, "fd"), (1, "analytic")):
        c.setJacobianMode(mode)
        for B in (4096, 262144):
            q = c.ambient_uniform_batch(0xC3, 0, B)
            out = torch.empty_like(q)
            for wpc in ((6, 7, 8, 10, 12) if mode == 0 else (4, 8)):
                ctx.set_waves_per_cu(wpc)
                t0 = time.time()
                ms = timed(lambda: c.project_batch(q, out=out), reps=2 if mode == 0 else 5)
                _, ok, it = c.project_batch(q, out=out)
                print("%-8s B=%-7d waves/CU=%d  %9.3f ms  %12.0f proj/s   ok=%.3f iters=%.2f  (wall %.1fs)"
                      % (name, B, wpc, ms, B / ms * 1e3, ok.float().mean().item(), it.float().mean().item(),
                         time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
