#!/usr/bin/env python3
"""Round-6 experiments on the GPU box (development aid).

    python tools/exp_r6.py rounds     time per Newton round of the analytic mode's two kernels against the number of wavefronts
                                      that share a SIMD (a batch of n samples that all need the full 250 rounds)
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402
from tools.measure import CFG, timed  # noqa: E402


def rounds(argv):
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    c.setJacobianMode(1)
    c.setTolerance(1e-9, 1e-9)  # nothing converges: every sample does max_iter rounds
    for kernel, small, per_wave in (("row16", 1 << 40, 4), ("pair", 0, 32)):
        ctx.set_option("analytic_small_batch", small)
        ctx.set_option("analytic_handover", 0)
        for waves in (64, 256, 1024, 2048, 3072, 4096, 6144):
            B = waves * per_wave
            q = c.ambient_uniform_batch(0xC3, 0, B)
            out = torch.empty_like(q)
            ms = timed(lambda: c.project_batch(q, out=out), reps=5)
            it = c.project_batch(q, out=out)[2]
            print("%s: %5d wavefronts (%6d samples, %d rounds each): %.3f ms = %.2f us per round" % (kernel, waves, B, int(it.max()), ms, ms * 1e3 / 251), flush=True)




if __name__ == "__main__":
    {"rounds": rounds}[sys.argv[1]](sys.argv[2:])
