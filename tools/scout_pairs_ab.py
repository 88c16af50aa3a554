#!/usr/bin/env python3
"""FP32 scouts on lane pairs (option "scout_pairs": one arm per lane, two lanes per sample / edge) against one lane per sample:
prediction quality and run time of the calls that use them.  Interleaved; results compared bit for bit.  GPU box."""
import ctypes as C
import sys

import numpy as np
import torch

import os
os.environ.setdefault("CCMP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "closed_chain_motion_planner_amd", "lib", "libccmp_debug.so"))  # reads the scout's predictions: a hook of include/ccmp_debug.h
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402
from measure import near_edges, timed  # noqa: E402

ctx = Context(0)
L = _lib.lib()
L.ccmp_ctx_debug_lpt_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]


def mean(fn, reps=10):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for obj in sys.argv[1:] or ["Wine_Bottle", "stefan"]:
    c = KinematicChainConstraint.from_yaml("tests/golden/config/%s.yaml" % obj, ctx=ctx)
    # prediction quality at 32768 samples (both scouts against the true iteration counts)
    B = 32768
    q = c.ambient_uniform_batch(0xC3, 0, B)
    preds = {}
    for pairs in (0, 1):
        ctx.set_option("scout_pairs", pairs)
        _, _, it = c.project_batch(q)
        torch.cuda.synchronize()
        pred = np.zeros(B, dtype=np.uint16)
        assert L.ccmp_ctx_debug_lpt_pred(ctx.handle, pred.ctypes.data, B) == 0
        preds[pairs] = pred.astype(np.int32)
    itn = it.cpu().numpy().astype(np.int32)
    for pairs in (0, 1):
        p = preds[pairs]
        print("%-11s scout_pairs=%d vs FD iterations: equal %.3f  |d|<=2 %.3f  corr %.4f" % (obj, pairs, (p == itn).mean(), (np.abs(p - itn) <= 2).mean(),
                                                                                          np.corrcoef(p, itn)[0, 1]))
    print("%-11s the two scouts agree on %.4f of the samples, |d| <= 1 on %.4f" % (obj, (preds[0] == preds[1]).mean(), (np.abs(preds[0] - preds[1]) <= 1).mean()), flush=True)
    for B in (3072, 4096, 8192, 14336, 16384, 32768, 65536, 131072, 262144):
        q = c.ambient_uniform_batch(0xC2 if B == 4096 else 0xC3, 0, B)
        out = torch.empty_like(q)
        ctx.set_option("scout_pairs", 0)
        c.project_batch(q, out=out)
        torch.cuda.synchronize()
        ref = out.clone()
        row = []
        for rnd in range(2):
            for pairs, bpc in ((0, 2), (1, 2), (1, 4), (1, 8)):
                ctx.set_option("scout_pairs", pairs)
                ctx.set_option("scout_pair_blocks_per_cu", bpc)
                c.project_batch(q, out=out)
                torch.cuda.synchronize()
                assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), (obj, B, pairs, bpc)
                row.append("%s %.3f" % ("one" if not pairs else "pair/%d" % bpc, mean(lambda: c.project_batch(q, out=out))))
        print("%-11s B=%6d ms  %s" % (obj, B, "  ".join(row)), flush=True)
    for E in (8192, 16384, 32768, 65536):
        frm, to = near_edges(c, E)
        call = lambda: c.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)
        ctx.set_option("scout_pairs", 0)
        ref = call()
        torch.cuda.synchronize()
        row = []
        for rnd in range(2):
            for pairs in (0, 1):
                ctx.set_option("scout_pairs", pairs)
                got = call()
                torch.cuda.synchronize()
                assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3]), (obj, E, pairs)
                row.append("%s %.3f" % ("pair" if pairs else "one", timed(call, 5)))
        print("%-11s E=%6d extend best ms  %s" % (obj, E, "  ".join(row)), flush=True)
    ctx.set_option("scout_pairs", 0)
    ctx.set_option("scout_pair_blocks_per_cu", 2)
