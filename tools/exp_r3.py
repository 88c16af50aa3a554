#!/usr/bin/env python3
"""round-3 scratch experiments on the GPU box (not part of the product)"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402
from tools.measure import near_edges, timed, CFG  # noqa: E402


def geo_stats():
    ctx = Context(0)
    c = KinematicChainConstraint.from_yaml(CFG % "Wine_Bottle", ctx=ctx)
    frm, to = near_edges(c, 16384)
    st, n, okg, its = c.discrete_geodesic_batch(frm, to, 1024)
    torch.cuda.synchronize()
    n_, its_ = n.cpu().numpy(), its.cpu().numpy()
    d = (to - frm).norm(dim=1).cpu().numpy()
    print("dist percentiles 1/50/99/max", np.percentile(d, [1, 50, 99]), d.max())
    print("n_states percentiles 50/90/99/99.9/max", np.percentile(n_, [50, 90, 99, 99.9]), n_.max())
    print("newton its percentiles 50/90/99/99.9/max", np.percentile(its_, [50, 90, 99, 99.9]), its_.max())
    rounds = its_ + n_  # one evaluation round per projected state on top of the updates
    print("rounds mean %.1f p50/p90/p99/p99.9/max" % rounds.mean(), np.percentile(rounds, [50, 90, 99, 99.9]), rounds.max())
    for cap in (4, 8, 12, 16, 24, 32, 64, 128):
        print("edges with n > %d: %d" % (cap, (n_ > cap).sum()))
    print("corr(dist/delta, n) = %.3f, corr(dist, rounds) = %.3f" % (np.corrcoef(d / 0.25, n_)[0, 1], np.corrcoef(d, rounds)[0, 1]))
    order = np.argsort(-rounds)
    print("top-10 rounds", rounds[order[:10]], "their n", n_[order[:10]], "their dist", d[order[:10]])
    for cap in (8, 16, 32, 64):
        ms = timed(lambda: c.discrete_geodesic_batch(frm, to, cap), reps=5)
        print("E=16384 cap=%d: %.3f ms -> %.2f M edges/s" % (cap, ms, 16384 / ms / 1e3), flush=True)
    # order sensitivity: longest-first by true rounds (upper bound for any predictor), cap 16 and 64
    idx = torch.as_tensor(order.copy(), device=frm.device)
    f2, t2 = frm[idx].contiguous(), to[idx].contiguous()
    for cap in (16, 64):
        ms = timed(lambda: c.discrete_geodesic_batch(f2, t2, cap), reps=5)
        print("E=16384 cap=%d longest-first (true rounds): %.3f ms" % (cap, ms), flush=True)
    idx = torch.as_tensor(np.argsort(-d).copy(), device=frm.device)
    f2, t2 = frm[idx].contiguous(), to[idx].contiguous()
    for cap in (16, 64):
        ms = timed(lambda: c.discrete_geodesic_batch(f2, t2, cap), reps=5)
        print("E=16384 cap=%d longest-first (dist): %.3f ms" % (cap, ms), flush=True)
    L = _lib.lib()
    for E in (1, 5, 64, 256, 1024, 4096):
        f, t = frm[:E].contiguous(), to[:E].contiguous()
        states = torch.empty((E, 64, 14), dtype=torch.float64, device=f.device)
        nn = torch.empty(E, dtype=torch.int32, device=f.device)
        ok = torch.empty(E, dtype=torch.uint8, device=f.device)
        it = torch.empty(E, dtype=torch.int32, device=f.device)
        s = torch.cuda.current_stream().cuda_stream

        def raw():
            L.ccmp_geodesic_batch(ctx.handle, C.byref(c.problem), f.data_ptr(), t.data_ptr(), E, 64, states.data_ptr(), nn.data_ptr(),
                                  ok.data_ptr(), it.data_ptr(), s)
        ms_raw = timed(raw, reps=10)
        ms_py = timed(lambda: c.discrete_geodesic_batch(f, t, 64), reps=10)
        raw()
        torch.cuda.synchronize()
        r = (it + nn).max().item()
        print("E=%d raw C call %.3f ms, python mirror %.3f ms, longest edge %d rounds -> %.2f us per round" % (E, ms_raw, ms_py, r, ms_raw * 1e3 / r), flush=True)
    # host-call latency for 5 edges
    sp_f, sp_t = frm[:5].cpu().numpy(), to[:5].cpu().numpy()
    from closed_chain_motion_planner_amd.space import jy_ProjectedStateSpace
    sp = jy_ProjectedStateSpace(c)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        sp.discreteGeodesicBatch(sp_f, sp_t, True)
        ts.append(time.perf_counter() - t0)
    print("space.discreteGeodesicBatch 5 edges host->host: median %.1f us" % (np.median(ts[5:]) * 1e6))


if __name__ == "__main__":
    globals()[sys.argv[1]]()
