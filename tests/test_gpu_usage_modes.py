"""Ways of calling the library beyond "one call, default stream, wait": non-default streams, HIP graphs, two contexts
side by side.

The *_batch entry points are asynchronous on the caller's stream and keep no host-side state per call, so a planner can
record a launch-bound step (sample -> project -> pre-filter -> compact, or a handful of extend edges) into a HIP graph once
and replay it.  Every replay must redo ALL of the work — the queue heads, counters and the scout's histogram are cleared by
kernels for that reason (a captured hipMemsetAsync left later replays with exhausted queues) — and produce the eager
results, bit for bit."""
import numpy as np
import pytest

from closed_chain_motion_planner_amd import _lib  # (option defaults are asked of the library)
from conftest import config_path
from test_gpu_parity import _constraint

pytestmark = pytest.mark.gpu


def _capture_and_replay(work, replays=3):
    import torch

    ref = [t.clone() for t in work()]  # eager, default stream: also grows the context's workspaces before any capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # the same call on a non-default stream
        on_side = work()
        side.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref, on_side))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = work()
    for rep in range(replays):
        for t in outs:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, outs)), "replay %d differs from the eager call" % rep
    return ref


@pytest.mark.parametrize("mode,B", [(0, 1), (0, 4096), (0, 40000), (0, 70000), (1, 3000), (1, 120000)])
def test_projector_calls_replay_from_a_graph(gpu_ctx, mode, B):
    """single state, latency kernel with its queue, scout + throughput kernel + hand-over (occupancy-driven and at once),
    analytic mode: the latency kernel alone, the lane-pair kernel with its hand-over"""
    c = _constraint("Wine_Bottle", gpu_ctx, mode=mode)

    def work():
        q, ok, it, _ = c.sample_project_batch(0x6A + B, 7, B)
        return q, ok, it

    ref = _capture_and_replay(work)
    assert int(ref[1].sum().item()) > 0 or B == 1


def test_sample_filter_compact_and_extend_replay_from_a_graph(gpu_ctx):
    """a planner step: sampleUniform x 8192 -> proxy pre-filter chained behind the projector's flags -> compaction of the
    survivors; and growTree-shaped extend edges (checkMotion in one launch)"""
    import torch
    from closed_chain_motion_planner_amd.scene import ProxyValidityChecker

    c = _constraint("Wine_Bottle", gpu_ctx)
    scene = ProxyValidityChecker(c).scene
    kept = torch.empty((8192, 14), dtype=torch.float64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")

    def step():
        q, ok, _, _ = c.sample_project_batch(0x57E9, 0, 8192, want_iters=False)
        clr, pair, free = scene.clearance_batch(q, 0.0, ok=ok)
        kept.zero_()  # rows past the count are not written by the compaction
        c.compact_valid(q, free, out=kept, cnt=cnt)
        return q, ok, clr, pair, free, kept, cnt

    ref = _capture_and_replay(step)
    n = int(ref[6].item())
    assert 0 < n == int(ref[4].sum().item())
    frm = ref[5][:64].contiguous()
    to = c.sample_near_project_batch(0x57EA, 0, frm, 0.5, 64, want_iters=False)[0]

    slot = torch.arange(32, device="cuda")[None, :, None]

    def extend():
        st, ns, ok, its = c.discrete_geodesic_batch(frm, to, 32, check_target=True)
        return torch.where(slot < ns[:, None, None], st, 0.0), ns, ok, its  # the list past n_states is not written

    st, ns, ok, its = _capture_and_replay(extend)
    assert int(ok.sum().item()) > 0 and int(ns.max().item()) > 1


def test_bulk_extend_call_replays_from_a_graph(gpu_ctx):
    """a roadmap built in bulk: the extend step's hybrid form (scout, cut of its order, latency blocks on the context's side stream,
    ten-edges-per-wavefront kernel on the caller's) forks and joins through events — capturable, and every replay redoes all of it"""
    import torch

    c = _constraint("Wine_Bottle", gpu_ctx)
    E = 8 * gpu_ctx.num_cus + 900
    q, ok, _, _ = c.sample_project_batch(0x57EB, 0, 8 * E, want_iters=False)
    frm = q[ok == 1][:E].contiguous()
    to = c.sample_near_project_batch(0x57EC, 0, frm, 0.6, E, want_iters=False)[0]
    slot = torch.arange(8, device="cuda")[None, :, None]

    def extend():
        st, ns, okf, its, carry = c.discrete_geodesic_batch(frm, to, 8, want_carry=True, round_budget=40)
        return torch.where(slot < ns[:, None, None], st, 0.0), ns, okf, its, carry

    try:
        gpu_ctx.set_option("geodesic_scout_min", 0)
        gpu_ctx.set_option("geodesic_group_min", 0)
        gpu_ctx.set_option("geodesic_group_pred", 14)  # a front and a rest of comparable size
        st, ns, okf, its, carry = _capture_and_replay(extend)
        assert int((okf == 2).sum().item()) > 0 and int((okf == 1).sum().item()) > 0 and int(ns.max().item()) == 9
    finally:
        gpu_ctx.set_option("geodesic_scout_min", _lib.get_option(None, "geodesic_scout_min"))
        gpu_ctx.set_option("geodesic_group_min", _lib.get_option(None, "geodesic_group_min"))
        gpu_ctx.set_option("geodesic_group_pred", -1)


def test_two_contexts_on_two_streams_run_side_by_side(gpu_ctx):
    """"use one context per stream for concurrency" (include/ccmp.h): two contexts of one device, each on its own stream,
    launched back to back without synchronising in between — reference arithmetic with scout and hand-over on one, the
    analytic mode's two kernels on the other — give what they give alone"""
    import torch
    from closed_chain_motion_planner_amd import Context

    ctx2 = Context(0)
    a = _constraint("Wine_Bottle", gpu_ctx, mode=0)
    b = _constraint("stefan", ctx2, mode=1)
    qa = a.ambient_uniform_batch(0x2C0, 0, 50000)
    qb = b.ambient_uniform_batch(0x2C1, 0, 120000)
    ref_a = [t.clone() for t in a.project_batch(qa)]
    ref_b = [t.clone() for t in b.project_batch(qb)]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        with torch.cuda.stream(s1):
            out_a = a.project_batch(qa)
        with torch.cuda.stream(s2):
            out_b = b.project_batch(qb)
        torch.cuda.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(ref_a, out_a)), rep
        assert all(torch.equal(x, y) for x, y in zip(ref_b, out_b)), rep
    ctx2.close()


def test_second_context_keeps_the_split_launch_concurrent(gpu_ctx):
    """The split launch needs its side stream on another hardware queue than the caller's stream: HIP spreads streams over a
    few queues, and with a second context in the process the side stream once landed on the NULL stream's queue — the front
    ran alone in front of the throughput kernel and a 16 384-sample call took 2.34 ms instead of 1.68 (round 4, kernel trace).
    The side stream now has a priority of its own.  Three contexts, the same call on each: none more than 15 % slower than
    the quickest (the serialised form is 37 % slower)."""
    import statistics

    import torch
    from closed_chain_motion_planner_amd import Context, KinematicChainConstraint

    ctxs = [gpu_ctx, Context(0), Context(0)]
    cons = [KinematicChainConstraint.from_yaml(config_path("Wine_Bottle"), ctx=cx) for cx in ctxs]
    B = 16384
    q = cons[0].ambient_uniform_batch(0xC3, 0, B)
    outs = [torch.empty_like(q) for _ in cons]
    for c, o in zip(cons, outs):
        for _ in range(3):
            c.project_batch(q, out=o)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0].view(torch.int64), o.view(torch.int64)) for o in outs[1:])
    med = []
    for c, o in zip(cons, outs):
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            c.project_batch(q, out=o)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        med.append(statistics.median(ts))
    assert max(med) < 1.15 * min(med), med
