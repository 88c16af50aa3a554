"""Test infrastructure (run by hand, not collected by pytest).  Large-sample parity campaign on the GPU box: the HIP path (default policy, through the C ABI) against the det-build
CPU oracle on all host threads, bit for bit.  Writes profiles/parity_campaign.json.

    python tests/parity_campaign.py [samples_per_case] [edges_per_case]        (default 400000 / 20000; ~1 min of oracle time per case on the box's 16 cores)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # oracle_binding lives next to this file
import torch  # noqa: E402
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint, _lib  # noqa: E402
from oracle_binding import Oracle, build_oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 20000  # geodesic edges per case
import bench  # noqa: E402  (usable_cores: affinity bounded by the cgroup quota — the box reports 256 threads and grants 16)
NCPU = bench.usable_cores()[0]
build_oracle()
orc = Oracle("det")
ctx = Context(0)
# (object, tolerances, seed, jacobian mode, samples); the analytic mode (its own extension) against the oracle's analytic
# mode (lane-pair kernel + hand-over to the latency kernel; a 2-million-sample case where the hand-over is a sliver of the call)
# variants (the general kernel instantiations at scale — every shipped configuration takes the stock-structure ones):
# "calibrated" = DH calibration offsets, different for the two arms (ccmp_set_calibration: axes tilt, offsets fill in, the
# arms stop being twins); "tilted" = arm 2 on a base rotated about two axes and arm 1 on a 1-ulp-off identity (tool_pose's
# full product instead of the diag(+-1) shortcut)
cases = [("Wine_Bottle", None, 0xA1, 0, N, None), ("stefan", None, 0xA2, 0, N, None), ("dumbbell", None, 0xA3, 0, N, None),
         ("stefan", (5e-4, 2.5e-3), 0xA4, 0, N, None),
         ("Wine_Bottle", None, 0xA8, 0, N // 2, "calibrated"), ("stefan", None, 0xA9, 0, N // 2, "tilted"),
         ("Wine_Bottle", None, 0xA5, 1, N, None), ("Wine_Bottle", None, 0xA6, 1, min(N, 250000), None), ("stefan", None, 0xA7, 1, min(N, 250000), None),
         ("dumbbell", None, 0xAB, 1, N // 2, None), ("stefan", None, 0xAC, 1, N // 2, "tilted"),
         ("Wine_Bottle", None, 0xAA, 1, N // 2, "calibrated")]
report = {"samples_per_case": N, "host_threads": NCPU, "cases": []}
from closed_chain_motion_planner_amd.scene import ProxyValidityChecker  # noqa: E402
for obj, tol, seed, mode, N, variant in cases:
    c = KinematicChainConstraint.from_yaml(os.path.join(ROOT, "tests", "golden", "config", obj + ".yaml"), ctx=ctx)
    c.setJacobianMode(mode)
    if tol:
        c.setTolerance(*tol)
    if variant == "calibrated":
        import ctypes as C
        from closed_chain_motion_planner_amd import _lib
        for arm in (0, 1):
            dh = (C.c_double * 28)(*[1e-3 * ((7 * i + 3 * arm) % 5 - 2) for i in range(28)])
            assert _lib.lib().ccmp_set_calibration(C.byref(c.problem), arm, dh) == 0
        c.setInitialPosition(np.array(c.problem.start_joint[:]))
    elif variant == "tilted":
        a, b = 0.3, -0.7
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Rz = np.array([[np.cos(b), -np.sin(b), 0], [np.sin(b), np.cos(b), 0], [0, 0, 1]])
        tilt = (Rz @ Rx).reshape(-1)
        for k in range(9):
            c.problem.base_R[9 + k] = float(tilt[k])
        c.problem.base_R[0] = float(np.nextafter(1.0, 0.0))
        c.setInitialPosition(np.array(c.problem.start_joint[:]))
    if variant:  # a problem this script modified after loading (calibration, tilted base): adopted as bytes
        P = orc.problem_from_bytes(bytes(c.problem))
    else:  # a shipped configuration: the oracle's own set-up from the YAML, compared with the product's byte for byte
        P = orc.checker_problem(os.path.join(ROOT, "tests", "golden", "config", obj + ".yaml"))
        P.tol_pos, P.tol_rot, P.jacobian_mode = c.problem.tol_pos, c.problem.tol_rot, c.problem.jacobian_mode
        assert bytes(P) == bytes(c.problem)
    # projector on resident inputs
    q = c.ambient_uniform_batch(seed, 0, N)
    out, ok, it = c.project_batch(q)
    torch.cuda.synchronize()
    t0 = time.time()
    q_cpu, ok_cpu, it_cpu = orc.project_batch(P, q.cpu().numpy(), NCPU)
    t_cpu = time.time() - t0
    out_h, ok_h, it_h = out.cpu().numpy(), ok.cpu().numpy(), it.cpu().numpy().astype(np.int32)
    same_q = int((out_h.view(np.uint64) == q_cpu.view(np.uint64)).all(axis=1).sum())
    entry = {
        "object": obj, "variant": variant or "shipped configuration", "jacobian_mode": "analytic" if mode else "finite-difference (reference arithmetic)",
        "tolerance": list(tol) if tol else [1e-3, 5e-3], "seed": seed, "samples": N,
        "rows_bit_identical": same_q, "ok_mismatches": int((ok_h != ok_cpu).sum()), "iteration_mismatches": int((it_h != it_cpu).sum()),
        "max_abs_dq": float(np.nanmax(np.abs(out_h - q_cpu))), "ok_fraction": float(ok_cpu.mean()), "mean_iterations": float(it_cpu.mean()),
        "max_iterations": int(it_cpu.max()), "oracle_seconds": round(t_cpu, 1), "oracle_threads": NCPU,
    }
    # the same samples again in mid-size batches — the sizes of the split launch (round 4: the predicted-longest samples on latency
    # blocks beside the throughput kernel, 12288 .. 90112 samples) — against the oracle rows already computed
    # (analytic mode: 4 096 — the latency kernel alone — and the lane-pair kernel with its hand-over at two sizes)
    if True:
        mid_same, mid_n = 0, 0
        for chunk in ((20000, 60000, 13000) if mode == 0 else (4096, 20000, 60000)):
            for a in range(0, min(N, 180000) - chunk + 1, chunk):
                o2, k2, i2 = c.project_batch(q[a:a + chunk].contiguous())
                mid_same += int(((o2.cpu().numpy().view(np.uint64) == q_cpu[a:a + chunk].view(np.uint64)).all(axis=1)
                                 & (k2.cpu().numpy() == ok_cpu[a:a + chunk]) & (i2.cpu().numpy().astype(np.int32) == it_cpu[a:a + chunk])).sum())
                mid_n += chunk
        entry["mid_size_batches_samples"] = mid_n
        entry["mid_size_batches_rows_bit_identical"] = mid_same
        assert mid_same == mid_n
    # fused sampler path: sampleUniform = ambient sample -> project -> enforceBounds
    n2 = N // 4
    sq, sok, sit, _ = c.sample_project_batch(seed + 0x100, 12345, n2)
    sq_cpu, sok_cpu, sit_cpu = orc.sample_project_batch(P, seed + 0x100, 12345, n2, NCPU)
    entry["sampler_rows_bit_identical"] = int((sq.cpu().numpy().view(np.uint64) == sq_cpu.view(np.uint64)).all(axis=1).sum())
    entry["sampler_samples"] = n2
    entry["sampler_ok_mismatches"] = int((sok.cpu().numpy() != sok_cpu).sum())
    # extend step (reference arithmetic; the analytic mode's step loop is covered by tests/test_gpu_callers.py): near-neighbour edges from valid projected states, GPU batch vs the oracle's
    # batch driver; a list that does not fit is reported as max_states + 1 by both
    ne, same_edges, nst = 0, 0, np.zeros(1)
    if mode == 0:
        good = out[ok == 1]
        ne = min(NE, good.shape[0])
        frm = good[:ne].contiguous()
        to, _, _, _ = c.sample_near_project_batch(seed + 0x200, 0, frm, 0.6, ne, want_iters=False)
        maxs = 32
        st_d, nst_d, gok_d, gits_d, carry_d = c.discrete_geodesic_batch(frm, to, maxs, want_carry=True)
        # every edge that did not fit, continued from its last stored state until it is whole, against the oracle's
        # uninterrupted traversal (round 3: ccmp_geodesic_batch_ex)
        whole = c.continue_geodesics(to, st_d, nst_d, gok_d, gits_d, carry_d, maxs)
        cont_same = 0
        for e, (st_e, ok_e, its_e) in whole.items():
            okf, stf, itf = orc.discrete_geodesic(P, frm[e].cpu().numpy(), to[e].cpu().numpy(), interpolate=True, max_states=8192)
            cont_same += int(stf.shape == st_e.shape and np.array_equal(np.ascontiguousarray(st_e).view(np.uint64), stf.view(np.uint64))
                             and bool(ok_e) == okf and its_e == itf)
        entry["geodesic_edges_continued"] = len(whole)
        entry["geodesic_edges_continued_bit_identical"] = cont_same
        entry["geodesic_longest_edge_states"] = max([v[0].shape[0] for v in whole.values()], default=int(nst_d.max()))
        assert cont_same == len(whole)
        st, nst, gok, gits = st_d.cpu().numpy(), nst_d.cpu().numpy(), gok_d.cpu().numpy(), gits_d.cpu().numpy()
        sc, nc, okc, itc = orc.discrete_geodesic_batch(P, frm.cpu().numpy(), to.cpu().numpy(), maxs, NCPU)
        for e in range(ne):
            m = min(int(nst[e]), maxs)
            same_edges += int(nst[e] == nc[e] and gok[e] == okc[e] and gits[e] == itc[e]
                              and np.array_equal(st[e, :m].view(np.uint64), sc[e, :m].view(np.uint64)))
    # the extend step's bulk form (late round 4): a call with a round budget over tens of thousands of edges runs its short edges ten to
    # a wavefront on the throughput layout (geodesic_group_kernel) beside latency blocks for the long ones — forced here whatever the
    # edge count; every edge the budget did not suspend against the oracle's bounded traversal
    if mode == 0:
        nb = min(49152, good.shape[0])
        fb = good[:nb].contiguous()
        tb, _, _, _ = c.sample_near_project_batch(seed + 0x300, 0, fb, 0.6, nb, want_iters=False)
        ctx.set_option("geodesic_group_min", 0)
        sb, nbs, okb, itb, _ = c.discrete_geodesic_batch(fb, tb, 16, want_carry=True, round_budget=128)
        ctx.set_option("geodesic_group_min", _lib.get_option(None, "geodesic_group_min"))  # the library's own default
        sb, nbs, okb, itb = sb.cpu().numpy(), nbs.cpu().numpy(), okb.cpu().numpy(), itb.cpu().numpy()
        scb, ncb, okcb, itcb = orc.discrete_geodesic_batch(P, fb.cpu().numpy(), tb.cpu().numpy(), 16, NCPU)
        liveb = okb != 2
        bulk_same = sum(int(nbs[e] == ncb[e] and okb[e] == okcb[e] and itb[e] == itcb[e]
                            and np.array_equal(sb[e, : min(int(nbs[e]), 16)].view(np.uint64), scb[e, : min(int(ncb[e]), 16)].view(np.uint64)))
                        for e in range(nb) if liveb[e])
        entry["bulk_extend_edges"] = int(liveb.sum())
        entry["bulk_extend_edges_bit_identical"] = bulk_same
        entry["bulk_extend_edges_suspended_by_the_budget"] = int((~liveb).sum())
        assert bulk_same == int(liveb.sum())
    # proxy clearance (the pre-filter ahead of the host's validity test) on the projected states: default scene, both
    # kernels (64-state tiles for the whole batch, one block per state for its first 4096 states)
    if mode == 0:
        scn = ProxyValidityChecker(c).scene
        clr, pair, _ = scn.clearance_batch(out)
        clr1, pair1, _ = scn.clearance_batch(out[:4096].contiguous())
        t0 = time.time()
        clr_cpu, pair_cpu = orc.clearance_batch(P, scn.spheres, scn.boxes, scn.allowed, out_h)
        clr_h, pair_h = clr.cpu().numpy(), pair.cpu().numpy()
        same = (clr_h.view(np.uint64) == clr_cpu.view(np.uint64)) | (np.isnan(clr_h) & np.isnan(clr_cpu))
        entry["clearance_states"] = N
        entry["clearance_bit_identical"] = int((same & (pair_h == pair_cpu)).sum())
        entry["clearance_small_batch_bit_identical"] = int(((clr1.cpu().numpy().view(np.uint64) == clr_cpu[:4096].view(np.uint64))
                                                            & (pair1.cpu().numpy() == pair_cpu[:4096])).sum())
        entry["clearance_pairs_per_state"] = scn.num_pairs
        entry["clearance_oracle_seconds"] = round(time.time() - t0, 1)
        assert entry["clearance_bit_identical"] == N and entry["clearance_small_batch_bit_identical"] == min(N, 4096)
    entry["geodesic_edges"] = ne
    entry["geodesic_edges_bit_identical"] = int(same_edges)
    entry["geodesic_mean_states"] = float(nst.mean())
    report["cases"].append(entry)
    print(json.dumps(entry), flush=True)
    assert same_edges == ne
    assert same_q == N and entry["ok_mismatches"] == 0 and entry["iteration_mismatches"] == 0
    assert entry["sampler_rows_bit_identical"] == n2 and entry["sampler_ok_mismatches"] == 0
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
for path in (os.path.join(ROOT, "profiles", "parity_campaign.json"), os.path.join(ROOT, "gpurun_out", "parity_campaign.json")):
    os.makedirs(os.path.dirname(path), exist_ok=True)  # gpurun_out/ is what travels back from the GPU box
    with open(path, "w") as f:
        json.dump(report, f, indent=1)
print("all cases bit-identical")
