#!/usr/bin/env python3
"""bench.py — constraint projections / second of the MI355X projector (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode fd|analytic] [--obj NAME]

One "step" = one pass of the hot path over one batch of synthetic samples per GPU:
project B uniform joint samples of config Wine_Bottle onto the closed-chain manifold
(KinematicChainConstraint::project, reference arithmetic = FD-faithful mode, bit-identical to the
CPU oracle), compact the valid states, and — for N > 1 — all-gather them over RCCL/xGMI.  Inputs
are generated on the device before the timed region (resident in HBM).  Default workload is
BASELINE.json configs[2] (Wine_Bottle, B = 262144, "HBM-roofline run"), which is also the per-GPU
shard of the 8-GPU config (2097152 / 8); configs[1] (B = 4096) is timed as a secondary figure.

Prints ONE JSON line on rank 0.  `value` is whole-job projections/s (all ranks' samples / max-over-
ranks time).  `roofline` prices the dominant kernel against HBM as the metric asks (225 algorithmic
bytes per projection); the path is FP64-VALU bound, so the honest efficiency figure
`fp64` (2.4 kflop per Newton iteration x iterations / kernel time vs 78.6 TFLOP/s) sits beside it.
`cpu_baseline` times the CPU oracle (glibc build = what the reference would call) on this box's
host cores, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak
BYTES_PER_PROJECTION = 225      # 112 B in + 112 B out + 1 B flag
FLOP_PER_NEWTON_ITER = 2400.0   # analytic formulation = algorithmic minimum
SEEDS = {4096: 0xC2, 262144: 0xC3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=262144, help="samples per GPU per step")
    ap.add_argument("--mode", choices=("fd", "analytic"), default="fd")
    ap.add_argument("--obj", default="Wine_Bottle")
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--tol", default="", help="tolerance1,tolerance2 (default: the reference's 1e-3,5e-3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--backend", default="nccl", help="rehearsal only: gloo lets several ranks share one GPU")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: every rank uses this GPU")
    return ap.parse_args()


def cpu_baseline(obj, seed, ncores, gpu_check=None):
    """CPU oracle (port of the reference algorithm, glibc sin/cos, FD Jacobian) on a bounded sample
    of the same workload: first `sample` samples of the bench batch, all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import yaml
    from oracle_binding import Oracle

    O = Oracle("libm")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "golden", "config", obj + ".yaml")))
    P = O.problem(cfg)
    # calibrate on 256 samples single-thread, then size the sample for ~20 s of CPU work
    q = O.ambient_uniform_batch(P, seed, 0, 256)
    t0 = time.time()
    O.project_batch(P, q, 1)
    t1 = time.time() - t0
    per_core = 256 / t1
    sample = int(max(1024, min(65536, per_core * 20.0)))
    q = O.ambient_uniform_batch(P, seed, 0, sample)
    t0 = time.time()
    _, ok, it = O.project_batch(P, q, ncores)
    wall = time.time() - t0
    # the same algorithm with the analytic Jacobian on the CPU (fair algorithmic comparison for the fast mode)
    Pa = O.problem(cfg)
    Pa.jacobian_mode = 1
    t0 = time.time()
    O.project_batch(Pa, q, ncores)
    wall_a = time.time() - t0
    # in-run parity evidence: the det build of the oracle vs the GPU on the first 2048 samples of the batch
    parity = None
    if gpu_check is not None:
        import numpy as np
        Od = Oracle("det")
        Pd = Od.problem_from_bytes(gpu_check["problem_bytes"])
        n = min(2048, sample)
        qd, okd, itd = Od.project_batch(Pd, gpu_check["q_in"][:n], ncores)  # the very inputs the GPU projected
        d = np.abs(qd - gpu_check["q_out"][:n]).max(axis=1)
        parity = {"samples": n, "max_abs_dq": float(d.max()), "n_gt_1e-6": int((d > 1e-6).sum()),
                  "bit_identical": bool(np.array_equal(qd.view(np.uint64), gpu_check["q_out"][:n].view(np.uint64))),
                  "ok_mismatches": int((okd != gpu_check["ok"][:n]).sum()),
                  "iteration_mismatches": int((itd != gpu_check["iters"][:n]).sum())}
    return {
        "analytic_jacobian_value": sample / wall_a, "parity_gpu_vs_det_oracle": parity,
        "value": sample / wall, "unit": "projections/s", "cores": ncores, "kind": "port",
        "sample": "first %d samples of the bench batch, FD-faithful C oracle (glibc libm, -O2), %d threads; "
                  "single thread: %.1f projections/s" % (sample, ncores, per_core),
        "single_thread_value": per_core, "mean_iters": float(it.mean()),
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.force_device >= 0:
            local_rank = args.force_device
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    from closed_chain_motion_planner_amd import CCMP_JAC_ANALYTIC, CCMP_JAC_FD, Context, KinematicChainConstraint
    from closed_chain_motion_planner_amd.build import build_library
    from closed_chain_motion_planner_amd.distributed import gather_valid

    # no-op when the in-tree .so is current; raises if it cannot be built.  One rank per node builds, the others wait:
    # eight ranks compiling into the same build directory would race.
    if world > 1:
        if local_rank == 0:
            build_library()
        dist.barrier()
    else:
        build_library()
    dev = local_rank if world > 1 else 0
    ctx = Context(dev)
    if args.waves_per_cu:
        ctx.set_waves_per_cu(args.waves_per_cu)
    c = KinematicChainConstraint.from_yaml(os.path.join(ROOT, "tests", "golden", "config", args.obj + ".yaml"), ctx=ctx)
    c.setJacobianMode(CCMP_JAC_FD if args.mode == "fd" else CCMP_JAC_ANALYTIC)
    if args.tol:
        c.setTolerance(*[float(v) for v in args.tol.split(",")])

    B = args.batch
    seed = SEEDS.get(B, 0xC3) if world == 1 else 0xC5
    # synthetic inputs, resident in HBM before the timed region: sampleUniform's ambient samples,
    # global index space sharded contiguously over ranks (rank r owns [r*B, (r+1)*B))
    q_in = c.ambient_uniform_batch(seed, rank * B, B)
    q_out = torch.empty_like(q_in)
    torch.cuda.synchronize()

    kernel_ms = []

    def step(record):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _, ok, it = c.project_batch(q_in, out=q_out, want_iters=True)
        e1.record()
        q_valid, cnt = c.compact_valid(q_out, ok)
        gathered = None
        if world > 1:
            gathered, _ = gather_valid(q_valid, cnt)
        if record:
            kernel_ms.append((e0, e1))
        return ok, it, q_valid, cnt, gathered

    if world > 1:
        # communicator set-up (RCCL builds its rings on the first collective): not a step, so that --warmup 0 still
        # times steps and not the rendezvous
        gather_valid(torch.zeros((1, 14), dtype=torch.float64, device="cuda"), torch.ones(1, dtype=torch.int64, device="cuda"))
    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ok, it, q_valid, cnt, gathered = step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kms = sum(a.elapsed_time(b) for a, b in kernel_ms) / max(1, len(kernel_ms))
    sum_iters = float(it.to(torch.float64).sum().item())
    ok_frac = float(ok.to(torch.float64).mean().item())
    n_valid = int(cnt.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = world * B * args.steps / elapsed
    kernel = "project_fd_kernel+project_fd_flat_kernel+scout_kernel" if args.mode == "fd" else "project_fast_kernel"
    achieved_gbs = BYTES_PER_PROJECTION * B / (kms * 1e-3) / 1e9
    traffic, valu, executed = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):  # PMC numbers come from separate rocprofv3 --pmc passes (tools/profile.sh), not from this run
        try:
            tj = json.load(open(tpath))
            if tj.get("kernel") == kernel and tj.get("batch") == B:
                traffic = tj.get("hbm_bytes_per_launch")
                if tj.get("executed_fp64_flop_per_launch"):
                    ef = tj["executed_fp64_flop_per_launch"]
                    executed = {"achieved": ef / (kms * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": ef / (kms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                "source": "SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 per launch, profiles/%s" % tj.get("tag")}
                if tj.get("valu_wave_insts_per_launch"):
                    n = tj["valu_wave_insts_per_launch"]
                    valu = {"executed_valu_wave_insts_per_launch": n, "cycles_per_inst": 4,
                            "frac_of_fp64_issue_ceiling": n * 4.0 / (kms * 1e-3 * 2.4e9 * 1024),
                            "source": "SQ_INSTS_VALU, profiles/%s" % tj.get("tag")}
        except Exception:
            traffic = None
    fp64_tflops = FLOP_PER_NEWTON_ITER * sum_iters / (kms * 1e-3) / 1e12
    line = {
        "metric": "constraint projections/sec (dual-Panda %s)" % args.obj,
        "value": value, "unit": "projections/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "config/%s.yaml, batch=%d uniform joint samples per GPU (BASELINE configs[2]; per-GPU shard of "
                        "configs[4]), KinematicChainConstraint::project, tol (%g m, %g rad), cap 250" % (args.obj, B, c.problem.tol_pos, c.problem.tol_rot),
            "jacobian_mode": "finite-difference (reference arithmetic, bit-identical to the CPU oracle)"
            if args.mode == "fd" else "analytic (fast mode, not bit-comparable)",
            "global_batch": world * B, "parallelism": "sample-sharded x%d, all-gather of valid states" % world,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel, "kernel_ms": kms,
            "note": "metric asks for %HBM; the kernel is FP64-VALU bound (see fp64)",
            "fp64": {"achieved": fp64_tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": fp64_tflops / FP64_VECTOR_PEAK_TFLOPS,
                     "algorithmic_flop": "2400 per Newton iteration x %.0f iterations per launch" % sum_iters},
            "newton_iterations_per_s": sum_iters / (kms * 1e-3), "valu_issue": valu, "fp64_executed": executed,
        },
        "stats": {"ok_fraction": ok_frac, "mean_newton_iters": sum_iters / B, "valid_states_rank0": n_valid},
    }

    if world == 1 and not args.no_secondary:
        # BASELINE configs[1]: batch 4096 (latency-bound); and the other Jacobian mode at the same batch
        def quick(mode, b, reps):
            c.setJacobianMode(mode)
            qi = c.ambient_uniform_batch(SEEDS.get(b, 0xC3), 0, b)
            qo = torch.empty_like(qi)
            c.project_batch(qi, out=qo)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                c.project_batch(qi, out=qo)
            e1.record()
            torch.cuda.synchronize()
            return b * reps / (e0.elapsed_time(e1) * 1e-3)

        main_mode = CCMP_JAC_FD if args.mode == "fd" else CCMP_JAC_ANALYTIC
        other = CCMP_JAC_ANALYTIC if args.mode == "fd" else CCMP_JAC_FD
        def single_latency_us(reps=40):
            # the reference-signature call: one state through the host entry point (copy in, launch, synchronise, copy out)
            import numpy as np
            c.setJacobianMode(main_mode)
            qs = c.ambient_uniform_batch(0xC1, 0, reps).cpu().numpy()
            ts = []
            for i in range(reps):
                x = qs[i].copy()
                t0 = time.perf_counter()
                c.project(x)
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts[4:]) * 1e6)

        line["secondary"] = {
            "batch4096_projections_per_s": quick(main_mode, 4096, 10),
            "batch32768_projections_per_s": quick(main_mode, 32768, 10),
            "single_project_call_median_us": single_latency_us(),
            ("analytic" if args.mode == "fd" else "fd") + "_mode_projections_per_s": quick(other, B, 5),
        }
        c.setJacobianMode(main_mode)

    if world == 1 and not args.no_cpu_baseline:
        try:
            gpu_check = None
            if args.mode == "fd" and not args.tol:
                gpu_check = {"problem_bytes": bytes(c.problem), "q_in": q_in[:2048].cpu().numpy(), "q_out": q_out[:2048].cpu().numpy(),
                             "ok": ok[:2048].cpu().numpy(), "iters": it[:2048].cpu().numpy().astype("int32")}
            line["cpu_baseline"] = cpu_baseline(args.obj, seed, os.cpu_count() or 1, gpu_check)
        except Exception as e:  # the oracle is a checker; its absence must not fail the GPU bench
            line["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
