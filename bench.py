#!/usr/bin/env python3
"""bench.py — constraint projections / second of the MI355X projector (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode fd|analytic] [--obj NAME]

One "step" = one pass of the hot path over one batch of synthetic samples per GPU:
project B uniform joint samples of config Wine_Bottle onto the closed-chain manifold
(KinematicChainConstraint::project, reference arithmetic = FD-faithful mode, bit-identical to the
CPU oracle), compact the valid states, and — for N > 1 — all-gather them over RCCL/xGMI (one
fixed-capacity collective, no host synchronisation inside the step).  Inputs are generated on the
device before the timed region (resident in HBM).  Default workload is BASELINE.json configs[2]
(Wine_Bottle, B = 262144, "HBM-roofline run"), which is also the per-GPU shard of the 8-GPU config
(2097152 / 8); configs[1] (B = 4096), configs[3] (stefan, both tolerance sets) and the extend step
are timed as secondary figures, each with an in-run bitwise check against the det oracle.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(fresh child processes, one per GPU, rendezvous on 127.0.0.1) and relays rank 0's line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.

Prints ONE JSON line on rank 0.  `value` is whole-job projections/s (all ranks' samples / max-over-
ranks time).  `roofline` prices the dominant kernel against HBM as the metric asks (225 algorithmic
bytes per projection); the path is FP64-VALU bound, so the honest efficiency figure
`fp64` (2.4 kflop per Newton iteration x iterations / kernel time vs 78.6 TFLOP/s) sits beside it.
`cpu_baseline` times the CPU oracle (glibc build = what the reference would call) on this box's
host cores — the count actually usable, not the count the OS reports — rank 0, N = 1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak
BYTES_PER_PROJECTION = 225      # 112 B in + 112 B out + 1 B flag
FLOP_PER_NEWTON_ITER = 2400.0   # analytic formulation = algorithmic minimum
SEEDS = {4096: 0xC2, 262144: 0xC3}
GATHER_CAPACITY = 0.5           # gather block = this fraction of the shard (about 22 % of uniform samples are valid)


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
    ap.add_argument("--one-process", type=int, default=0,
                    help="internal: this process is the ONE-process child — it drives that many GPUs through the C ABI's sharded entry "
                         "points (SURVEY.md section 8(e): 'implement both, report both') and prints its own JSON line")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, rendezvous over gloo on the CPU, print the launch facts and exit (no GPU work: "
                         "checks the launcher on a box without GPUs)")
    return ap.parse_args()


# ---- N ranks from one command ------------------------------------------------------------------------------------------
def launch_ranks(n):
    """Parent of `python bench.py --gpus N`: N fresh child processes, one rank each; this process never touches a GPU
    (no exec of a process that has initialised HIP).  Relays rank 0's JSON line; non-zero exit if any rank fails."""
    from closed_chain_motion_planner_amd.build import build_library

    build_library()  # hipcc needs no GPU; the ranks then find the library current and do not race to build it
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", CCMP_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's line is collected by a reader thread; the parent polls EVERY child: the first rank that dies (build, device
    # initialisation, rendezvous) takes the others down with it instead of leaving them in a collective until torch's
    # own timeout, and the whole job has a deadline
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("CCMP_BENCH_TIMEOUT", "2400"))
    rc, why = 0, ""
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc, why = (bad[0][1] if bad[0][1] > 0 else 1), "rank %d exited with code %d" % bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                rc, why = 124, "deadline (CCMP_BENCH_TIMEOUT) passed"
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # only the children started above, by handle
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    reader.join(timeout=10)
    text = out0[0] if out0 else ""
    sys.stdout.write(text or "")
    sys.stdout.flush()
    if rc != 0:
        print("bench.py: %s; the other ranks were stopped" % why, file=sys.stderr)
        sys.exit(rc)


# ---- one process, n GPUs: the reference's own shape (src/main.cpp:27-63 is a single process) ---------------------------------------
def one_process_child(args):
    """The child started after the ranks are done (run_one_process below): ONE process, one context per GPU, the batch
    sharded inside the C ABI — `direct`: ccmp_sample_project_sharded_host, every GPU returns its shard straight to the host, no
    collective; `rccl`: ccmp_sample_project_sharded, ncclCommInitAll over the devices + one ncclAllGather of the compacted
    valid states.  262 144 samples per GPU and call (BASELINE configs[4]'s shard), results in HOST memory as the host tree
    consumes them, so these rates include the download that `value` (device-resident, one process per GPU) does not.
    Prints one JSON line; the direct form's result is flushed to --one-process-out before the collective is touched."""
    import numpy as np
    import torch  # noqa: F401  (loads the HIP runtime the library binds to)

    from closed_chain_motion_planner_amd import Communicator, Context, KinematicChainConstraint

    n = args.one_process
    devs = [args.force_device] * n if args.force_device >= 0 else list(range(n))
    ctxs = [Context(d) for d in devs]
    c = KinematicChainConstraint.from_yaml(os.path.join(ROOT, "tests", "golden", "config", args.obj + ".yaml"), ctx=ctxs[0])
    B = args.batch * n
    out = {"gpus": n, "devices": devs, "samples_per_call": B, "samples_per_gpu": args.batch, "results_in": "host memory (pageable)"}

    def timed(fn, reps=5, warm=2):
        for _ in range(warm):
            r = fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), r

    sec, (q, ok, it) = timed(lambda: c.sample_project_sharded_host(0xC5, 0, B, ctxs))
    launch_ms, start_ms = c.sharded_host_last_timing(ctxs)
    out["direct"] = {"entry": "ccmp_sample_project_sharded_host", "collective": "none", "shards": n, "ms_per_call": sec * 1e3,
                     "projections_per_s": B / sec, "ok_fraction": float(ok.mean()), "shard_launch_ms": launch_ms, "shard_start_ms_on_gpu": start_ms}
    ref_ok = ok.copy()
    print(json.dumps(out), flush=True)  # (a first line: kept by the parent if the collective below never returns)
    if len(set(devs)) == n:  # RCCL wants distinct devices: a rehearsal with two contexts on one card has no collective form
        try:
            comm = Communicator(ctxs)
            sec, (valid, counts, full) = timed(lambda: c.sample_project_sharded(0xC5, 0, B, comm, block_rows=max(1, args.batch // 2)))
            kernel_ms, gather_ms = comm.last_timing()
            out["rccl"] = {"entry": "ccmp_sample_project_sharded", "collective": "ncclAllGather of fixed-capacity blocks of valid states",
                           "ranks": n, "ms_per_call": sec * 1e3, "projections_per_s": B / sec, "valid_states": int(sum(counts)),
                           "kernel_ms_per_gpu": kernel_ms, "gather_ms_per_gpu": gather_ms,
                           "same_flags_as_direct": bool(np.array_equal(full[1], ref_ok))}
        except Exception as e:
            out["rccl"] = {"error": repr(e)}
    else:
        out["rccl"] = {"skipped": "the contexts share a device (rehearsal): RCCL needs one device per rank"}
    print(json.dumps(out), flush=True)


def run_one_process(args, n):
    """starts the one-process child (a fresh process: never an exec of one that has touched a GPU) with a deadline, and
    returns its last JSON line — {"error": ...} if it produced none"""
    cmd = [sys.executable, os.path.abspath(__file__), "--one-process", str(n), "--batch", str(args.batch), "--obj", args.obj]
    if args.force_device >= 0:
        cmd += ["--force-device", str(args.force_device)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "CCMP_BENCH_CHILD")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        so, se = p.communicate(timeout=float(os.environ.get("CCMP_ONE_PROCESS_TIMEOUT", "240")))
    except subprocess.TimeoutExpired:
        p.kill()  # this child only, by handle
        so, se = p.communicate()
        se = (se or "") + "\n[deadline passed: child stopped]"
    lines = [ln for ln in (so or "").splitlines() if ln.startswith("{")]
    if lines:
        try:
            res = json.loads(lines[-1])
            if p.returncode not in (0, None):
                res["child_exit_code"] = p.returncode
                res["child_stderr_tail"] = (se or "")[-400:]
            return res
        except Exception:
            pass
    return {"error": "the one-process child produced no result (exit code %s)" % p.returncode, "stderr_tail": (se or "")[-600:]}


# ---- host cores -----------------------------------------------------------------------------------------------------------
def usable_cores():
    """(cores this process may use, how that was found, what the OS reports): scheduler affinity bounded by the cgroup's
    CPU quota — a leased box reports every hardware thread of the machine but grants a share of them"""
    reported = os.cpu_count() or 1
    n, how = reported, "os.cpu_count"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, how = a, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    quota = None
    try:  # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, how = max(1, int(quota)), "cgroup cpu quota"
    return max(1, n), how, reported


def cpu_baseline(obj, seed, gpu_check=None):
    """CPU oracle (port of the reference algorithm, glibc sin/cos, FD Jacobian) on a bounded sample of the same
    workload: the first `sample` samples of the bench batch on the host cores this process can really use.  Threads
    take samples in dynamic chunks (oracle/ccmp_oracle.c: orc_run).  If the detected core count still does not scale
    (a share enforced some other way), the count is halved until a short probe reaches 75 % parallel efficiency."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import yaml
    from oracle_binding import Oracle

    O = Oracle("libm")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "golden", "config", obj + ".yaml")))
    P = O.problem(cfg)

    def rate(q, n):
        t0 = time.perf_counter()
        out = O.project_batch(P, q, n)
        return len(q) / (time.perf_counter() - t0), out

    rate(O.ambient_uniform_batch(P, seed, 0, 128), 1)  # page the library in, let the clocks ramp
    per_core = max(rate(O.ambient_uniform_batch(P, seed, 0, 512), 1)[0] for _ in range(2))
    cores, how, reported = usable_cores()
    probes = []
    while cores > 1:
        qp = O.ambient_uniform_batch(P, seed, 0, int(max(1024, per_core * cores * 1.5)))  # ~1.5 s per probe
        r = max(rate(qp, cores)[0] for _ in range(2))  # best of two: a probe must not lose cores to a noisy neighbour
        probes.append({"threads": cores, "efficiency": r / (cores * per_core)})
        if r >= 0.75 * cores * per_core:
            break
        cores, how = max(1, cores // 2), "scaling probe (detected count did not scale)"
    sample = int(max(1024, min(131072, per_core * cores * 15.0)))  # ~15 s of CPU work
    q = O.ambient_uniform_batch(P, seed, 0, sample)
    value, (q_libm, ok, it) = rate(q, cores)
    # the same algorithm with the analytic Jacobian on the CPU (fair algorithmic comparison for the fast mode)
    Pa = O.problem(cfg)
    Pa.jacobian_mode = 1
    t0 = time.perf_counter()
    O.project_batch(Pa, q, cores)
    wall_a = time.perf_counter() - t0
    # in-run parity evidence: the det build of the oracle vs the GPU on the first 8192 samples of the batch (< 1 s on 16 threads)
    parity, vs_libm = None, None
    if gpu_check is not None:
        m = 8192
        parity = det_parity(obj, gpu_check["problem"], gpu_check["q_in"][:m], gpu_check["q_out"][:m], gpu_check["ok"][:m],
                            gpu_check["iters"][:m], cores)
        # how far the GPU's results are from what the reference's libm gives — the glibc build timed
        # above, on the same leading samples of the batch.  (The GPU is bit-identical to the det build; this is the
        # distance between the two rounding models through ~33 Newton iterations)
        import numpy as np

        m = min(sample, len(gpu_check["q_out"]))
        d = np.abs(q_libm[:m] - gpu_check["q_out"][:m]).max(axis=1)
        di = it[:m].astype(np.int64) - gpu_check["iters"][:m].astype(np.int64)
        vs_libm = {"samples": int(m), "max_abs_dq": float(d.max()), "median_abs_dq": float(np.median(d)), "n_gt_1e-6": int((d > 1e-6).sum()),
                   "iteration_flips_pm1": int((np.abs(di) == 1).sum()), "iteration_diffs_gt1": int((np.abs(di) > 1).sum()),
                   "ok_mismatches": int((ok[:m] != gpu_check["ok"][:m]).sum())}
    return {
        "value": value, "unit": "projections/s", "cores": cores, "kind": "port",
        "sample": "first %d samples of the bench batch, FD-faithful C oracle (glibc libm, -O2), %d threads taking samples "
                  "in dynamic chunks; single thread: %.1f projections/s" % (sample, cores, per_core),
        "cores_reported_by_os": reported, "cores_found_by": how, "scaling_probes": probes,
        "single_thread_value": per_core, "parallel_efficiency": value / (cores * per_core),
        "analytic_jacobian_value": sample / wall_a, "mean_iters": float(it.mean()), "parity_gpu_vs_det_oracle": parity,
        "parity_gpu_vs_libm_oracle": vs_libm,
    }


def checker_problem(O, obj, product_problem):
    """The checker's problem: built by the ORACLE'S OWN set-up code from the YAML fixture of `obj` (never adopted from the
    product's struct), with the parameters this bench changes after loading — the two tolerances and the Jacobian switch —
    carried over as plain numbers, then compared with the product's problem byte for byte: the two set-up paths check
    each other in every in-run parity figure."""
    own = O.checker_problem(os.path.join(ROOT, "tests", "golden", "config", obj + ".yaml"))
    own.tol_pos, own.tol_rot, own.jacobian_mode = product_problem.tol_pos, product_problem.tol_rot, product_problem.jacobian_mode
    if bytes(own) != bytes(product_problem):
        raise RuntimeError("set-up mismatch: the product's problem for %s differs from the oracle's own" % obj)
    return own


def libm_problem(O, obj, product_problem):
    """The glibc build's problem for the CPU TIMING legs: its own set-up from the YAML fixture (its start frame is computed with
    glibc's sine/cosine, so its bytes legitimately differ from the product's in the last place — no comparison here), with the
    tolerances the bench has set."""
    import yaml

    own = O.problem(yaml.safe_load(open(os.path.join(ROOT, "tests", "golden", "config", obj + ".yaml"))))
    own.tol_pos, own.tol_rot, own.jacobian_mode = product_problem.tol_pos, product_problem.tol_rot, product_problem.jacobian_mode
    return own


def det_parity(obj, product_problem, q_in, q_out, ok, iters, threads):
    """the det build of the oracle on the very inputs the GPU projected"""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle

    Od = Oracle("det")
    Pd = checker_problem(Od, obj, product_problem)
    qd, okd, itd = Od.project_batch(Pd, q_in, threads)
    d = np.abs(qd - q_out).max(axis=1)
    return {"samples": len(q_in), "max_abs_dq": float(d.max()), "n_gt_1e-6": int((d > 1e-6).sum()),
            "bit_identical": bool(np.array_equal(qd.view(np.uint64), q_out.view(np.uint64))),
            "ok_mismatches": int((okd != ok).sum()), "iteration_mismatches": int((itd != iters).sum())}


def gather_probe(c, vg, rank, world, torch, dist, n=4096, seed=0xC5):
    """single-GPU and G-GPU outputs must be bit-identical.  After the timed region every rank projects
    its contiguous shard of an n-sample probe (sampler on the GLOBAL index), the shards' valid states are all-gathered
    through the bench's own blocks, and rank 0 compares them — count for count, bit for bit — with the compacted valid
    states of the whole probe projected by itself alone."""
    from closed_chain_motion_planner_amd.distributed import shard_range

    lo, hi = shard_range(n, rank, world)
    q, ok, _, _ = c.sample_project_batch(seed, lo, hi - lo, want_iters=False)
    c.compact_valid(q, ok, out=vg.rows, cnt=vg.count)
    vg.launch()
    states, counts = vg.unpack()
    res = {"samples": n, "valid_states": int(sum(counts)), "valid_per_rank": [int(v) for v in counts]}
    if rank == 0:
        qa, oka, _, _ = c.sample_project_batch(seed, 0, n, want_iters=False)
        alone = qa[oka == 1]
        res["bit_identical_to_one_gpu"] = bool(alone.shape == states.shape and torch.equal(alone.view(torch.int64), states.view(torch.int64)))
    return res


def main():
    args = parse()
    if args.one_process:
        return one_process_child(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args.gpus)  # before anything touches a GPU
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("error: --gpus %d but WORLD_SIZE=%d — launch with matching counts (or without WORLD_SIZE: "
                  "bench.py starts the ranks itself)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    if args.launch_check:  # launcher plumbing only: every rank joins a CPU rendezvous, nothing touches a GPU
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.tensor([float(rank)])
            dist.all_reduce(t)
            dist.destroy_process_group()
        else:
            t = torch.tensor([0.0])
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": float(t.item()),
                              "self_launched": os.environ.get("CCMP_BENCH_CHILD") == "1"}), flush=True)
        return

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

    from closed_chain_motion_planner_amd import CCMP_JAC_ANALYTIC, CCMP_JAC_FD, Context, KinematicChainConstraint
    from closed_chain_motion_planner_amd.build import build_library
    from closed_chain_motion_planner_amd.distributed import ValidGather

    # no-op when the in-tree .so is current; raises if it cannot be built.  One rank per node builds, the others wait:
    # eight ranks compiling into the same build directory would race.
    if world > 1:
        if local_rank == 0 or args.force_device >= 0 and rank == 0:
            build_library()
        dist.barrier()
    else:
        build_library()
    dev = local_rank if world > 1 else 0
    ctx = Context(dev)
    if args.waves_per_cu:
        ctx.set_waves_per_cu(args.waves_per_cu)
    cfg_path = lambda obj: os.path.join(ROOT, "tests", "golden", "config", obj + ".yaml")
    c = KinematicChainConstraint.from_yaml(cfg_path(args.obj), ctx=ctx)
    c.setJacobianMode(CCMP_JAC_FD if args.mode == "fd" else CCMP_JAC_ANALYTIC)
    if args.tol:
        c.setTolerance(*[float(v) for v in args.tol.split(",")])

    B = args.batch
    seed = SEEDS.get(B, 0xC3) if world == 1 else 0xC5
    # synthetic inputs, resident in HBM before the timed region: sampleUniform's ambient samples,
    # global index space sharded contiguously over ranks (rank r owns [r*B, (r+1)*B))
    q_in = c.ambient_uniform_batch(seed, rank * B, B)
    q_out = torch.empty_like(q_in)
    q_valid = torch.empty_like(q_in)
    cnt = torch.zeros(1, dtype=torch.int64, device=q_in.device)
    # N > 1: the compaction writes straight into the fixed-capacity send block of the all-gather (row 0 = count)
    vg = ValidGather(max(1, int(B * GATHER_CAPACITY)), q_in.device) if world > 1 else None
    torch.cuda.synchronize()

    kernel_ms = []

    def step(record):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _, ok, it = c.project_batch(q_in, out=q_out, want_iters=True)
        e1.record()
        if vg is None:
            c.compact_valid(q_out, ok, out=q_valid, cnt=cnt)
        else:
            c.compact_valid(q_out, ok, out=vg.rows, cnt=vg.count)
            vg.launch()  # one asynchronous collective behind this step's kernels; the next step's kernels do not wait for it
        if record:
            kernel_ms.append((e0, e1))
        return ok, it

    if world > 1:
        # communicator set-up (RCCL builds its rings on the first collective): not a step, so that --warmup 0 still
        # times steps and not the rendezvous
        vg.launch()
        vg.wait()
    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ok, it = step(True)
    e_done = torch.cuda.Event(enable_timing=True)
    if vg is not None:
        vg.wait()  # every step's all-gather belongs to the timed region
    e_done.record()  # behind the last collective on this stream: when the gathered states are there
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    local_elapsed = elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kms_steps = [a.elapsed_time(b) for a, b in kernel_ms]
    kms = sum(kms_steps) / max(1, len(kms_steps))
    sum_iters = float(it.to(torch.float64).sum().item())
    ok_frac = float(ok.to(torch.float64).mean().item())
    gathered = None
    if vg is not None:
        counts = vg.counts()  # after the timed region: what the host tree would read when it consumes the states
        n_valid = counts[rank]
        # what a first multi-GPU run is diagnosed by: every rank's own kernel time and wall clock, and how long the last
        # step's collective ran past the last projector kernel (one small all-gather, after the timed region)
        mine = torch.tensor([kms, local_elapsed * 1e3 / args.steps, kernel_ms[-1][1].elapsed_time(e_done)], dtype=torch.float64, device="cuda")
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank = torch.stack(per_rank).cpu()
        gathered = {"valid_states_all_ranks": int(sum(counts)), "capacity_rows_per_rank": vg.capacity,
                    "overflow": bool(max(counts) > vg.capacity), "bytes_sent_per_rank": int((vg.capacity + 1) * 14 * 8),
                    "bytes_received_per_rank": int(world * (vg.capacity + 1) * 14 * 8),
                    "kernel_ms_per_rank": [float(v) for v in per_rank[:, 0]],
                    "kernel_ms_min": float(per_rank[:, 0].min()), "kernel_ms_max": float(per_rank[:, 0].max()),
                    "step_ms_per_rank": [float(v) for v in per_rank[:, 1]],
                    "last_kernel_to_gather_done_ms_per_rank": [float(v) for v in per_rank[:, 2]],
                    "last_kernel_to_gather_done_ms_max": float(per_rank[:, 2].max()),
                    "probe": gather_probe(c, vg, rank, world, torch, dist)}
    else:
        n_valid = int(cnt.item())
    rccl_ranks = dist.get_world_size() if (world > 1 and args.backend == "nccl") else (1 if world == 1 else 0)

    if world > 1:
        dist.barrier()  # every rank is through its measurements: what follows on rank 0 has the GPUs to itself
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = world * B * args.steps / elapsed
    kernel = "project_fd_kernel+project_fd_flat_kernel+scout_kernel" if args.mode == "fd" else "project_pair_kernel+project_row16_kernel"
    achieved_gbs = BYTES_PER_PROJECTION * B / (kms * 1e-3) / 1e9
    traffic, valu, executed, profile_kernel = None, None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):  # PMC numbers come from separate rocprofv3 --pmc passes (tools/profile.sh), not from this run
        try:
            tj = json.load(open(tpath))
            if tj.get("kernel") == kernel and tj.get("batch") == B:
                traffic = tj.get("hbm_bytes_per_launch")
                if tj.get("kernel_median_ms"):  # the committed rocprofv3 summary's own figure for the dominant kernel (median over
                    # its profiled dispatches): must agree with what this run measures live (tests/test_bench_contract.py)
                    profile_kernel = {"ms": tj["kernel_median_ms"], "q1_ms": tj.get("kernel_q1_ms"), "q3_ms": tj.get("kernel_q3_ms"),
                                      "dispatches": tj.get("kernel_profiled_dispatches"), "kernel": tj.get("dominant_kernel"),
                                      "source": "profiles/%s_summary.md" % tj.get("tag")}
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
                    if tj.get("vector_pipes_busy_frac"):  # of that profiled launch, at the clock the chip held under it
                        valu["vector_pipes_busy_in_profiled_launch"] = tj["vector_pipes_busy_frac"]
                        valu["effective_clock_ghz_in_profiled_launch"] = tj.get("effective_clock_ghz")
        except Exception:
            traffic = None
    fp64_tflops = FLOP_PER_NEWTON_ITER * sum_iters / (kms * 1e-3) / 1e12
    line = {
        "metric": "constraint projections/sec (dual-Panda %s)" % args.obj,
        "value": value, "unit": "projections/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic", "rccl_ranks": rccl_ranks,
        "config": {
            "workload": "config/%s.yaml, batch=%d uniform joint samples per GPU (BASELINE configs[2]; per-GPU shard of "
                        "configs[4]), KinematicChainConstraint::project, tol (%g m, %g rad), cap 250" % (args.obj, B, c.problem.tol_pos, c.problem.tol_rot),
            "jacobian_mode": "finite-difference (reference arithmetic, bit-identical to the CPU oracle)"
            if args.mode == "fd" else "analytic (fast mode, not bit-comparable)",
            "global_batch": world * B, "parallelism": "sample-sharded x%d, one fixed-capacity all-gather of valid states per step (%s)"
                                                       % (world, args.backend if world > 1 else "none at N=1"),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_ratio": (traffic / (BYTES_PER_PROJECTION * B)) if traffic else None,
            "kernel": kernel, "kernel_ms": kms, "kernel_ms_median": sorted(kms_steps)[len(kms_steps) // 2] if kms_steps else None,
            "kernel_ms_per_step": kms_steps,
            "profile_kernel_ms": profile_kernel["ms"] if profile_kernel else None, "profile_kernel": profile_kernel,
            "note": "metric asks for %HBM; the kernel is FP64-VALU bound (see fp64); traffic above the algorithmic bytes comes from "
                    "scattered 112-B rows in longest-first order and partial-sector flag writes, harmless at this fraction of HBM",
            "fp64": {"achieved": fp64_tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": fp64_tflops / FP64_VECTOR_PEAK_TFLOPS,
                     "algorithmic_flop": "2400 per Newton iteration x %.0f iterations per launch" % sum_iters},
            "newton_iterations_per_s": sum_iters / (kms * 1e-3), "valu_issue": valu, "fp64_executed": executed,
        },
        "stats": {"ok_fraction": ok_frac, "mean_newton_iters": sum_iters / B, "valid_states_rank0": n_valid, "gather": gathered},
    }

    if world == 1 and not args.no_secondary:
        line["secondary"] = secondary(args, c, ctx, B, torch, cfg_path)

    if world == 1 and not args.no_cpu_baseline:
        try:
            gpu_check = None
            if args.mode == "fd" and not args.tol:
                m = min(B, 131072)  # the CPU leg's sample is at most this long
                gpu_check = {"problem": c.problem, "q_in": q_in[:m].cpu().numpy(), "q_out": q_out[:m].cpu().numpy(),
                             "ok": ok[:m].cpu().numpy(), "iters": it[:m].cpu().numpy().astype("int32")}
            line["cpu_baseline"] = cpu_baseline(args.obj, seed, gpu_check)
        except Exception as e:  # the oracle is a checker; its absence must not fail the GPU bench
            line["cpu_baseline"] = {"error": repr(e)}
    if world > 1:
        dist.destroy_process_group()
    if not args.no_secondary and args.mode == "fd":
        # the other multi-GPU form SURVEY.md section 8(e) names: ONE process driving all the GPUs through the C ABI (a fresh child,
        # started when every rank is done; with a deadline, and never allowed to take this line down with it)
        if world > 1:
            time.sleep(1.0)  # the other ranks are on their way out
        try:
            one = run_one_process(args, world)
        except Exception as e:
            one = {"error": repr(e)}
        line.setdefault("secondary", {})["one_process"] = one
    flatten_for_the_driver(line, B)
    print(json.dumps(line), flush=True)


def flatten_for_the_driver(line, B):
    """The driver's record of this line keeps only first-level SCALARS of `config`, `roofline` and `cpu_baseline`
    (VERDICT r3 #3): the figures SURVEY.md section 8(d) asks for — C1..C4, the extend step, single states, the three
    efficiency numbers, the parity classification — are therefore repeated there as flat keys (the nested forms stay)."""
    def get(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d

    rf, cfgd, sec, cb = line["roofline"], line["config"], line.get("secondary") or {}, line.get("cpu_baseline") or {}
    rf["fp64_algorithmic_frac"] = get(rf, "fp64", "frac")
    rf["fp64_executed_frac"] = get(rf, "fp64_executed", "frac")
    rf["valu_issue_frac"] = get(rf, "valu_issue", "frac_of_fp64_issue_ceiling")
    rf["pipes_busy_frac"] = get(rf, "valu_issue", "vector_pipes_busy_in_profiled_launch")
    rf["pmc_source"] = get(rf, "valu_issue", "source")  # builder-side counters rescaled by this run's kernel time
    if sec:
        st1, st2 = "stefan_batch%d_tol_1e-3_5e-3" % B, "stefan_batch%d_tol_5e-4_2.5e-3" % B
        geo = sec.get("discrete_geodesic") or {}
        flat = {
            "c1_dumbbell_cpu_single_thread_per_s": get(sec, "c1_dumbbell", "cpu_single_thread_projections_per_s"),
            "c1_dumbbell_gpu_per_s": get(sec, "c1_dumbbell", "gpu_projections_per_s"),
            "c1_bitwise": get(sec, "c1_dumbbell", "parity_vs_det_oracle", "bit_identical"),
            "c2_batch4096_per_s": sec.get("batch4096_projections_per_s"),
            "c2_bitwise": get(sec, "batch4096", "parity_vs_det_oracle", "bit_identical"),
            "batch32768_per_s": sec.get("batch32768_projections_per_s"),
            "c4_stefan_per_s": get(sec, st1, "projections_per_s"),
            "c4_stefan_tight_per_s": get(sec, st2, "projections_per_s"),
            "c4_bitwise": None if get(sec, st1, "parity_vs_det_oracle", "bit_identical") is None else bool(
                get(sec, st1, "parity_vs_det_oracle", "bit_identical") and get(sec, st2, "parity_vs_det_oracle", "bit_identical")),
            "extend_first_pass_edges_per_s": geo.get("edges_per_s"), "extend_first_pass_ms": geo.get("ms"),
            "extend_unfinished_edges": geo.get("overflowed_edges"), "extend_complete_ms": geo.get("complete_ms"),
            "extend_bitwise": None if get(geo, "parity_vs_det_oracle", "bit_identical") is None else bool(
                get(geo, "parity_vs_det_oracle", "bit_identical") and get(geo, "parity_vs_det_oracle", "continued_edges", "bit_identical")),
            "growtree_5_edges_ms": geo.get("growtree_5_edges_ms"),
            "extend_bulk_65536_edges_per_s": get(geo, "bulk", "edges_per_s"), "extend_bulk_65536_ms": get(geo, "bulk", "ms"),
            "extend_bulk_bitwise": get(geo, "bulk", "parity_vs_det_oracle", "bit_identical"),
            "extend_cpu_edges_per_s": get(geo, "cpu", "edges_per_s"), "extend_cpu_threads": get(geo, "cpu", "threads"),
            "growtree_5_edges_cpu_single_thread_ms": get(geo, "cpu", "growtree_5_edges_single_thread_ms"),
            "single_project_us": get(sec, "single_project_c_abi", "uniform_sample_median_us"),
            "single_project_near_manifold_us": get(sec, "single_project_c_abi", "near_manifold_median_us"),
            "single_project_cpu_us": get(sec, "single_project_c_abi", "uniform_sample_cpu_median_us"),
            "single_project_near_manifold_cpu_us": get(sec, "single_project_c_abi", "near_manifold_cpu_median_us"),
            "single_project_resident_us": get(sec, "single_project_c_abi", "uniform_sample_resident_median_us"),
            "single_project_near_manifold_resident_us": get(sec, "single_project_c_abi", "near_manifold_resident_median_us"),
            "single_is_satisfied_us": get(sec, "single_project_c_abi", "is_satisfied_median_us"),
            "single_is_satisfied_resident_us": get(sec, "single_project_c_abi", "is_satisfied_resident_median_us"),
            "single_function_us": get(sec, "single_project_c_abi", "function_median_us"),
            "single_function_resident_us": get(sec, "single_project_c_abi", "function_resident_median_us"),
            "single_resident_bitwise": get(sec, "single_project_c_abi", "resident_bit_identical_to_launched"),
            "one_edge_check_motion_us": get(sec, "single_project_c_abi", "one_edge_check_motion_median_us"),
            "one_edge_check_motion_resident_us": get(sec, "single_project_c_abi", "one_edge_check_motion_resident_median_us"),
            "host_buffer_pageable_per_s": get(sec, "host_buffer", "pageable", "projections_per_s"),
            "host_buffer_pinned_per_s": get(sec, "host_buffer", "pinned", "projections_per_s"),
            "c3_calibrated_per_s": get(sec, "c3_calibrated", "projections_per_s"),
            "c3_calibrated_bitwise": get(sec, "c3_calibrated", "parity_vs_det_oracle", "bit_identical"),
            "c3_calibrated_analytic_per_s": get(sec, "c3_calibrated", "analytic_projections_per_s"),
            "extend_calibrated_first_pass_edges_per_s": get(sec, "c3_calibrated", "extend_first_pass_edges_per_s"),
            "extend_calibrated_bitwise": get(sec, "c3_calibrated", "extend_parity_vs_det_oracle", "bit_identical"),
            "extend_analytic_edges_per_s": get(sec, "extend_analytic", "edges_per_s"),
            "extend_analytic_bitwise": get(sec, "extend_analytic", "parity_vs_det_oracle", "bit_identical"),
            "analytic_mode_per_s": sec.get("analytic_mode_projections_per_s"),
            "analytic_bitwise": get(sec, "analytic_mode", "parity_vs_det_oracle", "bit_identical"),
            "analytic_bitwise_samples": get(sec, "analytic_mode", "parity_vs_det_oracle", "samples"),
            "analytic_mode_2m_per_s": get(sec, "analytic_mode_batch2097152", "projections_per_s"),
            "analytic_mode_4096_per_s": get(sec, "analytic_mode_batch4096", "projections_per_s"),
            "proxy_clearance_states_per_s": get(sec, "proxy_clearance", "states_per_s"),
            "one_process_gpus": get(sec, "one_process", "gpus"),
            "one_process_direct_per_s": get(sec, "one_process", "direct", "projections_per_s"),
            "one_process_rccl_per_s": get(sec, "one_process", "rccl", "projections_per_s"),
        }
        cfgd.update(flat)
    if cb and "error" not in cb:
        cb["det_bit_identical"] = get(cb, "parity_gpu_vs_det_oracle", "bit_identical")
        cb["det_samples"] = get(cb, "parity_gpu_vs_det_oracle", "samples")
        cb["libm_samples"] = get(cb, "parity_gpu_vs_libm_oracle", "samples")
        cb["libm_n_gt_1e-6"] = get(cb, "parity_gpu_vs_libm_oracle", "n_gt_1e-6")
        cb["libm_max_abs_dq"] = get(cb, "parity_gpu_vs_libm_oracle", "max_abs_dq")
        cb["libm_iter_diffs_gt1"] = get(cb, "parity_gpu_vs_libm_oracle", "iteration_diffs_gt1")
        cb["libm_ok_mismatches"] = get(cb, "parity_gpu_vs_libm_oracle", "ok_mismatches")


def secondary(args, c, ctx, B, torch, cfg_path):
    """The other BASELINE configs and the callers' real shapes, timed with HIP events on the launch stream; every
    reference-arithmetic figure carries an in-run bitwise check of a 1024-sample slice against the det oracle."""
    import numpy as np
    from closed_chain_motion_planner_amd import CCMP_JAC_ANALYTIC, CCMP_JAC_FD, KinematicChainConstraint

    threads = usable_cores()[0]
    main_mode = CCMP_JAC_FD if args.mode == "fd" else CCMP_JAC_ANALYTIC
    other = CCMP_JAC_ANALYTIC if args.mode == "fd" else CCMP_JAC_FD

    def timed(fn, reps):
        # steady state, as the headline's warm-up steps give it: the chip drops its clocks within ~50 ms of idling (host-side
        # set-up between two secondaries is enough) and the first launches behind that run 7-10 % slower — measured with
        # tools/measure.py afterload: 4 096 samples 0.85 ms as the first ten launches after a pause, 0.77 ms as the next ten.
        # So: launches for ~10 ms (at least one, at most 32), then the timed ones back to back.
        t_w = time.perf_counter()
        for _ in range(32):
            fn()
            torch.cuda.synchronize()
            if time.perf_counter() - t_w > 0.010:
                break
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    def quick(con, mode, b, reps, check=False, obj=None):
        con.setJacobianMode(mode)
        qi = con.ambient_uniform_batch(SEEDS.get(b, 0xC3), 0, b)
        qo = torch.empty_like(qi)
        sec = timed(lambda: con.project_batch(qi, out=qo), reps)
        res = {"projections_per_s": b / sec, "ms": sec * 1e3}
        if check:  # reference arithmetic: against the det oracle; analytic mode: against the det oracle's analytic mode
            m = 1024 if mode == CCMP_JAC_FD else 4096
            _, ok, it = con.project_batch(qi, out=qo)
            try:
                res["parity_vs_det_oracle"] = det_parity(obj or args.obj, con.problem, qi[:m].cpu().numpy(), qo[:m].cpu().numpy(),
                                                         ok[:m].cpu().numpy(), it[:m].cpu().numpy().astype("int32"), threads)
            except Exception as e:
                res["parity_vs_det_oracle"] = {"error": repr(e)}
            res["mean_newton_iters"] = float(it.to(torch.float64).mean().item())
        return res

    def single_latency_us(reps=40):
        # the reference-signature call: one state through the host entry point (copy in, launch, synchronise, copy out)
        c.setJacobianMode(main_mode)
        qs = c.ambient_uniform_batch(0xC1, 0, reps).cpu().numpy()
        ts = []
        for i in range(reps):
            x = qs[i].copy()
            t0 = time.perf_counter()
            c.project(x)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts[4:]) * 1e6)

    def single_latency_c_abi_us(reps=64):
        # the same call as the C++ adapter makes it — ccmp_project_host with ready pointers, no Python in the timed region
        # but the ctypes call itself — from uniform samples and from states near the manifold (what the planner's samplers
        # and extend steps hand it: a valid state moved by up to 0.05 rad per joint)
        import ctypes as C
        from closed_chain_motion_planner_amd import _lib
        c.setJacobianMode(main_mode)
        L = _lib.lib()
        dp = C.POINTER(C.c_double)
        okb = (C.c_uint8 * 1)()
        far = c.ambient_uniform_batch(0xC1, 0, reps).cpu().numpy()
        q, ok, _ = c.project_batch(c.ambient_uniform_batch(0xC7, 0, 16 * reps))
        valid = q[ok == 1][:reps].cpu().numpy()
        rng = np.random.default_rng(0xC7)
        near = valid + rng.uniform(-0.05, 0.05, valid.shape)
        out = {}

        def time_calls(xs, resident):
            """median us of ccmp_project_host / ccmp_is_satisfied_host / ccmp_function_host over the states, results kept for the
            bitwise comparison of the two paths"""
            c.ctx.set_option("resident", 1 if resident else 0)
            tp, ts_, tf, its, res = [], [], [], [], []
            fb = (C.c_double * 2)()
            for i in range(xs.shape[0]):
                xi, xo = np.ascontiguousarray(xs[i]), np.zeros(14)
                it = (C.c_uint16 * 1)()
                a, b = xi.ctypes.data_as(dp), xo.ctypes.data_as(dp)
                t0 = time.perf_counter()
                L.ccmp_project_host(c.ctx.handle, C.byref(c.problem), a, b, okb, it, 1)
                t1 = time.perf_counter()
                L.ccmp_is_satisfied_host(c.ctx.handle, C.byref(c.problem), b, okb, 1)
                t2 = time.perf_counter()
                L.ccmp_function_host(c.ctx.handle, C.byref(c.problem), a, fb, 1)
                t3 = time.perf_counter()
                tp.append(t1 - t0), ts_.append(t2 - t1), tf.append(t3 - t2)
                its.append(int(it[0]))
                res.append((xo.tobytes(), int(it[0]), int(okb[0]), bytes(fb)))
            c.ctx.set_option("resident", 0)
            med = lambda v: float(np.median(v[8:]) * 1e6)
            return med(tp), med(ts_), med(tf), float(np.median(its[8:])), res

        same = True
        for name, xs in (("uniform_sample", far), ("near_manifold", near)):
            p0, s0, f0, its, r0 = time_calls(xs, False)
            out[name + "_median_us"] = p0
            out[name + "_median_newton_iters"] = its
            # the opt-in resident service kernel (ccmp_ctx_set_option "resident"): the same calls with no launch on the call path
            p1, s1, f1, _, r1 = time_calls(xs, True)
            out[name + "_resident_median_us"] = p1
            same = same and r0 == r1
            if name == "near_manifold":
                out.update(is_satisfied_median_us=s0, function_median_us=f0, is_satisfied_resident_median_us=s1, function_resident_median_us=f1)
        # ONE edge of checkMotion — isSatisfied(to) && discreteGeodesic(from, to), as the unchanged planner asks for them, one pair at a
        # time (src/planner/stefanBiPRM.cpp:397-398; the adapter's ccmp_geodesic_host_ex with E == 1): launched and through the service
        tos = valid + rng.uniform(-0.25, 0.25, valid.shape)  # a few delta-steps away, like growTree's neighbours
        for i in range(tos.shape[0]):
            c.project(tos[i])
        st, nb, cb = np.zeros((64, 14)), (C.c_int32 * 1)(), np.zeros(2)
        edge_res = {}
        for on in (0, 1):
            c.ctx.set_option("resident", on)
            ts, rs = [], []
            for i in range(valid.shape[0]):
                a, b = np.ascontiguousarray(valid[i]), np.ascontiguousarray(tos[i])
                t0 = time.perf_counter()
                L.ccmp_geodesic_host_ex(c.ctx.handle, C.byref(c.problem), a.ctypes.data_as(dp), b.ctypes.data_as(dp), 1, 64, st.ctypes.data_as(dp), nb, okb, None,
                                        cb.ctypes.data_as(dp), 0, 1)
                ts.append(time.perf_counter() - t0)
                rs.append((int(nb[0]), int(okb[0]), st[: min(int(nb[0]), 64)].tobytes()))
            edge_res[on] = (float(np.median(ts[8:]) * 1e6), rs)
        c.ctx.set_option("resident", 0)
        out["one_edge_check_motion_median_us"] = edge_res[0][0]
        out["one_edge_check_motion_resident_median_us"] = edge_res[1][0]
        out["one_edge_median_states"] = float(np.median([r[0] for r in edge_res[0][1]]))
        same = same and edge_res[0][1] == edge_res[1][1]
        out["resident_bit_identical_to_launched"] = bool(same)
        try:
            # the CPU path beside it: the same states through the glibc build of the oracle, one call at a time on one thread
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle_binding import Oracle

            Ol = Oracle("libm")
            Pl = libm_problem(Ol, args.obj, c.problem)
            for name, xs in (("uniform_sample", far), ("near_manifold", near)):
                ts = []
                for i in range(xs.shape[0]):
                    t0 = time.perf_counter()
                    Ol.project(Pl, xs[i])
                    ts.append(time.perf_counter() - t0)
                out[name + "_cpu_median_us"] = float(np.median(ts[8:]) * 1e6)
        except Exception as e:
            out["cpu_error"] = repr(e)
        return out

    def geodesic(n_edges=16384, first_pass=16, budget=128):
        # growTree-shaped edges (src/planner/stefanBiPRM.cpp:307-351: a milestone towards a near neighbour): from a valid
        # projected state to a projected sampleUniformNear state within 0.6 rad per joint (about 1.0 rad apart, ~4 states
        # per edge like the reference's recorded roadmap edges).  What is timed: the call with lists of
        # `first_pass` states and a budget of `budget` Newton rounds per edge — `edges_per_s` counts only the edges that are
        # COMPLETE after it, `overflowed_edges` says how many were not (list full or budget spent) — and the whole operation
        # (`complete_*`): the same call plus the continuation of every such edge, from its last stored state, until each
        # list is whole (one edge in 16384 creeps for 952 states: a serial chain of ~12 000 Newton rounds that the
        # reference pays as well).  `ms_unbounded_rounds` is the same call without the round budget: it lasts as long as
        # that one edge needs for its first 15 states (545 rounds).
        c.setJacobianMode(CCMP_JAC_FD)
        q, ok, _, _ = c.sample_project_batch(0x6E0, 0, 8 * n_edges, want_iters=False)
        frm = q[ok == 1][:n_edges].contiguous()
        to, _, _, _ = c.sample_near_project_batch(0x6E1, 0, frm, 0.6, n_edges, want_iters=False)
        call = lambda: c.discrete_geodesic_batch(frm, to, first_pass, want_carry=True, round_budget=budget)
        sec = timed(call, 5)
        sec_nb = timed(lambda: c.discrete_geodesic_batch(frm, to, first_pass), 3)
        sec64 = timed(lambda: c.discrete_geodesic_batch(frm, to, 64), 3)
        st, n, gok, its, carry = call()
        unfinished = (n > first_pass) | (gok == 2)
        over = int(unfinished.sum().item())

        def complete():
            r = call()
            # the continuation as the callers run it (space.discreteGeodesicBatch, ccmp::discreteGeodesicBatch): lists of 64, no round bound
            return r, c.continue_geodesics(to, r[0], r[1], r[2], r[3], r[4], first_pass, cont_states=64)

        complete()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, whole = complete()
        torch.cuda.synchronize()
        sec_all = time.perf_counter() - t0
        nn = n.clone()
        for e, (st_e, _, _) in whole.items():
            nn[e] = st_e.shape[0]
        # the planner's neighbour loop: five edges in one call (lists of 64 states, as the adapter asks for)
        f5, t5 = frm[:5].contiguous(), to[:5].contiguous()
        sec5 = timed(lambda: c.discrete_geodesic_batch(f5, t5, 64), 20)
        done = ~unfinished
        res = {"edges_per_s": (n_edges - over) / sec, "ms": sec * 1e3, "edges": n_edges, "max_states_first_pass": first_pass,
               "newton_round_budget_per_edge": budget, "overflowed_edges": over,
               "overflowed_list_full": int((n > first_pass).sum().item()), "overflowed_budget_spent": int((gok == 2).sum().item()),
               "complete_ms": sec_all * 1e3, "complete_edges_per_s": n_edges / sec_all,
               "longest_edge_states": int(nn.max().item()), "ms_unbounded_rounds": sec_nb * 1e3, "ms_lists_of_64": sec64 * 1e3,
               "edges_per_s_lists_of_64": (n_edges - int((nn > 64).sum().item())) / sec64,
               "growtree_5_edges_ms": sec5 * 1e3,
               "mean_states_per_edge": float(nn.to(torch.float64).mean().item()),
               "mean_newton_iters_per_edge_first_pass": float(its.to(torch.float64).mean().item()),
               "reached_fraction_of_complete_edges": float((gok[done] == 1).to(torch.float64).mean().item())}
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle_binding import Oracle

            Od = Oracle("det")
            Pd = checker_problem(Od, args.obj, c.problem)
            m = 1024
            sc, nc, okc, itc = Od.discrete_geodesic_batch(Pd, frm[:m].cpu().numpy(), to[:m].cpu().numpy(), first_pass, threads)
            sg, ng, og, ig = st[:m].cpu().numpy(), n[:m].cpu().numpy(), gok[:m].cpu().numpy(), its[:m].cpu().numpy()
            live = og != 2  # an edge suspended by the budget is compared through its continuation below
            same = all(np.array_equal(sg[e, : min(ng[e], first_pass)].view(np.uint64), sc[e, : min(nc[e], first_pass)].view(np.uint64))
                       for e in range(m) if live[e])
            res["parity_vs_det_oracle"] = {"edges": int(live.sum()), "bit_identical": bool(same and np.array_equal(ng[live], nc[live])),
                                           "flag_mismatches": int((og[live] != okc[live]).sum()),
                                           "iteration_mismatches": int((ig[live] != itc[live]).sum())}
            # continued edges against the oracle's uninterrupted traversal (all but the creeper: <= 256 states)
            cand = [e for e in whole if whole[e][0].shape[0] <= 256]
            okc_all = True
            for e in cand:
                okf, stf, itf = Od.discrete_geodesic(Pd, frm[e].cpu().numpy(), to[e].cpu().numpy(), interpolate=True, max_states=512)
                okc_all = okc_all and bool(stf.shape == whole[e][0].shape and np.array_equal(
                    np.ascontiguousarray(whole[e][0]).view(np.uint64), stf.view(np.uint64)) and bool(whole[e][1]) == okf and whole[e][2] == itf)
            res["parity_vs_det_oracle"]["continued_edges"] = {"edges": len(cand), "bit_identical": bool(okc_all)}
        except Exception as e:
            res["parity_vs_det_oracle"] = {"error": repr(e)}
        try:
            # a roadmap built in bulk: 65536 edges in one call of the same shape — from 40960 edges on the short edges run ten to a
            # wavefront on the throughput layout beside latency blocks for the long ones; a slice against the oracle
            nb = 65536
            qb, okb, _, _ = c.sample_project_batch(0x6E2, 0, 8 * nb, want_iters=False)
            fb = qb[okb == 1][:nb].contiguous()
            tb, _, _, _ = c.sample_near_project_batch(0x6E3, 0, fb, 0.6, nb, want_iters=False)
            bulk = lambda: c.discrete_geodesic_batch(fb, tb, first_pass, want_carry=True, round_budget=budget)
            secb = timed(bulk, 5)
            sb, nbs, okbs, itb, _ = bulk()
            unfin = int(((nbs > first_pass) | (okbs == 2)).sum().item())
            res["bulk"] = {"edges": nb, "ms": secb * 1e3, "edges_per_s": (nb - unfin) / secb, "unfinished_edges": unfin}
            Od2 = Oracle("det")
            Pd2 = checker_problem(Od2, args.obj, c.problem)
            m2 = 512
            sc2, nc2, okc2, itc2 = Od2.discrete_geodesic_batch(Pd2, fb[:m2].cpu().numpy(), tb[:m2].cpu().numpy(), first_pass, threads)
            sg2, ng2, og2, ig2 = sb[:m2].cpu().numpy(), nbs[:m2].cpu().numpy(), okbs[:m2].cpu().numpy(), itb[:m2].cpu().numpy()
            live2 = og2 != 2
            same2 = all(np.array_equal(sg2[e, : min(ng2[e], first_pass)].view(np.uint64), sc2[e, : min(nc2[e], first_pass)].view(np.uint64))
                        for e in range(m2) if live2[e])
            res["bulk"]["parity_vs_det_oracle"] = {"edges": int(live2.sum()), "bit_identical": bool(
                same2 and np.array_equal(ng2[live2], nc2[live2]) and np.array_equal(og2[live2], okc2[live2]) and np.array_equal(ig2[live2], itc2[live2]))}
        except Exception as e:
            res["bulk"] = {"error": repr(e)}
        try:
            # the CPU path beside it: the glibc build of the oracle — what the reference's loop would call — on
            # the first 4096 of the same edges with every usable core (lists of 64), and growTree's five edges on one thread
            Ol = Oracle("libm")
            Pl = libm_problem(Ol, args.obj, c.problem)
            mc = 4096
            fc, tc = frm[:mc].cpu().numpy(), to[:mc].cpu().numpy()
            t0 = time.perf_counter()
            Ol.discrete_geodesic_batch(Pl, fc, tc, 64, threads)
            cpu_sec = time.perf_counter() - t0
            best5 = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                Ol.discrete_geodesic_batch(Pl, fc[:5], tc[:5], 64, 1)
                best5 = min(best5, time.perf_counter() - t0)
            res["cpu"] = {"kind": "port (glibc build of the oracle)", "edges": mc, "threads": threads, "edges_per_s": mc / cpu_sec,
                          "growtree_5_edges_single_thread_ms": best5 * 1e3}
        except Exception as e:
            res["cpu"] = {"error": repr(e)}
        return res

    def host_buffer():
        # SURVEY.md §8d "report both": the reference's entry is host memory (an Eigen::Ref over the OMPL state's values,
        # src/base/jy_ProjectedStateSpace.cpp:10-15) — ccmp_project_host on the bench batch: upload, project, download.
        # Never `value`.
        import ctypes as C
        from closed_chain_motion_planner_amd import _lib

        c.setJacobianMode(main_mode)
        L = _lib.lib()
        qh = c.ambient_uniform_batch(SEEDS.get(B, 0xC3), 0, B).cpu()
        res = {"samples": B}
        # ONE pageable and ONE page-locked set of buffers for every variant, variants interleaved over two rounds, best median
        # per variant: a freshly pinned allocation is slow on its first uses (the first pinned variant measured read 1 ms worse
        # than the second, whichever it was)
        bufs = {}
        for pin in (False, True):
            mk = (lambda t: t.pin_memory()) if pin else (lambda t: t)
            bufs[pin] = (mk(qh.clone()), mk(torch.empty_like(qh)), mk(torch.empty(B, dtype=torch.uint8)), mk(torch.empty(B, dtype=torch.int16)))

        def run(pin):
            qi, qo, okh, ith = bufs[pin]
            rc = L.ccmp_project_host(ctx.handle, C.byref(c.problem), C.cast(qi.data_ptr(), C.POINTER(C.c_double)),
                                     C.cast(qo.data_ptr(), C.POINTER(C.c_double)), C.cast(okh.data_ptr(), C.POINTER(C.c_uint8)),
                                     C.cast(ith.data_ptr(), C.POINTER(C.c_uint16)), B)
            if rc != 0:
                raise RuntimeError("ccmp_project_host: %d" % rc)

        # page-locked caller buffers ("host_zero_copy", include/ccmp.h): the kernels read q_in and write q_out in place (default,
        # "pinned"), q_in is uploaded by one copy first, or everything is staged as for pageable memory
        variants = (("pageable", False, 2), ("pinned", True, 2), ("pinned_q_in_uploaded_first", True, 1), ("pinned_staged", True, 0))
        best = {}
        for rnd in range(2):
            for name, pin, zc in variants:
                ctx.set_option("host_zero_copy", zc)
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    run(pin)
                    ts.append(time.perf_counter() - t0)
                sec = float(np.median(ts[1:]))
                best[name] = min(best.get(name, 1e9), sec)
        for name, sec in best.items():
            res[name] = {"projections_per_s": B / sec, "ms": sec * 1e3}
        ctx.set_option("host_zero_copy", 2)
        return res

    def c1_dumbbell(b=1024, seed=0xC1):
        # BASELINE configs[0]: config/dumbbell.yaml on a single CPU thread (plumbing: YAML -> problem -> oracle), with the
        # GPU beside it on the same 1024 samples, bit for bit against the det oracle
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import yaml
        from oracle_binding import Oracle

        db = KinematicChainConstraint.from_yaml(cfg_path("dumbbell"), ctx=ctx)
        db.setJacobianMode(CCMP_JAC_FD)
        Ol = Oracle("libm")
        Pl = Ol.problem(yaml.safe_load(open(cfg_path("dumbbell"))))
        qi = db.ambient_uniform_batch(seed, 0, b)
        qn = qi.cpu().numpy()  # the CPU projects the very samples the GPU does
        Ol.project_batch(Pl, qn[:32], 1)
        t0 = time.perf_counter()
        _, okl, itl = Ol.project_batch(Pl, qn, 1)
        cpu_sec = time.perf_counter() - t0
        qo = torch.empty_like(qi)
        sec = timed(lambda: db.project_batch(qi, out=qo), 10)
        _, okg, itg = db.project_batch(qi, out=qo)
        res = {"samples": b, "cpu_single_thread_projections_per_s": b / cpu_sec, "cpu_threads": 1, "cpu_kind": "port (glibc build of the oracle)",
               "cpu_mean_newton_iters": float(itl.mean()), "cpu_ok_fraction": float(okl.mean()),
               "gpu_projections_per_s": b / sec, "gpu_ms": sec * 1e3, "same_ambient_samples_on_both_sides": True,
               "gpu_ok_fraction": float(okg.to(torch.float64).mean().item())}
        try:
            res["parity_vs_det_oracle"] = det_parity("dumbbell", db.problem, qi.cpu().numpy(), qo.cpu().numpy(), okg.cpu().numpy(),
                                                     itg.cpu().numpy().astype("int32"), threads)
        except Exception as e:
            res["parity_vs_det_oracle"] = {"error": repr(e)}
        return res

    def proxy_clearance():
        # the pre-filter ahead of the host's MoveIt test: default skeleton spheres + sub_table on the
        # states a sampleUniform batch produces, chained behind the projection's flags
        from closed_chain_motion_planner_amd.scene import ProxyValidityChecker

        c.setJacobianMode(CCMP_JAC_FD)
        chk = ProxyValidityChecker(c)
        sc = chk.scene
        q, ok, _, _ = c.sample_project_batch(0xC1EA, 0, B, want_iters=False)
        sec = timed(lambda: sc.clearance_batch(q, chk.margin, ok=ok, want_pair=True), 10)
        clr, pair, free = sc.clearance_batch(q, chk.margin, ok=ok)
        res = {"states_per_s": B / sec, "ms": sec * 1e3, "states": B, "spheres": len(sc.spheres), "boxes": len(sc.boxes),
               "pairs_per_state": sc.num_pairs, "pair_tests_per_s": B * sc.num_pairs / sec, "reject_below_m": chk.margin,
               "kept_fraction_of_valid": float(free.to(torch.float64).sum().item() / max(1.0, ok.to(torch.float64).sum().item()))}
        xs = q[:3].cpu().numpy()
        ts = []
        for i in range(24):
            t0 = time.perf_counter()
            sc.clearance(xs[i % 3])
            ts.append(time.perf_counter() - t0)
        res["single_state_call_median_us"] = float(np.median(ts[4:]) * 1e6)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle_binding import Oracle

            Od = Oracle("det")
            Pd = checker_problem(Od, args.obj, c.problem)
            m = 2048
            co, po = Od.clearance_batch(Pd, sc.spheres, sc.boxes, sc.allowed, q[:m].cpu().numpy())
            cg = clr[:m].cpu().numpy()
            res["parity_vs_det_oracle"] = {"states": m, "bit_identical": bool(np.array_equal(cg.view(np.uint64), co.view(np.uint64))
                                                                                 and np.array_equal(pair[:m].cpu().numpy(), po))}
        except Exception as e:
            res["parity_vs_det_oracle"] = {"error": repr(e)}
        return res

    def c3_calibrated():
        # C3 on CALIBRATED arms: PandaModel::initModel(dh) with DH offsets that differ per arm (src/kinematics/panda_rbdl.cpp:80-99;
        # the reference has the call commented out at ConstrainedPlanningCommon.cpp:97) — the kernels' general instantiations
        # (no exact zeros to skip, two sets of chain constants), checked against the det oracle on the problem's own bytes
        import ctypes as C
        from closed_chain_motion_planner_amd import _lib

        cal = KinematicChainConstraint.from_yaml(cfg_path(args.obj), ctx=ctx)
        for arm in (0, 1):
            dh = (C.c_double * 28)(*[(1e-3 if arm == 0 else -7e-4) * ((5 * i + 3 * arm) % 7 - 3) for i in range(28)])
            if _lib.lib().ccmp_set_calibration(C.byref(cal.problem), arm, dh) != 0:
                raise RuntimeError("ccmp_set_calibration failed")
        cal.setJacobianMode(CCMP_JAC_FD)
        qi = cal.ambient_uniform_batch(0xC3, 0, B)
        qo = torch.empty_like(qi)
        sec = timed(lambda: cal.project_batch(qi, out=qo), 3)
        _, ok, it = cal.project_batch(qi, out=qo)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_binding import Oracle

        Od = Oracle("det")
        Pd = Od.problem_from_bytes(bytes(cal.problem))  # a problem this bench modified after loading: adopted as bytes
        m = 2048
        qd, okd, itd = Od.project_batch(Pd, qi[:m].cpu().numpy(), threads)
        res = {"projections_per_s": B / sec, "ms": sec * 1e3, "mean_newton_iters": float(it.to(torch.float64).mean().item()),
               "parity_vs_det_oracle": {"samples": m, "bit_identical": bool(np.array_equal(qd.view(np.uint64), qo[:m].cpu().numpy().view(np.uint64))
                                                                                 and np.array_equal(okd, ok[:m].cpu().numpy())
                                                                                 and np.array_equal(itd, it[:m].cpu().numpy().astype("int32")))}}
        cal.setJacobianMode(CCMP_JAC_ANALYTIC)
        res["analytic_projections_per_s"] = B / timed(lambda: cal.project_batch(qi, out=qo), 3)
        # the extend step on the same model (geodesic_flat_kernel<false> + geodesic_group_kernel<false>): the first pass of the
        # growTree-shaped workload above, 16 384 edges, lists of 16 states, 128 Newton rounds per edge; 256 edges against the oracle
        cal.setJacobianMode(CCMP_JAC_FD)
        ne = 16384
        q2, ok2, _, _ = cal.sample_project_batch(0x6E0, 0, 8 * ne, want_iters=False)
        frm = q2[ok2 == 1][:ne].contiguous()
        to, _, _, _ = cal.sample_near_project_batch(0x6E1, 0, frm, 0.6, ne, want_iters=False)
        call = lambda: cal.discrete_geodesic_batch(frm, to, 16, want_carry=True, round_budget=128)
        sec_g = timed(call, 5)
        st, n, gok, its, _ = call()
        done = int(((n <= 16) & (gok != 2)).sum().item())
        res["extend_first_pass_edges_per_s"] = done / sec_g
        res["extend_first_pass_ms"] = sec_g * 1e3
        m = 256
        sc, nc, okc, itc = Od.discrete_geodesic_batch(Pd, frm[:m].cpu().numpy(), to[:m].cpu().numpy(), 16, threads)
        same = 0
        st_h, n_h, ok_h = st[:m].cpu().numpy(), n[:m].cpu().numpy(), gok[:m].cpu().numpy()
        for e in range(m):
            if ok_h[e] == 2:  # the round budget was spent: the oracle's traversal (no budget) is longer, its prefix must agree
                k = min(int(n_h[e]), 16)
                same += int(np.array_equal(st_h[e, :k].view(np.uint64), sc[e, :k].view(np.uint64)))
                continue
            k = min(int(n_h[e]), 16)
            same += int(n_h[e] == nc[e] and bool(ok_h[e]) == bool(okc[e]) and np.array_equal(st_h[e, :k].view(np.uint64), sc[e, :k].view(np.uint64)))
        res["extend_parity_vs_det_oracle"] = {"edges": m, "bit_identical": bool(same == m)}
        return res

    out = {}
    s4 = quick(c, main_mode, 4096, 10, check=True)
    s32 = quick(c, main_mode, 32768, 10)
    out["batch4096_projections_per_s"] = s4["projections_per_s"]
    out["batch4096"] = s4
    out["batch32768_projections_per_s"] = s32["projections_per_s"]
    out["single_project_call_median_us"] = single_latency_us()
    out["single_project_c_abi"] = single_latency_c_abi_us()
    om = "analytic" if args.mode == "fd" else "fd"
    so = quick(c, other, B, 5, check=True)
    out[om + "_mode_projections_per_s"] = so["projections_per_s"]
    out[om + "_mode"] = so
    if other == CCMP_JAC_ANALYTIC:  # the fast mode where its tail is amortised (C3's batch ends on the serial chain of its longest samples) and at C2's size
        out["analytic_mode_batch2097152"] = quick(c, other, 2097152, 3)
        out["analytic_mode_batch4096"] = quick(c, other, 4096, 10)
    c.setJacobianMode(main_mode)
    # BASELINE configs[3]: stefan (arms left + top), the reference's tolerances and the tighter set the baseline asks for
    st = KinematicChainConstraint.from_yaml(cfg_path("stefan"), ctx=ctx)
    out["stefan_batch%d_tol_1e-3_5e-3" % B] = quick(st, CCMP_JAC_FD, B, 3, check=True, obj="stefan")
    st.setTolerance(5e-4, 2.5e-3)
    out["stefan_batch%d_tol_5e-4_2.5e-3" % B] = quick(st, CCMP_JAC_FD, B, 3, check=True, obj="stefan")
    out["discrete_geodesic"] = geodesic()

    def extend_analytic(n_edges=16384, first_pass=16):
        # the same growTree-shaped workload through the extend step in ANALYTIC mode (a step loop around the batched analytic
        # projector): lists of 16 states, no round budget in that mode; 256 edges against the oracle's analytic traversal
        c.setJacobianMode(CCMP_JAC_FD)
        q, ok, _, _ = c.sample_project_batch(0x6E0, 0, 8 * n_edges, want_iters=False)
        frm = q[ok == 1][:n_edges].contiguous()
        to, _, _, _ = c.sample_near_project_batch(0x6E1, 0, frm, 0.6, n_edges, want_iters=False)
        c.setJacobianMode(CCMP_JAC_ANALYTIC)
        try:
            call = lambda: c.discrete_geodesic_batch(frm, to, first_pass)
            sec = timed(call, 5)
            st, n, gok, its = call()
            res = {"edges_per_s": int((n <= first_pass).sum().item()) / sec, "ms": sec * 1e3, "edges": n_edges, "overflowed_edges": int((n > first_pass).sum().item())}
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle_binding import Oracle

            Od = Oracle("det")
            Pd = checker_problem(Od, args.obj, c.problem)
            m = 256
            sc, nc, okc, itc = Od.discrete_geodesic_batch(Pd, frm[:m].cpu().numpy(), to[:m].cpu().numpy(), first_pass, threads)
            st_h, n_h = st[:m].cpu().numpy(), n[:m].cpu().numpy()
            live = np.arange(first_pass)[None, :] < np.minimum(n_h, first_pass)[:, None]
            res["parity_vs_det_oracle"] = {"edges": m, "bit_identical": bool(np.array_equal(n_h, nc) and np.array_equal(gok[:m].cpu().numpy(), okc)
                                                                                 and np.array_equal(its[:m].cpu().numpy(), itc)
                                                                                 and np.array_equal(st_h[live].view(np.uint64), sc[live].view(np.uint64)))}
            return res
        finally:
            c.setJacobianMode(main_mode)

    out["proxy_clearance"] = proxy_clearance()
    for name, fn in (("host_buffer", host_buffer), ("c1_dumbbell", c1_dumbbell), ("c3_calibrated", c3_calibrated), ("extend_analytic", extend_analytic)):
        try:
            out[name] = fn()
        except Exception as e:  # secondaries must not take the headline line down
            out[name] = {"error": repr(e)}
    c.setJacobianMode(main_mode)
    return out


if __name__ == "__main__":
    main()
