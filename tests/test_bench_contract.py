"""bench.py's one-line JSON contract (the driver parses it): keys, types, and the roofline / cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def test_bench_help_needs_no_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


@pytest.mark.gpu
def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16384"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    for k, t in REQUIRED.items():
        assert k in j, k
        assert isinstance(j[k], t) or (t is float and isinstance(j[k], int)), (k, type(j[k]))
    assert "vs_baseline" in j and j["vs_baseline"] is None  # BASELINE.md holds no published number
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["dtype"] == "f64" and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    cb = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1
    assert cb["parity_gpu_vs_det_oracle"]["bit_identical"] is True
    # SURVEY.md §8d's in-run classification against the glibc build (what the reference's libm gives), >= 4096 samples
    pl = cb["parity_gpu_vs_libm_oracle"]
    assert pl["samples"] >= 4096 and set(pl) >= {"n_gt_1e-6", "iteration_flips_pm1", "ok_mismatches", "max_abs_dq"}
    assert 0 < pl["n_gt_1e-6"] < 0.5 * pl["samples"] and pl["ok_mismatches"] < 0.05 * pl["samples"]  # ~20 % / ~0.5 %
    assert j["value"] > 1e6  # the north star's floor, even at this small batch
    sec = j["secondary"]
    hb = sec["host_buffer"]  # §8d "report both": the host-buffer (PCIe-inclusive) rate, never `value`
    assert hb["pageable"]["projections_per_s"] > 1e5 and hb["pinned"]["projections_per_s"] > 1e5
    c1 = sec["c1_dumbbell"]  # BASELINE configs[0]
    assert c1["samples"] == 1024 and c1["cpu_threads"] == 1 and c1["cpu_single_thread_projections_per_s"] > 50
    assert c1["parity_vs_det_oracle"]["bit_identical"] is True and c1["same_ambient_samples_on_both_sides"] is True
    sp = sec["single_project_c_abi"]  # ccmp_project_host as the C++ adapter calls it: from a uniform sample and near the manifold
    assert 10 < sp["near_manifold_median_us"] < sp["uniform_sample_median_us"] < 1000
    # the opt-in resident service kernel beside it: the same calls, the same bits, no launch on the call path
    assert sp["resident_bit_identical_to_launched"] is True and sp["near_manifold_resident_median_us"] < sp["near_manifold_median_us"]
    assert sp["is_satisfied_resident_median_us"] < sp["is_satisfied_median_us"] and sp["function_resident_median_us"] < sp["function_median_us"]
    assert sp["near_manifold_median_newton_iters"] < sp["uniform_sample_median_newton_iters"]
    g = sec["discrete_geodesic"]  # the complete operation is timed, overflowing edges are counted
    assert set(g) >= {"edges_per_s", "overflowed_edges", "complete_ms", "complete_edges_per_s", "growtree_5_edges_ms", "max_states_first_pass"}
    assert g["parity_vs_det_oracle"]["bit_identical"] is True and g["overflowed_edges"] < 0.01 * g["edges"]
    assert g["parity_vs_det_oracle"]["continued_edges"]["bit_identical"] is True and g["parity_vs_det_oracle"]["continued_edges"]["edges"] >= 1
    assert g["overflowed_edges"] == g["overflowed_list_full"] + g["overflowed_budget_spent"] and g["ms_unbounded_rounds"] > 0
    # the CPU path timed beside the latency rows: slower than the GPU on the same work
    assert g["cpu"]["edges_per_s"] < g["edges_per_s"] and g["cpu"]["growtree_5_edges_single_thread_ms"] > g["growtree_5_edges_ms"]
    assert sp["uniform_sample_cpu_median_us"] > sp["uniform_sample_median_us"]
    # the driver's record keeps first-level scalars of config / roofline / cpu_baseline only — the figures of
    # SURVEY.md section 8(d) are repeated there, flat
    for k in FLAT_CONFIG:
        assert k in j["config"] and isinstance(j["config"][k], (int, float, bool)), k
    for k in ("c1_bitwise", "c2_bitwise", "c4_bitwise", "extend_bitwise", "extend_bulk_bitwise"):
        assert j["config"][k] is True, k
    assert j["config"]["c2_batch4096_per_s"] == sec["batch4096_projections_per_s"] and j["config"]["extend_complete_ms"] == g["complete_ms"]
    for k in ("fp64_algorithmic_frac",):
        assert isinstance(r[k], float) and 0 < r[k] < 1
    for k in ("fp64_executed_frac", "valu_issue_frac", "pipes_busy_frac"):  # PMC-derived: None unless profiles/traffic_latest.json is for this batch
        assert k in r and (r[k] is None or 0 < r[k] < 1.5)
    for k in FLAT_CPU:
        assert k in cb and isinstance(cb[k], (int, float, bool)), k
    assert cb["det_bit_identical"] is True and cb["libm_n_gt_1e-6"] == pl["n_gt_1e-6"] and cb["libm_samples"] == pl["samples"]
    # SURVEY.md section 8(e) "implement both, report both": the one-process forms (a fresh child behind the ranks), here on one GPU —
    # the form without a collective and the RCCL form with a communicator of one rank
    one = sec["one_process"]
    assert one["gpus"] == 1 and one["direct"]["shards"] == 1 and one["direct"]["projections_per_s"] > 1e5
    assert one["rccl"]["ranks"] == 1 and one["rccl"]["projections_per_s"] > 1e5 and one["rccl"]["same_flags_as_direct"] is True
    assert len(one["rccl"]["kernel_ms_per_gpu"]) == 1 and one["rccl"]["gather_ms_per_gpu"][0] >= 0


@pytest.mark.gpu
def test_profile_evidence_agrees_with_the_live_measurement():
    """the headline at its full size, as the driver runs it (fewer steps, no secondaries): the committed rocprofv3 summary's median
    duration of the dominant kernel — which the line carries as roofline.profile_kernel_ms — must not exceed what this run
    measures per step by more than 3 % (a 4-call profile mean had come out ABOVE the driver's own ms_per_step)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-secondary",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    r = j["roofline"]
    assert len(r["kernel_ms_per_step"]) == 10 and r["kernel_ms_median"] > 0
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert tj.get("kernel_median_ms"), "profiles/traffic_latest.json carries no kernel_median_ms: re-run tools/profile.sh + summarize_profile.py"
    assert r["profile_kernel_ms"] == tj["kernel_median_ms"] and r["profile_kernel"]["dispatches"] >= 10
    assert r["profile_kernel_ms"] <= 1.03 * j["ms_per_step"], (r["profile_kernel_ms"], j["ms_per_step"])
    assert r["profile_kernel_ms"] >= 0.85 * r["kernel_ms_median"]  # and it is the same kernel on the same workload, not something else


FLAT_CONFIG = ("c1_dumbbell_cpu_single_thread_per_s", "c1_dumbbell_gpu_per_s", "c1_bitwise", "c2_batch4096_per_s", "c2_bitwise", "batch32768_per_s",
               "c4_stefan_per_s", "c4_stefan_tight_per_s", "c4_bitwise", "extend_first_pass_edges_per_s", "extend_first_pass_ms",
               "extend_unfinished_edges", "extend_complete_ms", "extend_bitwise", "growtree_5_edges_ms", "single_project_us",
               "single_project_near_manifold_us", "host_buffer_pageable_per_s", "host_buffer_pinned_per_s", "analytic_mode_per_s",
               "c3_calibrated_per_s", "c3_calibrated_bitwise", "c3_calibrated_analytic_per_s", "extend_calibrated_first_pass_edges_per_s", "extend_calibrated_bitwise", "extend_analytic_edges_per_s", "extend_analytic_bitwise", "analytic_bitwise", "analytic_bitwise_samples", "analytic_mode_2m_per_s", "analytic_mode_4096_per_s",
               "proxy_clearance_states_per_s", "extend_bulk_65536_edges_per_s", "extend_bulk_65536_ms", "extend_bulk_bitwise", "extend_cpu_edges_per_s", "extend_cpu_threads", "growtree_5_edges_cpu_single_thread_ms",
               "single_project_cpu_us", "single_project_near_manifold_cpu_us", "single_project_resident_us", "single_project_near_manifold_resident_us",
               "single_is_satisfied_us", "single_is_satisfied_resident_us", "single_function_us", "single_function_resident_us", "single_resident_bitwise",
               "one_edge_check_motion_us", "one_edge_check_motion_resident_us", "one_process_gpus", "one_process_direct_per_s", "one_process_rccl_per_s")
FLAT_CPU = ("det_bit_identical", "det_samples", "libm_samples", "libm_n_gt_1e-6", "libm_max_abs_dq", "libm_iter_diffs_gt1", "libm_ok_mismatches")


def test_flat_keys_are_first_level_scalars():
    """bench.flatten_for_the_driver on a line shaped like the real one (no GPU needed): every figure the driver must keep
    lands as a first-level scalar of config / roofline / cpu_baseline, equal to its nested source."""
    sys.path.insert(0, ROOT)
    import bench

    B = 262144
    par = {"bit_identical": True, "samples": 1024}
    line = {
        "config": {"workload": "w"},
        "roofline": {"bound": "hbm", "fp64": {"frac": 0.016}, "fp64_executed": {"frac": 0.449},
                     "valu_issue": {"frac_of_fp64_issue_ceiling": 0.824, "vector_pipes_busy_in_profiled_launch": 0.905, "source": "profiles/r03f"}},
        "secondary": {
            "batch4096_projections_per_s": 5.1e6, "batch4096": {"parity_vs_det_oracle": par}, "batch32768_projections_per_s": 11.4e6,
            "single_project_c_abi": {"uniform_sample_median_us": 122.0, "near_manifold_median_us": 54.7, "uniform_sample_cpu_median_us": 1100.0,
                                     "near_manifold_cpu_median_us": 350.0, "uniform_sample_resident_median_us": 110.0,
                                     "near_manifold_resident_median_us": 45.0, "is_satisfied_median_us": 21.0, "is_satisfied_resident_median_us": 8.0,
                                     "function_median_us": 21.0, "function_resident_median_us": 8.0, "resident_bit_identical_to_launched": True,
                                     "one_edge_check_motion_median_us": 150.0, "one_edge_check_motion_resident_median_us": 130.0},
            "analytic_mode_projections_per_s": 1.6e8, "analytic_mode": {"projections_per_s": 1.6e8, "parity_vs_det_oracle": {"bit_identical": True, "samples": 4096}},
            "analytic_mode_batch2097152": {"projections_per_s": 2.9e8}, "analytic_mode_batch4096": {"projections_per_s": 8.0e6},
            "stefan_batch%d_tol_1e-3_5e-3" % B: {"projections_per_s": 9.9e6, "parity_vs_det_oracle": par},
            "stefan_batch%d_tol_5e-4_2.5e-3" % B: {"projections_per_s": 9.0e6, "parity_vs_det_oracle": par},
            "discrete_geodesic": {"edges_per_s": 10.7e6, "ms": 1.52, "overflowed_edges": 29, "complete_ms": 42.8, "growtree_5_edges_ms": 0.41,
                                  "cpu": {"edges_per_s": 9000.0, "threads": 16, "growtree_5_edges_single_thread_ms": 7.5},
                                  "bulk": {"edges_per_s": 15.9e6, "ms": 4.1, "parity_vs_det_oracle": {"bit_identical": True}},
                                  "parity_vs_det_oracle": {"bit_identical": True, "continued_edges": {"bit_identical": True}}},
            "proxy_clearance": {"states_per_s": 6.6e8},
            "extend_analytic": {"edges_per_s": 2.0e7, "ms": 0.8, "parity_vs_det_oracle": {"edges": 256, "bit_identical": True}},
            "c3_calibrated": {"projections_per_s": 14.0e6, "analytic_projections_per_s": 1.5e8, "parity_vs_det_oracle": {"bit_identical": True, "samples": 2048},
                              "extend_first_pass_edges_per_s": 9.0e6, "extend_parity_vs_det_oracle": {"bit_identical": True, "edges": 256}},
            "host_buffer": {"pageable": {"projections_per_s": 14.5e6}, "pinned": {"projections_per_s": 13.2e6}},
            "c1_dumbbell": {"cpu_single_thread_projections_per_s": 560.0, "gpu_projections_per_s": 1.1e6, "parity_vs_det_oracle": par},
            "one_process": {"gpus": 1, "direct": {"projections_per_s": 13.9e6}, "rccl": {"projections_per_s": 13.1e6}},
        },
        "cpu_baseline": {"value": 14800.0, "parity_gpu_vs_det_oracle": {"bit_identical": True, "samples": 2048},
                         "parity_gpu_vs_libm_oracle": {"samples": 131072, "n_gt_1e-6": 26221, "max_abs_dq": 0.046, "iteration_diffs_gt1": 8620,
                                                       "ok_mismatches": 3}},
    }
    bench.flatten_for_the_driver(line, B)
    for k in FLAT_CONFIG:
        assert isinstance(line["config"][k], (int, float, bool)), k
    assert line["config"]["c4_stefan_tight_per_s"] == 9.0e6 and line["config"]["extend_complete_ms"] == 42.8 and line["config"]["c4_bitwise"] is True
    rf = line["roofline"]
    assert (rf["fp64_algorithmic_frac"], rf["fp64_executed_frac"], rf["valu_issue_frac"], rf["pipes_busy_frac"]) == (0.016, 0.449, 0.824, 0.905)
    for k in FLAT_CPU:
        assert isinstance(line["cpu_baseline"][k], (int, float, bool)), k
    assert line["cpu_baseline"]["libm_n_gt_1e-6"] == 26221 and line["cpu_baseline"]["det_bit_identical"] is True
    # a line without secondaries / with a failed CPU leg (N > 1, --no-secondary) must not break
    bare = {"config": {}, "roofline": {"fp64": {"frac": 0.01}}, "cpu_baseline": {"error": "x"}}
    bench.flatten_for_the_driver(bare, B)
    assert bare["roofline"]["fp64_executed_frac"] is None and "det_bit_identical" not in bare["cpu_baseline"]


def _run_bench(*argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT, env=e)


def test_bench_starts_its_own_ranks(ccmp_built):
    """`python bench.py --gpus N` as a bare command starts N ranks (fresh child processes, rendezvous on 127.0.0.1) and
    relays rank 0's line; here the ranks only meet on the CPU (--launch-check): no GPU in this container."""
    out = _run_bench("--gpus", "3", "--launch-check")
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert j == {"launch_check": True, "n_gpus": 3, "rank_sum": 3.0, "self_launched": True}
    # a launcher that already set WORLD_SIZE must agree with --gpus: no silent single-rank line
    out = _run_bench("--gpus", "8", "--launch-check", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
    out = _run_bench("--gpus", "1", "--launch-check")
    assert out.returncode == 0 and json.loads(out.stdout.splitlines()[-1])["n_gpus"] == 1


@pytest.mark.gpu
def test_bench_two_ranks_from_the_bare_command():
    """the N > 1 step for real — projection, capped compaction into the gather block, one all-gather per step — as two
    self-launched ranks sharing this box's one GPU over gloo (RCCL refuses two ranks on one device)"""
    out = _run_bench("--gpus", "2", "--backend", "gloo", "--force-device", "0", "--steps", "1", "--warmup", "1", "--batch", "32768")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 65536 and j["rccl_ranks"] == 0  # gloo rehearsal, not RCCL
    g = j["stats"]["gather"]
    assert g["overflow"] is False and 0.15 * 65536 < g["valid_states_all_ranks"] < 0.3 * 65536
    assert g["capacity_rows_per_rank"] == 16384 and j["value"] > 5e4  # gloo moves the blocks through the host: a rehearsal, not a rate
    # what a first multi-GPU run is diagnosed by: per-rank kernel and step times, the collective's tail, bytes moved, and
    # the §8e identity — gathered valid states of a 4096-sample probe == those of the same probe on one GPU, bit for bit
    assert len(g["kernel_ms_per_rank"]) == 2 and g["kernel_ms_min"] <= g["kernel_ms_max"] and len(g["step_ms_per_rank"]) == 2
    assert len(g["last_kernel_to_gather_done_ms_per_rank"]) == 2 and g["bytes_sent_per_rank"] == 16385 * 112
    assert g["bytes_received_per_rank"] == 2 * g["bytes_sent_per_rank"]
    pr = g["probe"]
    assert pr["samples"] == 4096 and pr["bit_identical_to_one_gpu"] is True and sum(pr["valid_per_rank"]) == pr["valid_states"] > 500
    # the other multi-GPU form of SURVEY.md section 8(e), in the same line: ONE process driving the GPUs through the C ABI's sharded
    # entry points — a fresh child started when the ranks are done; here two contexts on the one card (so: the form without the
    # collective; RCCL wants a device per rank and says so)
    one = j["secondary"]["one_process"]
    assert one["gpus"] == 2 and one["devices"] == [0, 0] and one["samples_per_call"] == 65536
    d = one["direct"]
    assert d["shards"] == 2 and d["collective"] == "none" and d["projections_per_s"] > 1e5 and 0.15 < d["ok_fraction"] < 0.3
    assert len(d["shard_launch_ms"]) == 2 and len(d["shard_start_ms_on_gpu"]) == 2
    assert "skipped" in one["rccl"]
    assert j["config"]["one_process_gpus"] == 2 and j["config"]["one_process_direct_per_s"] == d["projections_per_s"]
    assert j["config"]["one_process_rccl_per_s"] is None
