set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r5_gpu_suite5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_gpu_suite5.log
tail -5 gpurun_out/r5_gpu_suite5.log
timeout -k 10 600 python bench.py > gpurun_out/r5_bench3.json 2> gpurun_out/r5_bench3.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r5_bench3.json"))
print("value %.4g  ms_per_step %.3f  kernel_ms_median %.3f profile_kernel_ms %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_median"], j["roofline"]["profile_kernel_ms"]))
c = j["config"]
for k in ("c2_batch4096_per_s", "batch32768_per_s", "c4_stefan_per_s", "c4_stefan_tight_per_s", "extend_first_pass_edges_per_s", "extend_first_pass_ms", "extend_bulk_65536_edges_per_s", "extend_complete_ms",
          "growtree_5_edges_ms", "single_project_us", "single_project_near_manifold_us", "single_project_resident_us", "single_project_near_manifold_resident_us",
          "single_is_satisfied_us", "single_is_satisfied_resident_us", "single_function_us", "single_function_resident_us", "single_resident_bitwise", "one_edge_check_motion_us", "one_edge_check_motion_resident_us",
          "one_process_direct_per_s", "one_process_rccl_per_s", "host_buffer_pageable_per_s", "host_buffer_pinned_per_s", "analytic_mode_per_s"):
    print("  %-44s %s" % (k, c.get(k)))
PY
