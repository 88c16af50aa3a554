set -o pipefail
cd $GRAFT_REPO_ROOT
export R5_LIBB=closed_chain_motion_planner_amd/lib/libccmp_r4.so
export AB_LIB=closed_chain_motion_planner_amd/lib/libccmp_r4.so
echo "# round 5 (A, lib/libccmp.so) against the round-4 library (B, built from commit 8c3e306), interleaved on one device: project_batch, B/A time ratio (> 1: round 5 faster)" > gpurun_out/r5_vs_r4.log
for obj in Wine_Bottle stefan; do for B in 2560 4096 8192 12288 16384 24576 32768 65536 122880 262144; do timeout -k 10 120 python tools/ab.py run $obj $B 2>/dev/null | grep -E "ratio" | sed "s/^/$obj B=$B  /" >> gpurun_out/r5_vs_r4.log; done; done
cat gpurun_out/r5_vs_r4.log
export R5_SIZES=4096,8192,16384,32768,65536,131072
export R5_CFGS="now:;B/r4:"
export R5_ROUNDS=3
timeout -k 10 600 python tools/exp_r5.py bulk_ab > gpurun_out/r5_vs_r4_extend.log 2>&1; echo "rc=$?"
cat gpurun_out/r5_vs_r4_extend.log | grep -E "^(Wine|stefan)"
timeout -k 10 400 python -m pytest tests/test_gpu_callers.py tests/test_gpu_parity.py tests/test_gpu_usage_modes.py -m gpu -x -q > gpurun_out/r5_t6.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_t6.log
