set -o pipefail
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_two_ctx -- python3 tools/exp_r5.py two_contexts 24064 > gpurun_out/r5_two_ctx.log 2>&1
cat gpurun_out/r5_two_ctx.log | tail -6
python3 tools/exp_r5.py two_contexts_report gpurun_out/r5_two_ctx > gpurun_out/r5_two_ctx_report.log 2>&1
tail -40 gpurun_out/r5_two_ctx_report.log
