set -o pipefail
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r05a gpurun_out/prof_r05i gpurun_out/prof_r05j gpurun_out/prof_r05f
timeout -k 10 400 bash tools/profile.sh r05a > gpurun_out/r5_prof_a.log 2>&1; echo "prof a rc=$?"
timeout -k 10 400 bash tools/profile.sh r05i geodesic 12 > gpurun_out/r5_prof_i.log 2>&1; echo "prof i rc=$?"
timeout -k 10 400 bash tools/profile.sh r05j geodesic65536 12 > gpurun_out/r5_prof_j.log 2>&1; echo "prof j rc=$?"
timeout -k 10 400 bash tools/profile.sh r05f flat4096 12 > gpurun_out/r5_prof_f.log 2>&1; echo "prof f rc=$?"
export TMPDIR=/tmp
for E in 16384 65536; do rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_tl_final_$E -- python3 tools/exp_r5.py timeline $E > gpurun_out/r5_tl_final_$E.log 2>&1; done
echo done
