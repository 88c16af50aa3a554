set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/policy_check.py --log gpurun_out/r05_policy_check.log --reps 6 --rounds 3 > /dev/null 2>&1; echo "policy rc=$?"
grep -v "^    " gpurun_out/r05_policy_check.log | grep -E "<--|RESULT" | cut -c1-220
timeout -k 10 300 python tools/exp_r5.py resident_ab > gpurun_out/r5_resident_ab.log 2>&1; echo "resident_ab rc=$?"
grep -E "^(A|B) " gpurun_out/r5_resident_ab.log
timeout -k 10 400 bash tools/profile.sh r05a > gpurun_out/r5_prof_a.log 2>&1; echo "prof a rc=$?"
timeout -k 10 400 bash tools/profile.sh r05i geodesic 12 > gpurun_out/r5_prof_i.log 2>&1; echo "prof i rc=$?"
timeout -k 10 400 bash tools/profile.sh r05j geodesic65536 12 > gpurun_out/r5_prof_j.log 2>&1; echo "prof j rc=$?"
timeout -k 10 400 bash tools/profile.sh r05f flat4096 12 > gpurun_out/r5_prof_f.log 2>&1; echo "prof f rc=$?"
du -sh gpurun_out/prof_r05*
