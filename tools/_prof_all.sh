set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "calibrated_arms_runs_unfused or stock_structure" 2>&1 | tail -2
for w in analytic:r06b analytic2097152:r06c stefan:r06d stefan_tight:r06e calibrated:r06g; do
  wl=${w%%:*}; tag=${w#*:}
  timeout -k 10 420 bash tools/profile.sh $tag $wl 12 > gpurun_out/prof_$tag.out 2>&1 || echo "profile $tag failed"
  echo "done $tag $(date +%T)"
done
