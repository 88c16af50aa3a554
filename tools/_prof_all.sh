set -e
cd $GRAFT_REPO_ROOT
for w in analytic:r06b analytic2097152:r06c; do
  wl=${w%%:*}; tag=${w#*:}
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 420 bash tools/profile.sh $tag $wl 12 > gpurun_out/prof_$tag.out 2>&1 || echo "profile $tag failed"
  echo "done $tag $(date +%T)"
done
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r6_gpu_suite3.log 2>&1 || true
tail -3 gpurun_out/r6_gpu_suite3.log
