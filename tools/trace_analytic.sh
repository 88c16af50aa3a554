set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in "default:" "alone:analytic_handover=0" "h16:analytic_handover=16"; do
  tag=${cfg%%:*}; opts=${cfg#*:}
  CCMP_OPTS=$opts rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$tag -- python3 tools/measure.py run analytic 4 2 > gpurun_out/tr_$tag.log 2>&1
  f=$(find gpurun_out/tr_$tag -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'project_' in r['Kernel_Name']]
t0=None
for r in rows[-6:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print(sys.argv[2], r['Kernel_Name'][:40], 'start %.3f ms dur %.3f ms'%((s-t0)/1e6,(e-s)/1e6))
PY
done
