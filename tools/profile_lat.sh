#!/bin/bash
set -e
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary ${BENCH_ARGS}"
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_lat -- python3 $ARGS > $OUT/pmc_lat.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_lat2 -- python3 $ARGS > $OUT/pmc_lat2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_lat","pmc_lat2"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = "group" if "project_fd_kernel" in r["Kernel_Name"] else ("wave" if "project_fd_wave" in r["Kernel_Name"] else None)
            if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            print(d, k, {c: "%.4g" % (sum(v)/len(v)) for c, v in agg[k].items()})
PY
