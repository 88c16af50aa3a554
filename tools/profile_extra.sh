#!/bin/bash
# extra PMC passes: instruction cache, scalar data cache, FP64 op mix
set -e
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary ${BENCH_ARGS}"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH --output-format csv -d $OUT/pmc_ic -- python3 $ARGS > $OUT/pmc_ic.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_dc -- python3 $ARGS > $OUT/pmc_dc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/pmc_mix -- python3 $ARGS > $OUT/pmc_mix.log 2>&1 || true
python3 - <<PY
import csv, glob, collections
for d in ("pmc_ic","pmc_dc","pmc_mix"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = "group" if "project_fd_kernel" in r["Kernel_Name"] else ("wave" if "project_fd_wave" in r["Kernel_Name"] else None)
            if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            print(d, k, {c: "%.4g" % (sum(v)/len(v)) for c, v in agg[k].items()})
PY
