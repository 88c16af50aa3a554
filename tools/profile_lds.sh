#!/bin/bash
# LDS-pipe counters of a tools/measure.py workload (run on the GPU box from the repo root):  bash tools/profile_lds.sh <tag> <workload> [reps]
# -> gpurun_out/prof_<tag>/pmc_lds{1,2}; the program after `--` is python3 itself; counters only with --kernel-trace.
TAG=$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
ARGS="$ROOT/tools/measure.py run $2 ${3:-3}"
cd $ROOT && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds1 -- python3 $ARGS > $OUT/pmc_lds1.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds2 -- python3 $ARGS > $OUT/pmc_lds2.log 2>&1 || true
python3 - <<PY
import csv, glob, collections
for d in ("pmc_lds1", "pmc_lds2"):
    for f in sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % d))[-1:]:
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in per.items():
            print(d, k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()}, "launches", max(len(v) for v in cs.values()))
PY
