#!/bin/bash
# rocprofv3 passes over one workload; run on the GPU box from the repo root:
#   bash tools/profile.sh <tag>                      bench.py at its default configuration (C3)
#   bash tools/profile.sh <tag> <workload> [reps]    tools/measure.py run <workload>: flat4096 | flat1 | geodesic | geodesic65536 | analytic[<B>] | stefan | stefan_tight | calibrated | clearance (reps: profiled calls, default 12, behind 3 warm-ups)
# -> gpurun_out/prof_<tag>/{trace,pmc_*}/...; condense with: python tools/summarize_profile.py <tag> <kernel,...> <units per launch>
# The program after `--` is python3 itself (no env / bash -c / launcher hop: the profiler's library initialises the GPU
# before the program starts).  Counters are collected in their own passes, never together with a trace domain other
# than --kernel-trace.
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
# >= 10 profiled calls behind >= 3 warm-up calls (a 4-call mean with a cold outlier was thin evidence); the summary
# reports median and quartiles over the profiled calls only
REPS=${3:-12}
if [ -n "$2" ]; then ARGS="$ROOT/tools/measure.py run $2 $REPS 3"; else ARGS="$ROOT/bench.py --steps $REPS --warmup 3 --no-cpu-baseline --no-secondary ${BENCH_ARGS}"; fi
cd $ROOT && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/pmc_mix -- python3 $ARGS > $OUT/pmc_mix.log 2>&1 || true
find $OUT -name "*_kernel_stats.csv" | head -3
