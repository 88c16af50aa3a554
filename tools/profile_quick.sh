#!/bin/bash
# kernel-trace stats of an arbitrary python script:  bash tools/profile_quick.sh <tag> <script.py> [args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/$@ > $OUT/trace.log 2>&1
cat $OUT/trace/*/*_kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://; s/(ccmp_consts.*//' | head -14
