set -o pipefail
cd $GRAFT_REPO_ROOT
export R5_SIZES=8192,10240,12288,13312,14336,15360
export R5_CFGS="flat:geodesic_group=0;bulk:geodesic_group_min=0;bulk_c32:geodesic_group_min=0,geodesic_group_low_cut=32;bulk_c48:geodesic_group_min=0,geodesic_group_low_cut=48"
export R5_ROUNDS=3
timeout -k 10 500 python tools/exp_r5.py bulk_ab > gpurun_out/r5_ab6.log 2>&1; echo "ab rc=$?" >> gpurun_out/r5_ab6.log
cat gpurun_out/r5_ab6.log
