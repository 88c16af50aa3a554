set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1700 python tests/parity_campaign.py > gpurun_out/r5_campaign.log 2>&1; echo "campaign rc=$?"
tail -15 gpurun_out/r5_campaign.log
cp profiles/parity_campaign.json gpurun_out/r5_parity_campaign.json
