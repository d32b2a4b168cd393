#!/bin/bash
# GPU box: per-kernel MFMA utilisation (kernels serialised by the counter pass) + kernel trace of the training step.
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ttrace -o t -- python3 $R/bench.py --config dptn_av_train --pmc-run --steps 5 --warmup 2 > $O/${TAG}_ttrace.log 2>&1 || { echo trace failed; tail -5 $O/${TAG}_ttrace.log; exit 1; }
find $O/${TAG}_ttrace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_train_kernel_stats.csv
rm -rf $O/${TAG}_ttrace
timeout -k 10 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_tpmc -o m -- python3 $R/bench.py --config dptn_av_train --pmc-run --steps 1 --warmup 1 > $O/${TAG}_tpmc.log 2>&1 || { echo pmc failed; tail -5 $O/${TAG}_tpmc.log; exit 1; }
cd $R
python3 tools/pmc_summary.py $O/${TAG}_tpmc > $O/${TAG}_train_pmc_summary.txt
python3 tools/mfma_util.py $O/${TAG}_train_pmc_summary.txt > $O/${TAG}_train_mfma_utilisation.txt
rm -rf $O/${TAG}_tpmc
head -40 $O/${TAG}_train_mfma_utilisation.txt
