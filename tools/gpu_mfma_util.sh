#!/bin/bash
# GPU box: MFMA-busy fraction per kernel (kernels serialised by the counter pass = alone on the chip) for the forward of a configuration.
#   tools/gpu_mfma_util.sh <tag> <config>   -> gpurun_out/<tag>_<config>_mfma_utilisation.txt
set -o pipefail
TAG=${1:-x}; CFG=${2:-dptn_av}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_m -o m -- python3 $R/bench.py --config $CFG --pmc-run --steps 2 --warmup 1 > $O/${TAG}_pmc_m.log 2>&1 || { echo pmc failed; tail -5 $O/${TAG}_pmc_m.log; exit 1; }
cd $R
python3 tools/pmc_summary.py $O/${TAG}_pmc_m > $O/${TAG}_${CFG}_pmc_summary.txt
python3 tools/mfma_util.py $O/${TAG}_${CFG}_pmc_summary.txt | tee $O/${TAG}_${CFG}_mfma_utilisation.txt | head -14
rm -rf $O/${TAG}_pmc_m
