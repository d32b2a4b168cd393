#!/bin/bash
# GPU box: where the waves of a bench config's kernels spend their cycles (SQ wait / issue counters, kernels serialised).
#   tools/gpu_waits.sh <tag> [config]      -> gpurun_out/<tag>_waits.txt
set -o pipefail
TAG=${1:-x}
CFG=${2:-dptn_av}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/${TAG}_wpmc -o m -- python3 $R/bench.py --config $CFG --pmc-run --steps 1 --warmup 1 > $O/${TAG}_wpmc.log 2>&1 || { echo pmc failed; tail -5 $O/${TAG}_wpmc.log; exit 1; }
cd $R
python3 tools/pmc_summary.py $O/${TAG}_wpmc > $O/${TAG}_waits_raw.txt
python3 tools/waits_table.py $O/${TAG}_waits_raw.txt > $O/${TAG}_waits.txt
rm -rf $O/${TAG}_wpmc
