#!/bin/bash
# GPU box: instruction mix and pipe-busy counters of the training step's kernels (serialised by the counter pass).
#   tools/gpu_train_valu.sh <tag>   -> gpurun_out/<tag>_train_valu_raw.txt
set -o pipefail
TAG=${1:-x}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/${TAG}_vpmc -o m -- python3 $R/bench.py --config dptn_av_train --pmc-run --steps 1 --warmup 1 > $O/${TAG}_vpmc.log 2>&1 || { echo pmc failed; tail -5 $O/${TAG}_vpmc.log; exit 1; }
cd $R
python3 tools/pmc_summary.py $O/${TAG}_vpmc > $O/${TAG}_train_valu_raw.txt
rm -rf $O/${TAG}_vpmc
grep -A9 "attention_bwd_kernel<32, 5, 0>\|attention_bwd_kernel<32, 5, 1>\|attention_kernel<32, 5>\|gemm_ws_kernel<512" $O/${TAG}_train_valu_raw.txt | head -60
