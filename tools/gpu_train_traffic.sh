#!/bin/bash
# GPU box: per-kernel HBM traffic of the TRAINING step (config 4), rocprofv3 PMC, separate FETCH_SIZE / WRITE_SIZE passes.
#   tools/gpu_train_traffic.sh <tag>     -> gpurun_out/<tag>_train_pmc_traffic.json  (copy to profiles/r02_train_pmc_traffic.json)
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
CMD="python3 bench.py --config dptn_av_train --pmc-run --steps 1 --warmup 1"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_tpf -o f -- python3 $R/bench.py --config dptn_av_train --pmc-run --steps 1 --warmup 1 > $O/${TAG}_tpf.log 2>&1 || { echo pmc f failed; tail -5 $O/${TAG}_tpf.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_tpw -o w -- python3 $R/bench.py --config dptn_av_train --pmc-run --steps 1 --warmup 1 > $O/${TAG}_tpw.log 2>&1 || { echo pmc w failed; tail -5 $O/${TAG}_tpw.log; exit 1; }
cd $R
python3 tools/pmc_table.py --forwards 2 --config dptn_av_train --command "$CMD" --commit "$(cat $R/.commit 2>/dev/null || echo unknown)" --out $O/${TAG}_train_pmc_traffic.json $O/${TAG}_tpf $O/${TAG}_tpw
rm -rf $O/${TAG}_tpf $O/${TAG}_tpw
