#!/bin/bash
# GPU box: kernel timeline of two training steps (start / end of every launch) -> gpurun_out/<tag>_train_timeline.csv
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_tl -o t -- python3 $R/bench.py --config ${2:-dptn_av_train} --pmc-run --steps 2 --warmup 2 > $O/${TAG}_tl.log 2>&1 || { echo trace failed; tail -5 $O/${TAG}_tl.log; exit 1; }
find $O/${TAG}_tl -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_train_timeline.csv
rm -rf $O/${TAG}_tl
wc -l $O/${TAG}_train_timeline.csv
