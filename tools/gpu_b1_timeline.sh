#!/bin/bash
# GPU box: kernel timeline of bs = 1 forwards -> gpurun_out/<tag>_b1_timeline.txt
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for CFG in dptn_av dptn_audio; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_b1 -o t -- python3 $R/tools/b1_forward.py $CFG 6 > $O/${TAG}_b1.log 2>&1 || { echo trace failed; tail -5 $O/${TAG}_b1.log; exit 1; }
  F=$(find $O/${TAG}_b1 -name "*kernel_trace.csv" | head -1)
  { echo "== $CFG, bs = 1, T = 32000 =="; python3 $R/tools/b1_timeline_summary.py $F; } >> $O/${TAG}_b1_timeline.txt
  rm -rf $O/${TAG}_b1
done
cat $O/${TAG}_b1_timeline.txt
