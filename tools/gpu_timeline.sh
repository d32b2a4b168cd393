#!/bin/bash
# GPU box: kernel timelines (rocprofv3 --kernel-trace) of the forward configurations and of the training step, summarised by
# tools/timeline_summary.py (kernels / recurrence launches in flight over the last step) -> gpurun_out/<tag>_timeline_summaries.txt
set -o pipefail
TAG=${1:-r05}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
: > $O/${TAG}_timeline_summaries.txt
for CFG in dptn_av dptn_audio dprnn_av dptn_av_train; do
  ST=$([ $CFG = dprnn_av ] && echo "--steps 3 --warmup 1" || echo "--steps 4 --warmup 2")
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_tl -o t -- python3 $R/bench.py --config $CFG --pmc-run $ST > $O/${TAG}_tl.log 2>&1 || { echo timeline $CFG failed; tail -5 $O/${TAG}_tl.log; exit 1; }
  F=$(find $O/${TAG}_tl -name "*kernel_trace.csv" | head -1)
  MODE=$([ $CFG = dptn_av_train ] && echo train || { [ $CFG = dprnn_av ] && echo "forward 2" || echo "forward 3"; })
  { echo "== $CFG (bench.py --config $CFG --pmc-run $ST), commit $(cat $R/.commit 2>/dev/null) =="; python3 $R/tools/timeline_summary.py $F $MODE; echo; } >> $O/${TAG}_timeline_summaries.txt
  rm -rf $O/${TAG}_tl
done
cat $O/${TAG}_timeline_summaries.txt
