#!/bin/bash
# host side of the eager forward: HIP API trace statistics
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --hip-trace --stats --output-format csv -d $O/hiptrace_hip -o h -- python3 $R/bench.py --pmc-run --steps 10 --warmup 2 > $O/hiptrace_hip.log 2>&1 || { echo trace failed; tail -5 $O/hiptrace_hip.log; exit 1; }
find $O/hiptrace_hip -name "*hip_api_stats.csv" | head -1 | xargs -I{} cp {} $O/hiptrace_hip_api_stats.csv
rm -rf $O/hiptrace_hip
head -25 $O/hiptrace_hip_api_stats.csv
