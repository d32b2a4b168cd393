#!/bin/bash
# One GPU-box session: tests, headline bench, kernel trace + PMC passes over the same command.  usage: tools/gpu_round.sh <tag> [skiptests]
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$2" != "skiptests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=10 > $O/${TAG}_tests.log 2>&1
  echo "tests exit $?" | tee -a $O/${TAG}_tests.log
  tail -5 $O/${TAG}_tests.log
fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { echo bench failed; tail -20 $O/${TAG}_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/${TAG}_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["launch_ms"], "train", d.get("train_step",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"))
print(d["kernels_ms_per_step"])
PY
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o t -- python3 $R/bench.py --pmc-run --steps 10 --warmup 2 > $O/${TAG}_trace.log 2>&1 || { echo trace failed; tail -5 $O/${TAG}_trace.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_f -o f -- python3 $R/bench.py --pmc-run --steps 3 --warmup 1 > $O/${TAG}_pmc_f.log 2>&1 || { echo pmc f failed; tail -5 $O/${TAG}_pmc_f.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_w -o w -- python3 $R/bench.py --pmc-run --steps 3 --warmup 1 > $O/${TAG}_pmc_w.log 2>&1 || { echo pmc w failed; tail -5 $O/${TAG}_pmc_w.log; exit 1; }
cd $R
python3 tools/pmc_table.py --forwards 4 --commit "$(cat $R/.commit 2>/dev/null || echo unknown)" --out $O/${TAG}_pmc_traffic.json $O/${TAG}_pmc_f $O/${TAG}_pmc_w
find $O/${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_kernel_stats.csv
# keep the merge-back small: the raw counter CSVs are large
rm -rf $O/${TAG}_pmc_f $O/${TAG}_pmc_w $O/${TAG}_trace
ls -la $O | tail -12
