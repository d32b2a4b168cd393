#!/bin/bash
# headline bench (isolated rows) + kernel trace of config 2 (attn_block64) + MFMA-busy counters of both attention blocks
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/r03e_bench.json 2> $O/r03e_bench.err || { echo bench failed; tail -20 $O/r03e_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/r03e_bench.json"))
print("value", d["value"], d["ms_per_step"]); print(d["roofline"]["isolated"]); print({k:(v.get("value"),v.get("ms_per_step")) for k,v in d["other_configs"].items()}, d["latency_b1"].get("mean_ms"), d["train_step"].get("value"))
PY
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03e_trace2 -o t -- python3 $R/bench.py --config dptn_audio --pmc-run --steps 10 --warmup 2 > $O/r03e_trace2.log 2>&1 || { echo trace failed; tail -5 $O/r03e_trace2.log; exit 1; }
find $O/r03e_trace2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03e_audio_kernel_stats.csv
rm -rf $O/r03e_trace2
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r03e_mfma -o m -- python3 $R/bench.py --config dptn_audio --pmc-run --steps 2 --warmup 1 > $O/r03e_mfma.log 2>&1 || { echo pmc failed; tail -5 $O/r03e_mfma.log; exit 1; }
cd $R; python3 tools/pmc_summary.py $O/r03e_mfma > $O/r03e_mfma_summary.txt; python3 tools/mfma_util.py $O/r03e_mfma_summary.txt > $O/r03e_audio_mfma_utilisation.txt 2>&1; rm -rf $O/r03e_mfma
head -12 $O/r03e_audio_mfma_utilisation.txt
head -6 $O/r03e_audio_kernel_stats.csv | cut -c1-160
