#!/bin/bash
# GPU box: bench lines of the non-headline configurations + MFMA-utilisation counters of the headline run.  usage: tools/gpu_extra.sh <tag>
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for cfg in dptn_audio dprnn_av; do
  timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 > $O/${TAG}_bench_$cfg.json 2> $O/${TAG}_bench_$cfg.err || { echo "bench $cfg failed"; tail -5 $O/${TAG}_bench_$cfg.err; }
  python - <<PY
import json
d=json.load(open("$O/${TAG}_bench_$cfg.json"))
s=d.get("split_bf16_experiment") or {}
print("$cfg", d["value"], "ms", d["ms_per_step"], "whole frac", d["roofline"]["whole_path_frac"], "split", s.get("value"), s.get("agreement_db_vs_f32_run"))
PY
done
timeout -k 10 600 python bench.py --config dptn_av_train --steps 10 --warmup 3 > $O/${TAG}_bench_train.json 2> $O/${TAG}_bench_train.err || { echo "bench train failed"; tail -5 $O/${TAG}_bench_train.err; }
python -c "
import json; d=json.load(open('$O/${TAG}_bench_train.json')); print('train', d['value'], d['ms_per_step'], d['roofline']['frac'], d['last_step'])"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_m -o m -- python3 $R/bench.py --pmc-run --steps 3 --warmup 1 > $O/${TAG}_pmc_m.log 2>&1 || { echo pmc m failed; tail -5 $O/${TAG}_pmc_m.log; exit 1; }
cd $R
python3 tools/pmc_summary.py $O/${TAG}_pmc_m > $O/${TAG}_pmc_mfma_summary.txt
python3 tools/mfma_util.py $O/${TAG}_pmc_mfma_summary.txt | tee $O/${TAG}_mfma_utilisation.txt
rm -rf $O/${TAG}_pmc_m
