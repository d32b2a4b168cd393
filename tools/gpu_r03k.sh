#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hipgraph or use_graphs or sub_batches or split_policy" > $O/r03k_tests.log 2>&1
echo "tests exit $?"; tail -6 $O/r03k_tests.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/r03k_bench.json 2> $O/r03k_bench.err; echo "bench exit $?"; tail -4 $O/r03k_bench.err
python - <<PY
import json
d=json.load(open("$O/r03k_bench.json"))
print("value", d["value"], d["ms_per_step"], d["config"]["launch"], "eager", d["eager_ms_per_step"], d["roofline"]["whole_path_frac"])
print({k:(v.get("value"),v.get("ms_per_step"),v.get("launch")) for k,v in d["other_configs"].items()}, d["latency_b1"], d["train_step"].get("value"))
PY
