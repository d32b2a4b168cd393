#!/bin/bash
# N = 64 fused attention block: parity tests, then config 2 with / without it
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/r03c_tests.log 2>&1
echo "tests exit $?"; tail -15 $O/r03c_tests.log
timeout -k 10 300 python bench.py --config dptn_audio --steps 20 --warmup 5 > $O/r03c_bench_audio.json 2> $O/r03c_bench_audio.err
echo "bench exit $?"; tail -3 $O/r03c_bench_audio.err
python - <<PY
import json
d=json.load(open("$O/r03c_bench_audio.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["roofline"]["whole_path_frac"]); print(d["kernels_ms_per_step"])
PY
