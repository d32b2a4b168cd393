#!/bin/bash
# LayerNorm backward from the tape: gradient tests, then the training step with / without it
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_tail.py -x -q -m gpu > $O/r03f_tests.log 2>&1
echo "tests exit $?"; tail -8 $O/r03f_tests.log
timeout -k 10 400 python bench.py --config dptn_av_train --steps 10 --warmup 3 > $O/r03f_train.json 2> $O/r03f_train.err; echo "train bench exit $?"
python - <<PY
import json
d=json.load(open("$O/r03f_train.json")); print("train", d["value"], d["ms_per_step"], d["roofline"]["frac"]); print(d["kernels_ms_per_step"])
PY
