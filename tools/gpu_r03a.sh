#!/bin/bash
# round-3 first GPU session: the new tests (reference gradient fixtures, RCCL world-1 group, frozen parameter) and the new bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_backward.py tests/test_gpu_parallel.py tests/test_gpu_train_tail.py -x -q -m gpu -rA \
   -k "reference_gradients or rccl or frozen" > $O/r03_a_tests.log 2>&1
echo "tests exit $?"
grep -n "dB (the\|passed\|failed\|Error" $O/r03_a_tests.log | tail -20
timeout -k 10 600 python bench.py > $O/r03_a_bench.json 2> $O/r03_a_bench.err
echo "bench exit $?"; tail -5 $O/r03_a_bench.err; head -c 2500 $O/r03_a_bench.json
