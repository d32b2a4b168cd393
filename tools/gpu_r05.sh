#!/bin/bash
# GPU box, round 5: every profile the bench line and DESIGN quote, taken from ONE tree (write .commit first:
#   git rev-parse --short HEAD > .commit).   usage: tools/gpu_r05.sh <tag> [parts]     parts: any of  bench trace pmc mfma train
# Outputs under gpurun_out/<tag>_*; copy what is to be judged into profiles/r05_*.
set -o pipefail
TAG=${1:-r05}; PARTS=${2:-"bench trace pmc mfma train"}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
COMMIT="$(cat $R/.commit 2>/dev/null || echo unknown)"
has() { [[ " $PARTS " == *" $1 "* ]]; }
cd $R
cd /tmp
if has trace; then
  for MODE in "" "--serialize"; do
    SUF=$([ -z "$MODE" ] && echo as_run || echo serialised)
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o t -- python3 $R/bench.py --pmc-run $MODE --steps 10 --warmup 2 > $O/${TAG}_trace.log 2>&1 || { echo trace failed; tail -5 $O/${TAG}_trace.log; exit 1; }
    find $O/${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_kernel_stats_${SUF}.csv
    rm -rf $O/${TAG}_trace
  done
  for CFG in dptn_audio dprnn_av; do
    ST=$([ $CFG = dprnn_av ] && echo "--steps 2 --warmup 1" || echo "--steps 10 --warmup 2")
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o t -- python3 $R/bench.py --config $CFG --pmc-run --serialize $ST > $O/${TAG}_trace.log 2>&1 || { echo trace $CFG failed; tail -5 $O/${TAG}_trace.log; exit 1; }
    find $O/${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_kernel_stats_serialised_${CFG}.csv
    rm -rf $O/${TAG}_trace
  done
fi
if has pmc; then
  for CFG in dptn_av dptn_audio dprnn_av; do
    if [ $CFG = dprnn_av ]; then ST="--steps 1 --warmup 1"; NF=2; BATCH=32; SAMPLES=128000; else ST="--steps 3 --warmup 1"; NF=4; BATCH=16; SAMPLES=32000; fi
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 500 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_$C -o c -- python3 $R/bench.py --config $CFG --pmc-run $ST > $O/${TAG}_pmc.log 2>&1 || { echo pmc $CFG $C failed; tail -5 $O/${TAG}_pmc.log; exit 1; }
    done
    SUF=$([ $CFG = dptn_av ] && echo "" || echo "_$CFG")
    (cd $R && python3 tools/pmc_table.py --forwards $NF --config $CFG --batch $BATCH --samples $SAMPLES --command "python3 bench.py --config $CFG --pmc-run $ST" --commit "$COMMIT" \
        --out $O/${TAG}_pmc_traffic${SUF}.json $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE | head -8)
    rm -rf $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE
  done
fi
if has mfma; then
  : > $O/${TAG}_mfma_utilisation.txt
  for CFG in dptn_av dptn_audio dprnn_av; do
    ST=$([ $CFG = dprnn_av ] && echo "--steps 1 --warmup 1" || echo "--steps 2 --warmup 1")
    timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_m -o m -- python3 $R/bench.py --config $CFG --pmc-run $ST > $O/${TAG}_pmc_m.log 2>&1 || { echo mfma $CFG failed; tail -5 $O/${TAG}_pmc_m.log; exit 1; }
    echo "== $CFG forward (bench.py --config $CFG --pmc-run $ST): kernels serialised by the counter pass (alone on the chip), commit $COMMIT ==" >> $O/${TAG}_mfma_utilisation.txt
    (cd $R && python3 tools/pmc_summary.py $O/${TAG}_pmc_m | python3 tools/mfma_util.py >> $O/${TAG}_mfma_utilisation.txt)
    echo >> $O/${TAG}_mfma_utilisation.txt
    rm -rf $O/${TAG}_pmc_m
  done
  head -12 $O/${TAG}_mfma_utilisation.txt
fi
if has train; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ttrace -o t -- python3 $R/bench.py --config dptn_av_train --pmc-run --steps 5 --warmup 2 > $O/${TAG}_ttrace.log 2>&1 || { echo train trace failed; tail -5 $O/${TAG}_ttrace.log; exit 1; }
  find $O/${TAG}_ttrace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_train_kernel_stats.csv
  rm -rf $O/${TAG}_ttrace
  timeout -k 10 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_tpmc -o m -- python3 $R/bench.py --config dptn_av_train --pmc-run --steps 1 --warmup 1 > $O/${TAG}_tpmc.log 2>&1 || { echo train mfma failed; tail -5 $O/${TAG}_tpmc.log; exit 1; }
  echo "== DPTN-AV training step (config 4), B = 16 as 8 + 8: kernels serialised by the counter pass, commit $COMMIT ==" > $O/${TAG}_train_mfma_utilisation.txt
  (cd $R && python3 tools/pmc_summary.py $O/${TAG}_tpmc | python3 tools/mfma_util.py >> $O/${TAG}_train_mfma_utilisation.txt)
  rm -rf $O/${TAG}_tpmc
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_tp_$C -o c -- python3 $R/bench.py --config dptn_av_train --pmc-run --steps 1 --warmup 1 > $O/${TAG}_tp.log 2>&1 || { echo train pmc $C failed; tail -5 $O/${TAG}_tp.log; exit 1; }
  done
  (cd $R && python3 tools/pmc_table.py --forwards 2 --config dptn_av_train --command "python3 bench.py --config dptn_av_train --pmc-run --steps 1 --warmup 1" --commit "$COMMIT" \
      --out $O/${TAG}_train_pmc_traffic.json $O/${TAG}_tp_FETCH_SIZE $O/${TAG}_tp_WRITE_SIZE | head -6)
  rm -rf $O/${TAG}_tp_FETCH_SIZE $O/${TAG}_tp_WRITE_SIZE
fi
if has bench; then
  # the bench line quotes the traffic tables: give it the ones just taken (same tree, same digest)
  for f in pmc_traffic.json pmc_traffic_dptn_audio.json pmc_traffic_dprnn_av.json train_pmc_traffic.json; do
    [ -f $O/${TAG}_$f ] && cp $O/${TAG}_$f $R/profiles/r05_$f
  done
fi
cd $R
if has bench; then
  timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { echo bench failed; tail -20 $O/${TAG}_bench.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/${TAG}_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["class"], d["roofline"]["frac"], d["roofline"]["launch_ms"], "train", d.get("train_step",{}).get("value"),
      "b1", d.get("latency_b1",{}).get("mean_ms"), "others", {k: v.get("value") for k, v in d.get("other_configs",{}).items()})
PY
fi
ls -la $O | grep ${TAG}_ | tail -20
