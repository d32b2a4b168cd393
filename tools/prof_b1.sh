set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; export TMPDIR=/tmp
cd /tmp
for V in 0; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1prof_$V -o t -- python3 $R/bench.py --batch 1 --pmc-run --steps 20 --warmup 3 > $O/b1prof_$V.log 2>&1 || { echo failed $V; tail -5 $O/b1prof_$V.log; exit 1; }
  find $O/b1prof_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/b1_kernel_stats.csv
  rm -rf $O/b1prof_$V
done
