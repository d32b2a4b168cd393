#!/bin/bash
# GPU box: per-kernel device time of the training step (config 4) for two BUILDS of the library, by rocprofv3 --kernel-trace --stats
# of the same command: the tree's libdptnav.so and another one (default: speech_separation_amd/libdptnav_base.so).
#   tools/gpu_train_ab_trace.sh [other.so]      -> gpurun_out/trainab_{tree,other}_kernel_stats.csv
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
OTHER=${1:-$R/speech_separation_amd/libdptnav_base.so}
cd /tmp
for LEG in tree other; do
  if [ $LEG = other ]; then export DPTNAV_LIB=$OTHER; else unset DPTNAV_LIB; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trainab_$LEG -o t -- python3 $R/bench.py --config dptn_av_train --pmc-run --steps 5 --warmup 2 > $O/trainab_$LEG.log 2>&1 || { echo "$LEG failed"; tail -5 $O/trainab_$LEG.log; exit 1; }
  find $O/trainab_$LEG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/trainab_${LEG}_kernel_stats.csv
  rm -rf $O/trainab_$LEG
done
python3 - <<PY
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"][:100]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    return d
a, b = load("$O/trainab_other_kernel_stats.csv"), load("$O/trainab_tree_kernel_stats.csv")
print(f"{'kernel':100s} {'other ms':>10s} {'tree ms':>10s}  (totals over the profiled steps)")
for k in sorted(set(a) | set(b), key=lambda k: -max(a.get(k, (0, 0))[1], b.get(k, (0, 0))[1]))[:22]:
    print(f"{k:100s} {a.get(k, (0, 0))[1]:10.2f} {b.get(k, (0, 0))[1]:10.2f}")
print(f"{'sum':100s} {sum(v[1] for v in a.values()):10.2f} {sum(v[1] for v in b.values()):10.2f}")
PY
