#!/usr/bin/env python3
"""Kernel timeline of the LAST bs = 1 forward in a rocprofv3 kernel trace (tools/gpu_b1_timeline.sh): device time per
kernel class, idle gaps between consecutive kernels, span.   usage: b1_timeline_summary.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ends = [i for i, e in enumerate(ev) if "decoder_gather" in e[2]]
i0, i1 = ends[-2] + 1, ends[-1]
sel = ev[i0:i1 + 1]
span = sel[-1][1] - sel[0][0]
busy = collections.Counter()
cnt = collections.Counter()
gap = 0
for k, (a, b, n) in enumerate(sel):
    short = n.split("(")[0][:90]
    busy[short] += b - a
    cnt[short] += 1
    if k:
        gap += max(0, a - sel[k - 1][1])
print(f"last forward: {len(sel)} launches, span {span / 1e3:.1f} us, kernels {sum(busy.values()) / 1e3:.1f} us, idle between kernels {gap / 1e3:.1f} us")
for n, v in busy.most_common(14):
    print(f"  {v / 1e3:8.1f} us  {cnt[n]:3d} x {v / cnt[n] / 1e3:7.1f} us  {n}")
