#!/usr/bin/env python3
"""Concurrency summary of a rocprofv3 kernel timeline (tools/gpu_train_timeline.sh): for the LAST complete step, how long
0 / 1 / 2 / 3 kernels were in flight, which kernels ran alone, and how long a recurrence (lstm16 / bptt) was the only
kernel on the chip.   usage: timeline_summary.py <kernel_trace.csv> [train|forward [sub-batches per forward]]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
if mode == "train":      # a step ends with its AdamW launches
    marks = sorted({e[1] for e in ev if "adamw" in e[2]})
    ends, last = [], None
    for b in marks:
        if last is None or b - last > 20e6:
            ends.append(b)
        last = b
    s0, s1 = ends[-2], ends[-1]
else:                    # a forward ends with the decoder gathers of its sub-batches
    marks = sorted(e[1] for e in ev if "decoder_gather" in e[2])
    per = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, len(marks) // 4)      # gathers per forward = its sub-batches
    s0, s1 = marks[-1 - per], marks[-1]
sel = [e for e in ev if e[0] >= s0 and e[1] <= s1 + 1e6]
pts = sorted([(a, 1, n) for a, b, n in sel] + [(b, -1, n) for a, b, n in sel])
act, cur, lastt = collections.Counter(), 0, s0
hist, alone, rhist = collections.Counter(), collections.Counter(), collections.Counter()
REC = ("lstm16", "bptt", "lstm_recurrence", "lstm4_kernel")      # (lstm16 also matches lstm16x / lstm16x128: the fused recurrences)
lstm_active = lstm_alone = 0
for t, d, n in pts:
    dt = t - lastt
    if dt > 0:
        hist[cur] += dt
        live = [k for k in act if act[k] > 0]
        is_lstm = any(any(r in k for r in REC) for k in live)
        rhist[sum(act[k] for k in live if any(r in k for r in REC))] += dt
        if cur == 1:
            alone[live[0][:70]] += dt
        if is_lstm:
            lstm_active += dt
            if cur == 1:
                lstm_alone += dt
    act[n] += d
    cur += d
    lastt = t
print(f"last step: {(s1 - s0) / 1e6:.2f} ms, {len(sel)} launches")
print("kernels in flight -> ms:", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
print(f"a recurrence in flight {lstm_active / 1e6:.2f} ms, of which as the ONLY kernel {lstm_alone / 1e6:.2f} ms")
print("recurrence launches in flight -> ms:", {k: round(v / 1e6, 2) for k, v in sorted(rhist.items())})
print("alone on the chip (ms):")
for k, v in alone.most_common(8):
    print(f"  {v / 1e6:7.2f}  {k}")
