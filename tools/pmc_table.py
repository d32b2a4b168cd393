#!/usr/bin/env python3
"""Per-kernel HBM traffic table from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section).

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f -- python3 bench.py --pmc-run --steps 3 --warmup 1
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w -- python3 bench.py --pmc-run --steps 3 --warmup 1
    python3 tools/pmc_table.py --forwards 4 --commit <sha> --out profiles/r02_pmc_traffic.json gpurun_out/pmc_f gpurun_out/pmc_w

FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC counter budget), are reported in KiB, and on gfx950 FETCH_SIZE counts
exactly half of the bytes of wide (16 B/lane, LDS-DMA included) streaming reads -> doubled FOR THE KERNELS THAT READ THAT WAY
(WIDE_READERS below: every kernel of this library that streams its input with 16-byte loads or LDS-DMA); kernels with
narrower loads (gathers, reductions, torch helpers) keep the raw count.  Both columns are written (`fetch_raw_bytes`,
`fetch_bytes`), so the total is reproducible either way; WRITE_SIZE is exact.  Infinity
Cache hits are counted (memory-side counters of L2), so "HBM bytes" here means bytes crossing L2 <-> fabric.
`--forwards` = forwards the profiled command ran (warm-up + steps): launches per step = launches / forwards.  The
command runs NO isolated / event-profiling pass (bench.py --pmc-run), so per-kernel means are not mixed.
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--forwards", type=int, required=True)
ap.add_argument("--commit", default="?")
ap.add_argument("--config", default="dptn_av")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--samples", type=int, default=32000)
ap.add_argument("--command", default="python3 bench.py --pmc-run --steps 3 --warmup 1")
ap.add_argument("--out", required=True)
a = ap.parse_args()
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_separation_amd.build import source_digest  # noqa: E402  (the tree the profiled command ran from)

# kernels whose global reads are 16 B per lane (float4 / global_load_lds dwordx4) -- the gfx950 half-count applies to them
WIDE_READERS = ("lstm16_kernel", "lstm16x", "lstm16s_kernel", "lstm32s_kernel", "lstm_recurrence_kernel", "lstm_bptt", "gemm_ws_kernel", "fcln_kernel", "attn_block",
                "attention_kernel", "attention_long_kernel", "attention_bwd_kernel", "wgrad", "taps_fold_kernel", "attn_pack_kernel",
                "slab_reduce_frag_kernel", "grad_add_kernel", "adamw_kernel", "grad_clip", "pit_", "sisnr", "fold_decoder_kernel")

acc = defaultdict(lambda: defaultdict(list))
for d in a.dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))

kernels, total = [], 0.0
for name, c in acc.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    short = re.sub(r"^void (\(anonymous namespace\)::)?", "", re.sub(r"\(float.*|\(int.*", "", name))
    if short.startswith(("__amd_rocclr", "at::native")):       # runtime fills / torch helpers: not the path's kernels
        continue
    nf, nw = len(c["FETCH_SIZE"]), len(c["WRITE_SIZE"])
    fetch_raw = sum(c["FETCH_SIZE"]) / nf * 1024.0
    wide = any(w in short for w in WIDE_READERS)
    fetch = fetch_raw * (2.0 if wide else 1.0)
    write = sum(c["WRITE_SIZE"]) / nw * 1024.0
    per_step = nf / a.forwards
    kernels.append({"name": short, "launches_per_step": round(per_step, 3), "launches_sampled": nf,
                    "fetch_raw_bytes_per_launch": round(fetch_raw), "wide_reads": wide, "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
                    "hbm_bytes_per_launch": round(fetch + write), "hbm_bytes_per_step": round((fetch + write) * per_step)})
    total += (fetch + write) * per_step
kernels.sort(key=lambda r: -r["hbm_bytes_per_step"])
json.dump({"what": "HBM (L2 <-> fabric) bytes per kernel launch and per forward step, rocprofv3 PMC, kernels serialised by the "
                   "counter pass", "commit": a.commit, "csrc_digest": source_digest(), "command": a.command, "config": a.config, "batch": a.batch,
           "samples": a.samples, "forwards_profiled": a.forwards,
           "corrections": "KiB -> bytes; FETCH_SIZE x2 for the kernels that read 16 B per lane / through LDS-DMA (gfx950 tallies those at "
                          "half; `wide_reads` per row), raw for the others; WRITE_SIZE exact",
           "bytes_per_step": round(total), "kernels": kernels}, open(a.out, "w"), indent=1)
print(f"{a.out}: {total / 1e9:.2f} GB per step over {len(kernels)} kernels")
for r in kernels[:12]:
    print(f"  {r['hbm_bytes_per_step'] / 1e9:8.2f} GB/step  {r['launches_per_step']:6.1f} x {r['hbm_bytes_per_launch'] / 1e6:9.1f} MB  {r['name'][:90]}")
