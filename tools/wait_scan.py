#!/usr/bin/env python3
"""Static scan of generated gfx950 assembly for waits that sit out a request just issued (round 5; the lstm16x128 finding: the
compiler put `s_waitcnt vmcnt(0)` eleven instructions behind two LDS-DMA requests, every step).  For every kernel and every loop
body it lists the `s_waitcnt vmcnt(N)` whose youngest COVERED vector-memory instruction (the (N+1)-th last one issued before it, in
program order inside the loop) is closer than `--near` instructions: such a wait pays most of a memory round trip.

    hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o k.s csrc/<unit>.hip ; python3 tools/wait_scan.py k.s [--near 150]
"""
import re
import sys

near = 150
files = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--near":
        near = int(args.pop(0))
    else:
        files.append(a)
VMEM = re.compile(r"^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store)")
for path in files:
    kern = None
    body = []
    def flush():
        if not kern or not body:
            return
        # loops: label lines mentioning "Loop Header"; a loop body = from header label to the last branch back to it
        labels = {l.split(":")[0].strip(): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        heads = [(i, l.split(":")[0].strip()) for i, l in enumerate(body) if "Loop Header" in l and re.match(r"^\.LBB", l)]
        for hi, name in heads:
            ends = [i for i, l in enumerate(body) if re.search(r"s_c?branch\w*\s+" + re.escape(name) + r"\b", l) and i > hi]
            if not ends:
                continue
            lo, hi2 = hi, max(ends)
            ins = [(i, l.strip()) for i, l in enumerate(body[lo:hi2 + 1], lo) if l.startswith("\t") and not l.strip().startswith(";")]
            vm = []
            nmf = sum(1 for _, l in ins if l.startswith("v_mfma"))
            for k, (i, l) in enumerate(ins):
                if VMEM.match(l):
                    vm.append(k)
                m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
                if m and vm:
                    n = int(m.group(1))
                    if n < len(vm):
                        covered = vm[len(vm) - 1 - n]
                        dist = k - covered
                        if dist < near:
                            mf = sum(1 for _, x in ins[covered:k] if x.startswith("v_mfma"))
                            print(f"{path}: {kern[:90]}: loop {name} ({hi2 - lo} lines, {nmf} MFMAs): vmcnt({n}) {dist} instructions "
                                  f"({mf} MFMAs) behind `{ins[covered][1][:60]}`")
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush()
            kern, body = m.group(1), []
        elif kern is not None:
            body.append(line.rstrip("\n"))
    flush()
