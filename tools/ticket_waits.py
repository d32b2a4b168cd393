#!/usr/bin/env python3
"""Build check: does any token-tile kernel wait for its tile-ticket atomic right where it is issued?

The GEMM engine / weight-gradient kernels request the ticket of the tile after the next one in front of their MFMA block
and publish it an iteration later, so the atomic's round trip is hidden -- unless the register allocator parks the result in
an AGPR, which needs the value at once (`s_waitcnt vmcnt(0)` directly behind `global_atomic_add`: the A-tile prefetch's
HBM latency and the atomic's round trip exposed on every tile).  This script compiles csrc/dptnav.hip to assembly and
lists, per kernel, its ticket atomics ('.' = free, 'X' = waited for within three instructions); the third one of a kernel
is the one inside the tile loop.    python3 tools/ticket_waits.py [--all]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "speech_separation_amd", "csrc", "dptnav.hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "dptnav.s")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm",
                    "-amdgpu-atomic-optimizer-strategy=None", "-S", "--cuda-device-only", "-w", src, "-o", out], check=True)
    lines = open(out).read().split("\n")
cur, res = None, {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
    if "global_atomic_add" in l and cur:
        block = []                      # the rest of the atomic's basic block, three instructions at most
        for nxt in lines[i + 1:i + 4]:
            if "s_cbranch" in nxt or re.match(r"^\.?\w+:", nxt.strip()):
                break
            block.append(nxt)
        res.setdefault(cur, []).append("vmcnt(0)" in " ".join(block))
names = subprocess.run(["c++filt"], input="\n".join(res), capture_output=True, text=True).stdout.split("\n")
bad = 0
for (k, v), n in zip(res.items(), names):
    loop_exposed = len(v) >= 3 and v[2]
    bad += loop_exposed
    if loop_exposed or "--all" in sys.argv:
        print("".join("X" if e else "." for e in v), n[:160])
print(f"{len(res)} kernels with ticket atomics, {bad} wait for the one inside their tile loop")
sys.exit(1 if bad else 0)
