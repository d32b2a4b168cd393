#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (one line per kernel)."""
import re
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
keys = [('vgpr', r'VGPRs'), ('agpr', r'AGPRs'), ('scratch', r'ScratchSize \[bytes/lane\]'),
        ('occ', r'Occupancy \[waves/SIMD\]'), ('lds', r'LDS Size \[bytes/block\]')]
for b in blocks:
    name = b.split('\n')[0].strip()
    vals = []
    for label, k in keys:
        m = re.search(k + r': (\S+)', b)
        vals.append(f"{label}={m.group(1) if m else '?'}")
    print(f"{name[:90]:92s} " + ' '.join(vals))
