#!/usr/bin/env python3
"""Table of a tools/pmc_summary.py dump of the SQ wait counters: per kernel, the share of wave lifetime parked at
s_waitcnt / barriers (WAIT_ANY), stalled at issue incl. a busy MFMA pipe (WAIT_INST_ANY) and issuing (ACTIVE_INST_ANY);
LDS issue stalls and bank-conflict cycles; MFMA-busy share of the wave lifetime (busy cycles / 4 SIMD-cycles per quad)."""
import sys

ker, cur = {}, None
for line in open(sys.argv[1]):
    line = line.rstrip("\n")
    if line and not line.startswith(" "):
        cur = line
        ker[cur] = {}
    elif line.strip():
        p = line.split()
        ker[cur][p[0]] = (float(p[1]), int(p[2].strip("(n=)")))
rows = []
for k, v in ker.items():
    wc = v.get("SQ_WAVE_CYCLES", (0, 0))[0]
    if wc > 0:
        rows.append((wc * v["SQ_WAVE_CYCLES"][1], k, v, wc))
print(f"{'kernel':62s} {'n':>5s} {'wave quad-cyc':>14s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'w_lds':>6s} {'a_lds':>6s} {'bankc':>6s}")
for tot, k, v, wc in sorted(rows, reverse=True)[:40]:
    f = lambda c: v.get(c, (0,))[0] / wc
    print(f"{k[:62]:62s} {v['SQ_WAVE_CYCLES'][1]:5d} {wc:14.0f} {f('SQ_WAIT_ANY'):8.3f} {f('SQ_WAIT_INST_ANY'):9.3f} {f('SQ_ACTIVE_INST_ANY'):7.3f} "
          f"{f('SQ_WAIT_INST_LDS'):6.3f} {f('SQ_ACTIVE_INST_LDS'):6.3f} {f('SQ_LDS_BANK_CONFLICT'):6.3f}")
