#!/usr/bin/env python3
"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass
(tools/pmc_summary.py output on stdin or as file argument).  PMC passes serialise the kernels, so every figure is the
kernel ALONE on the chip.  SQ_VALU_MFMA_BUSY_CYCLES sums busy cycles over the chip's 1024 SIMDs (256 CUs x 4);
GRBM_GUI_ACTIVE sums active cycles over the 8 XCDs:   utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)."""
import re
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
rows, name, vals = [], None, {}
for line in txt.splitlines():
    m = re.match(r"\s+(\w+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
    if m:
        vals[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    else:
        if name and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
            rows.append((name, vals))
        name, vals = line.strip(), {}
if name and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
    rows.append((name, vals))
print(f"{'kernel':78s} {'launches':>8s} {'active cyc/XCD':>15s} {'MFMA busy cyc':>15s} {'MFMA util':>9s}")
for name, v in sorted(rows, key=lambda r: -r[1]["GRBM_GUI_ACTIVE"][0] * r[1]["GRBM_GUI_ACTIVE"][1]):
    busy, n = v["SQ_VALU_MFMA_BUSY_CYCLES"]
    act = v["GRBM_GUI_ACTIVE"][0] / 8.0
    if busy <= 0:
        continue
    print(f"{name[:78]:78s} {n:8d} {act:15.0f} {busy:15.0f} {busy / (act * 1024.0):9.3f}")
