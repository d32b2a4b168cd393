#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs: mean counter value per kernel name.  usage: pmc_summary.py <dir> [...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\((?!anonymous).*", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))[:60]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")
