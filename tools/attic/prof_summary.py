#!/usr/bin/env python3
"""Summarise rocprofv3 csv output dirs: per-kernel avg duration and mean counters."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "gemv"
import json
for f in sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True)):
    d = defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in d.items():
        if filt in k:
            v2 = sorted(v)[len(v) // 4:]
            print(f"TRACE {k[:70]:70s} n={len(v):4d} avg_us={sum(v)/len(v):8.2f} med_us={sorted(v)[len(v)//2]:8.2f} min_us={min(v):8.2f}")
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    d = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in d.items():
        if filt in k:
            print("PMC  ", k[:70], {n: round(sum(v) / len(v), 1) for n, v in c.items()})
