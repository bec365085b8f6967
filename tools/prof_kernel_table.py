#!/usr/bin/env python3
"""per-kernel totals of a rocprofv3 --kernel-trace csv dir: calls, avg us, total us, share (kernel names shortened)"""
import csv, glob, re, sys
from collections import defaultdict
root = sys.argv[1]
d = defaultdict(list)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    name = re.sub(r"\(.*", "", k).replace("void amq::", "amq::")[:90]
    print(f"{name:90s} n={len(v):6d} avg_us={sum(v)/len(v):9.2f} total_ms={sum(v)/1e3:9.3f} share={100*sum(v)/tot:5.1f}%")
