#!/usr/bin/env python3
"""gpurun_out/r01 (raw rocprofv3 output) -> profiles/r01_* (tracked summaries)."""
import csv
import glob
import json
import os
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r01")
import sys
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")   # on the GPU box: a directory under gpurun_out/
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "r01_bench.json"))
ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.reader(open(ks)))
with open(os.path.join(dst, "r01_bench_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if "amq" in r[0]:
            w.writerow(r)

cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in cnt.items() if "amq" in k}
with open(os.path.join(dst, "r01_pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1, sort_keys=True)

# HBM traffic per GEMV launch: FETCH_SIZE is in KiB and reports exactly half of a wide coalesced streaming read on
# gfx950 (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE is exact.  Weighted by launches per token:
# 64 x gemv<RMSNORM> (q/k/v, gate/up), 32 x gemv<NONE> (o_proj), 32 x gemv<SILU_MUL> (down_proj).
def hbm(sym):
    d = summary[sym]
    return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
g = {k: v for k, v in summary.items() if "gemv_kernel<" in k}
w = {"<1,": 64, "<0,": 32, "<2,": 32}
tot = sum(hbm(k) * n for k in g for tag, n in w.items() if tag in k)
out = {"hbm_bytes_per_launch": tot / 128.0,
       "per_symbol_hbm_bytes": {k: hbm(k) for k in g},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_decode.py "
                 "(eager launches of the bench workload); (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch, averaged "
                 "per symbol and weighted 64/32/32 per token"}
with open(os.path.join(dst, "r01_gemv_pmc.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
