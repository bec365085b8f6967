#!/usr/bin/env python3
"""gpurun_out/<round> (raw rocprofv3 output of tools/collect_round.sh) -> <dst>/<round>_* (tracked summaries).
usage: make_profile_summary.py <round> [dst=profiles]"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", R)
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")   # on the GPU box: a directory under gpurun_out/
os.makedirs(dst, exist_ok=True)

for name in ("bench.json", "bench_config4.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{R}_{name}"))
ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    rows = list(csv.reader(open(ks[0])))
    with open(os.path.join(dst, f"{R}_bench_kernel_stats.csv"), "w") as f:
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
with open(os.path.join(dst, f"{R}_pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1, sort_keys=True)

# HBM traffic per GEMV launch: FETCH_SIZE is in KiB and reports exactly half of a wide coalesced streaming read on
# gfx950 (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE is exact.  Per token the bench workload launches
# 64 x gemv<PRO_RMSNORM> (q/k/v, gate/up), 32 x gemv<PRO_NONE> (o_proj), 32 x gemv<PRO_SILU_MUL> (down_proj): the first
# template argument of the kernel symbol is the prologue (1 / 0 / 2).
def hbm(sym):
    d = summary[sym]
    return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0


g = {k: v for k, v in summary.items() if "gemv_kernel<" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v}
w = {"<1,": 64, "<0,": 32, "<2,": 32}
per_pro = defaultdict(list)
for k in g:
    for tag in w:
        if "gemv_kernel" + tag in k:
            per_pro[tag].append(hbm(k))
if len(per_pro) == 3:
    tot = sum(w[tag] * sum(v) / len(v) for tag, v in per_pro.items())
    out = {"hbm_bytes_per_launch": tot / 128.0,
           "per_symbol_hbm_bytes": {k: hbm(k) for k in g},
           "round": R,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_decode.py "
                     "(eager launches of the bench workload, current kernels: symbols listed); (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                     "per dispatch, averaged per symbol and weighted 64/32/32 per token"}
    # VALU instructions per wave of every GEMV symbol seen in the SQ passes (the default arithmetic's kernels and -- second pass, `prof_decode.py 4 gs`
    # -- the opt-in group-scale kernels: fourth template argument 3)
    out["valu_insts_per_wave"] = {k: round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1) for k, v in summary.items()
                                  if "gemv_kernel<" in k and v.get("SQ_WAVES") and "SQ_INSTS_VALU" in v}
    with open(os.path.join(dst, f"{R}_gemv_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:1200])
else:
    print("no complete FETCH/WRITE data for the three GEMV prologue variants:", list(g))
