#!/usr/bin/env python3
"""GEMV launches (one row, hipGraph of 64 launches over rotating weight copies, HIP events) over the same layer quantized with groups of
128, 64 and 32: what the finer groups' second / fourth (scale, zero) pair per tile row costs the weight-streaming kernel.
usage: fine_group_gemv.py [--shapes N,K;N,K] [--bits 2,3,4]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
from amq_amd.hqq_format import random_hqq

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="4096,4096;12288,4096;22016,4096;4096,11008")
ap.add_argument("--bits", default="2,3,4")
args = ap.parse_args()
dev = torch.device("cuda:0")
COPIES = 8
for shp in args.shapes.split(";"):
    n, k = (int(v) for v in shp.split(","))
    for bits in (int(b) for b in args.bits.split(",")):
        row = {"N": n, "K": k, "bits": bits}
        for group in (128, 64, 32):
            ws = []
            for c in range(COPIES):
                h = random_hqq(n, k, bits, seed=c, group=group).to(dev)
                ws.append(ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), bits, n, k, group=group))
            x = torch.randn(1, k, device=dev).half()
            y = torch.empty(1, n, device=dev, dtype=torch.float16)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for qn, mn in ws:
                    ops.gemv(x, qn, mn, bits, ops.MODE_HQQ, n, k, out=y)
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for i in range(64):
                        qn, mn = ws[i % COPIES]
                        ops.gemv(x, qn, mn, bits, ops.MODE_HQQ, n, k, out=y)
                for _ in range(3):
                    g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(20):
                    g.replay()
                e1.record(st)
                e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (20 * 64)
            nbytes = n * k * bits / 8 + n * k / group * 4
            row[f"us_g{group}"] = round(us, 2)
            row[f"TBs_g{group}"] = round(nbytes / us / 1e6, 3)
        print(json.dumps(row), flush=True)
