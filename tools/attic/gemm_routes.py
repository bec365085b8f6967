#!/usr/bin/env python3
"""Many-row GEMM: the hand-written kernel families against dequantize + library GEMM, one process, interleaved rounds
(HIP events on the launch stream; random operands; checks every route against an fp32 matmul on the dequantized weights).
usage: gemm_routes.py [--shapes N,K;N,K] [--m 4096,32768] [--bits 3,4] [--rounds 5]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
from amq_amd.llama import _synthetic_linear

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="13824,5120;5120,5120;5120,13824")
ap.add_argument("--m", default="2048,32768")
ap.add_argument("--bits", default="3,4")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--routes", default="1,3,lib", help="1 tiled, 3 ring (rows by shape), 4 ring 128-row tiles, lib")
args = ap.parse_args()
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)


def run(route, x, l, y):
    if route == "lib":
        ops.LIB_GEMM_ROWS = 1
        ops.gemm(x, l.qn, l.mn, l.bits, l.mode, l.N, l.K, out=y)
        ops.LIB_GEMM_ROWS = 0
    else:
        ops.gemm(x, l.qn, l.mn, l.bits, l.mode, l.N, l.K, out=y, route=int(route))


ops.LIB_GEMM_ROWS = 0
routes = args.routes.split(",")
for shp in args.shapes.split(";"):
    n, k = (int(v) for v in shp.split(","))
    for bits in (int(v) for v in args.bits.split(",")):
        l = _synthetic_linear(n, k, bits, gen, dev)
        w = ops.dequantize(l.qn, l.mn, bits, l.mode, n, k)
        for m in (int(v) for v in args.m.split(",")):
            x = (torch.randn(m, k, device=dev, generator=gen) * 0.5).half()
            y = torch.empty(m, n, device=dev, dtype=torch.float16)
            ref = x[:256].float() @ w.float().t()
            rms = ref.pow(2).mean().sqrt().item()
            times = {r: [] for r in routes}
            errs = {}
            for r in routes:
                run(r, x, l, y)
                torch.cuda.synchronize()
                errs[r] = ((y[:256].float() - ref).abs().max().item()) / rms
            for _ in range(args.rounds):
                for r in routes:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run(r, x, l, y)
                    e1.record()
                    torch.cuda.synchronize()
                    times[r].append(e0.elapsed_time(e1) * 1e3)
            fl = 2.0 * m * n * k
            out = {"N": n, "K": k, "bits": bits, "M": m}
            for r in routes:
                t = sorted(times[r])
                out[f"us_{r}"] = round(t[len(t) // 2], 1)
                out[f"TF_{r}"] = round(fl / t[len(t) // 2] / 1e6, 1)
                out[f"err_{r}"] = round(errs[r], 5)
            print(json.dumps(out), flush=True)
