#!/usr/bin/env python3
"""The bfloat16 few-row kernel (amq::gemv_bf16_kernel) beside the fp16 GEMV on the 7B layer shapes: us per launch over weights cold in HBM
(every launch its own copy of the layer, > 512 MB in rotation; one hipGraph of all launches, replayed), HIP events, and the share of the 8 TB/s roofline on the algorithmic bytes
(BASELINE.md section 3).  usage: bf16_bench.py [rows ...]   (default 1 8)"""
import json
import sys

import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from amq_amd import ops
from amq_amd.hqq_format import random_hqq

SHAPES = [("q/k/v/o", 4096, 4096), ("gate/up", 11008, 4096), ("down", 4096, 11008)]


def layer_bytes(bits, n, k, m):
    return n * k * bits // 8 + 4 * n * k // 128 + 2 * m * k + 2 * m * n


def time_launches(fn, copies, reps=5):
    """one hipGraph holding the `copies` launches (a Python call per launch costs more than the kernel), best replay of `reps`"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for c in range(copies):
            fn(c)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / copies)
    return best


def main():
    rows = [int(a) for a in sys.argv[1:]] or [1, 8]
    dev = torch.device("cuda:0")
    out = []
    for name, n, k in SHAPES:
        for bits in (4, 3, 2):
            per = n * k * bits // 8
            copies = max(8, (600 << 20) // per)
            h = random_hqq(n, k, bits, seed=bits).to(dev)
            s, z = h.scale.reshape(-1).contiguous(), h.zero.reshape(-1).contiguous()
            f16 = [ops.repack_from_hqq(h.W_q, s, z, bits, n, k) for _ in range(copies)]
            sb, zb = s.float().to(torch.bfloat16), z.float().to(torch.bfloat16)
            qb, mb = ops.repack_from_hqq(h.W_q, sb, zb, bits, n, k)
            b16 = [(q, mb) for q, _ in f16]                      # the payloads are shared: only the meta differs
            for m in rows:
                x16 = torch.randn(m, k, device=dev).half()
                xb = x16.to(torch.bfloat16)
                y16 = torch.empty(m, n, dtype=torch.float16, device=dev)
                yb = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
                t16 = time_launches(lambda c: ops.gemv(x16, f16[c][0], f16[c][1], bits, ops.MODE_HQQ, n, k, out=y16), copies)
                tb = time_launches(lambda c: ops.linear_bf16(xb, b16[c][0], b16[c][1], bits, n, k, out=yb), copies)
                by = layer_bytes(bits, n, k, m)
                rec = {"layer": name, "N": n, "K": k, "bits": bits, "rows": m, "us_fp16": round(t16, 2), "us_bf16": round(tb, 2),
                       "frac_fp16": round(by / t16 / 8e6, 3), "frac_bf16": round(by / tb / 8e6, 3)}
                print(json.dumps(rec), flush=True)
                out.append(rec)
            del f16, b16
            torch.cuda.empty_cache()


def gemm_leg(m=8192, n=13824, k=5120, bits=3):
    """the batched end: dequantize once + the MFMA-bound GEMM kernel, bf16 instantiation beside the fp16 one (13B gate/up shape)"""
    dev = torch.device("cuda:0")
    h = random_hqq(n, k, bits, seed=1).to(dev)
    s, z = h.scale.reshape(-1).contiguous(), h.zero.reshape(-1).contiguous()
    q16, m16 = ops.repack_from_hqq(h.W_q, s, z, bits, n, k)
    qb, mb = ops.repack_from_hqq(h.W_q, s.float().to(torch.bfloat16), z.float().to(torch.bfloat16), bits, n, k)
    x16 = torch.randn(m, k, device=dev).half()
    xb = x16.to(torch.bfloat16)
    y16 = torch.empty(m, n, dtype=torch.float16, device=dev)
    yb = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    t16 = time_launches(lambda c: ops.gemm(x16, q16, m16, bits, ops.MODE_HQQ, n, k, out=y16, route=ops.GEMM_DEQ), 4)
    tb = time_launches(lambda c: ops.linear_bf16(xb, qb, mb, bits, n, k, out=yb), 4)
    fl = 2.0 * m * n * k
    print(json.dumps({"gemm": [m, n, k], "bits": bits, "us_fp16": round(t16, 1), "us_bf16": round(tb, 1),
                      "tflops_fp16": round(fl / t16 / 1e6, 1), "tflops_bf16": round(fl / tb / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--gemm":
        gemm_leg()
        sys.exit(0)
    main()
