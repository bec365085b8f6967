#!/usr/bin/env python3
"""down_proj of a short prompt pass (rows x 11008 -> 4096, 7B) in isolation: the product's route for row-major x (ops.gemm, AUTO) against the few-row kernel
over fragment-ordered x (ops.gemm_xfrag), cold rotating weights, hipGraph replay.  usage: down64_bench.py [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
from amq_amd.llama import _synthetic_linear
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
N, K = 4096, 11008
for bits in (2, 3, 4):
    copies = 40
    w = [_synthetic_linear(N, K, bits, gen, dev) for _ in range(copies)]
    x = torch.randn(rows, K, device=dev, generator=gen).half()
    xf = ops.xfrag(x, rows, K)
    res = torch.randn(rows, N, device=dev, generator=gen).half()
    y = torch.empty(rows, N, device=dev, dtype=torch.float16)
    forms = {"gemm AUTO (row-major x)": lambda l: ops.gemm(x, l.qn, l.mn, l.bits, l.mode, N, K, residual=res, out=y),
             "gemm_xfrag (fragment-ordered x)": lambda l: ops.gemm_xfrag(xf, rows, l.qn, l.mn, l.bits, l.mode, N, K, residual=res, out=y)}
    outs = {}
    for name, fn in forms.items():
        for l in w:
            fn(l)
        outs[name] = fn(w[0]).clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for i in range(40):
                    fn(w[i % copies])
        torch.cuda.current_stream().wait_stream(side)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record(); e1.synchronize()
        print(f"{bits} bit rows {rows}: {name:34s} {e0.elapsed_time(e1) * 1e3 / 120:7.2f} us per call", flush=True)
    a, b = list(outs.values())
    print("   max |diff| between the two:", (a.float() - b.float()).abs().max().item(), " same bits:", torch.equal(a, b))
    del w
