#!/usr/bin/env python3
"""One grouped few-row launch (q/k/v or gate/up of a short prompt pass) in isolation: rotating cold weight copies, hipGraph replay, HIP events.
usage: fewrow_bench.py [rows=64] [forms: list of form:blocks, e.g. 1:0,2:3,2:6]   (form 1 = skinny grouped kernel, 2 = streaming kernel; blocks per workgroup)
Prints us per launch for the 7B q/k/v (3 x 4096 x 4096) and gate/up (2 x 11008 x 4096) launches at 3 bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops
from amq_amd.llama import _synthetic_linear

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
forms = [tuple(int(v) for v in f.split(":")) for f in (sys.argv[2] if len(sys.argv) > 2 else "1:0,2:3,2:6").split(",")]
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
K = 4096
for name, segs in (("q/k/v", [(4096, K)] * 3), ("gate/up", [(11008, K)] * 2)):
    per = sum(n for n, _ in segs) * K * 3 // 8
    copies = max(2, min(48, (512 << 20) // per + 1))
    w = [[_synthetic_linear(n, k, 3, gen, dev) for n, k in segs] for _ in range(copies)]
    x = torch.randn(rows, K, device=dev, generator=gen).half()
    xf = ops.xfrag(x, rows, K)
    ys = [torch.empty(rows, n, device=dev, dtype=torch.float16) for n, _ in segs]
    for form, blocks in forms:
        def launches(iters=48):
            for i in range(iters):
                ops.gemm_xfrag_grouped(xf, rows, [l.seg(y) for l, y in zip(w[i % copies], ys)], K, form=form, blocks_per_wg=blocks)
        launches(copies)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                launches()
        torch.cuda.current_stream().wait_stream(side)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record(); e1.synchronize()
        print(f"{name:8s} rows {rows:4d} form {form} blocks/wg {blocks}: {e0.elapsed_time(e1) * 1e3 / (3 * 48):7.2f} us per launch", flush=True)
    del w
