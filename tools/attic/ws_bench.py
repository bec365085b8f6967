#!/usr/bin/env python3
"""time GEMM routes on the configs[3] shapes at M = 32768 (A/B of build variants: python tools/with_variant.py <tag> tools/attic/ws_bench.py [routes] [NxK,NxK])"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
from amq_amd.llama import _synthetic_linear
routes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "5").split(",")]
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
shapes = ((13824, 5120), (5120, 13824)) if len(sys.argv) < 3 else tuple(tuple(int(v) for v in t.split("x")) for t in sys.argv[2].split(","))
for (n, k) in shapes:
    for bits in (4, 3, 2):
        l = _synthetic_linear(n, k, bits, gen, dev)
        x = (torch.randn(32768, k, device=dev, generator=gen) * 0.5).half()
        y = torch.empty(32768, n, device=dev, dtype=torch.float16)
        out = {"N": n, "K": k, "bits": bits}
        for r in routes:
            ts = []
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.gemm(x, l.qn, l.mn, bits, l.mode, n, k, out=y, route=r); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            t = sorted(ts[1:])[2]
            out[f"TF_{r}"] = round(2.0 * 32768 * n * k / t / 1e9, 1)
        print(json.dumps(out), flush=True)
