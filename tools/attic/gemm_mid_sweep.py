#!/usr/bin/env python3
"""mid-size many-row launches (7B / 13B prompts of 512 .. 4096 rows): TFLOP/s of every hand-written GEMM route (0 = what launch_gemm picks)
usage: gemm_mid_sweep.py [bits]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
from amq_amd.llama import _synthetic_linear
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
for (n, k) in ((4096, 4096), (11008, 4096), (4096, 11008), (5120, 5120), (13824, 5120), (5120, 13824)):
    l = _synthetic_linear(n, k, bits, gen, dev)
    for m in (512, 1024, 2048, 4096):
        x = (torch.randn(m, k, device=dev, generator=gen) * 0.5).half()
        y = torch.empty(m, n, device=dev, dtype=torch.float16)
        out = {"N": n, "K": k, "M": m}
        for r in (0, 1, 3, 4, 5):
            ts = []
            for i in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.gemm(x, l.qn, l.mn, bits, l.mode, n, k, out=y, route=r); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            out[f"TF_{r}"] = round(2.0 * m * n * k / sorted(ts[2:])[2] / 1e9)
        print(json.dumps(out), flush=True)
