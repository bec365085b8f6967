#!/usr/bin/env python3
"""Run one GEMV shape a few dozen times over rotating weight buffers (for
rocprofv3 --kernel-trace / --pmc).  usage: prof_gemv.py N K bits [M] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops  # noqa: E402
from amq_amd.hqq_format import random_hqq  # noqa: E402

n, k, bits = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = int(sys.argv[4]) if len(sys.argv) > 4 else 1
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 40
dev = torch.device("cuda:0")
h = random_hqq(n, k, bits, seed=1).to(dev)
qn0, mn0 = ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), bits, n, k)
per = qn0.numel() * 4 + mn0.numel() * 2
copies = max(2, min(64, (768 << 20) // per + 1))
bufs = [(qn0.clone(), mn0.clone()) for _ in range(copies)]
x = torch.randn(m, k, device=dev).half()
y = torch.empty(m, n, device=dev, dtype=torch.float16)
for i in range(iters):
    q, mt = bufs[i % copies]
    ops.gemv(x, q, mt, bits, ops.MODE_HQQ, n, k, out=y)
torch.cuda.synchronize()
print("done", n, k, bits, m, iters)
