#!/usr/bin/env python3
"""usage: prof_gemm.py M N K bits [iters] [route] -- repeated GEMM launches for rocprofv3 (route: 1 tiled, 3 ring)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
ops.LIB_GEMM_ROWS = 0        # time / check the fused unpack + MFMA kernel itself
from amq_amd.hqq_format import random_hqq
m, n, k, bits = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
route = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = torch.device("cuda:0")
h = random_hqq(n, k, bits, seed=1).to(dev)
qn, mn = ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), bits, n, k)
x = torch.randn(m, k, device=dev).half()
y = torch.empty(m, n, device=dev, dtype=torch.float16)
if route < 0:                 # the dequantize + library GEMM route
    ops.LIB_GEMM_ROWS, route = 1, 0
for _ in range(iters):
    ops.gemm(x, qn, mn, bits, 0, n, k, out=y, route=route)
torch.cuda.synchronize()
print("done")
