#!/usr/bin/env python3
"""own MFMA GEMM vs (bit-exact dequantize kernel -> fp16 scratch) + library GEMM, per row count: where does handing the
dequantized weights to hipBLASLt win?  usage: lib_gemm_crossover.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
from amq_amd.hqq_format import random_hqq

dev = torch.device("cuda:0")


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for n, k in ((5120, 5120), (13824, 5120), (5120, 13824), (4096, 4096)):
    for bits in (3,):
        h = random_hqq(n, k, bits, seed=1).to(dev)
        qn, mn = ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), bits, n, k)
        w = torch.empty(n, k, dtype=torch.float16, device=dev)
        for m in (256, 512, 1024, 2048, 4096, 16384, 32768):
            x = torch.randn(m, k, device=dev).half()
            y = torch.empty(m, n, dtype=torch.float16, device=dev)
            iters = 50 if m <= 4096 else 10
            ops.LIB_GEMM_ROWS = 0                    # the fused unpack + MFMA kernel itself
            t_own = timeit(lambda: ops.gemm(x, qn, mn, bits, ops.MODE_HQQ, n, k, out=y), iters)

            def lib():
                ops.dequantize(qn, mn, bits, ops.MODE_HQQ, n, k, out=w)
                torch.matmul(x, w.t(), out=y)
            t_lib = timeit(lib, iters)
            t_deq = timeit(lambda: ops.dequantize(qn, mn, bits, ops.MODE_HQQ, n, k, out=w), iters)
            fl = 2.0 * m * n * k
            print(f"{n}x{k} b{bits} M={m:6d}: own {t_own:9.1f} us ({fl/t_own/1e6:7.1f} TF)   dequant+lib {t_lib:9.1f} us ({fl/t_lib/1e6:7.1f} TF)   dequant alone {t_deq:6.1f} us", flush=True)
