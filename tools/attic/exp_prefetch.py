#!/usr/bin/env python3
"""Experiment: what is a GEMV launch worth when the head of every row-tile (what the workgroups request first) is
already in the Infinity Cache?  cold = rotating buffers; head = first `frac` of each row-tile's bytes touched by a
framework reduction just before the launch; hot = everything touched."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
from amq_amd.hqq_format import random_hqq

dev = torch.device("cuda:0")


def run(n, k, bits, frac):
    h = random_hqq(n, k, bits, seed=1).to(dev)
    qn0, mn0 = ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), bits, n, k)
    copies = 24
    bufs = [(qn0.clone(), mn0.clone()) for _ in range(copies)]
    x = torch.randn(1, k, device=dev).half()
    y = torch.empty(1, n, device=dev, dtype=torch.float16)
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    rt_bytes = (k // 128) * 64 * 4 * bits
    res = {}
    for mode in ("cold", "head", "hot"):
        ts = []
        for it in range(12):
            q, m = bufs[it % copies]
            junk.fill_(it)                                    # evict L2 / Infinity Cache
            if mode != "cold":
                v = q.view(torch.uint8).view(n // 16, rt_bytes)
                f = 1.0 if mode == "hot" else frac
                _ = v[:, : int(rt_bytes * f)].to(torch.int32).sum()
                _ = m.view(torch.int32).sum() if mode == "hot" else None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.gemv(x, bufs[(it + 5) % copies][0], bufs[(it + 5) % copies][1], bits, ops.MODE_HQQ, n, k, out=y)   # predecessor kernel (other weights)
            e0.record()
            ops.gemv(x, q, m, bits, ops.MODE_HQQ, n, k, out=y)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    print(f"{n}x{k} b{bits} frac {frac}: " + "  ".join(f"{m} {t:.2f} us" for m, t in res.items()), flush=True)


for n, k in ((12288, 4096), (22016, 4096), (4096, 4096), (4096, 11008)):
    for bits in (4, 3):
        run(n, k, bits, 0.5)
