#!/usr/bin/env python3
"""Does the wave-specialised GEMM (amq_gemm_ws.hip, route 5) earn its place in the product library?  For mid-size launches (below the dequantize-once
threshold) times the ring kernel with 256- and 128-row tiles, the wave-specialised kernel and AUTO's choice; prints, per shape, AUTO's route and what the
wave-specialised kernel gains over the better ring tile (it stays only where that is >= 3 %).
usage: ws_vs_ring.py [out.txt]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops
from amq_amd.llama import _synthetic_linear
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
lines = []


def t_route(x, l, n, k, y, r):
    ts = []
    for i in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(x, l.qn, l.mn, l.bits, l.mode, n, k, out=y, route=r); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts[1:])[2]


hdr = f"{'M':>6s} {'N':>6s} {'K':>6s} bits  ring256  ring128       ws     auto   ws/best-ring   (ms; TF of the best)"
print(hdr); lines.append(hdr)
wins = 0
for (n, k) in ((4096, 4096), (11008, 4096), (5120, 5120), (13824, 5120), (8192, 8192), (1024, 8192), (28672, 8192), (4096, 11008)):
    for bits in (3,):
        l = _synthetic_linear(n, k, bits, gen, dev)
        for m in (768, 1024, 1536, 2048, 3072, 4096, 5120):
            x = (torch.randn(m, k, device=dev, generator=gen) * 0.5).half()
            y = torch.empty(m, n, device=dev, dtype=torch.float16)
            t = {r: t_route(x, l, n, k, y, r) for r in (ops.GEMM_RING, ops.GEMM_RING128, ops.GEMM_WS, ops.GEMM_AUTO)}
            best_ring = min(t[ops.GEMM_RING], t[ops.GEMM_RING128])
            auto_is_ws = abs(t[ops.GEMM_AUTO] - t[ops.GEMM_WS]) < 0.02 * t[ops.GEMM_WS] and abs(t[ops.GEMM_AUTO] - best_ring) > 0.02 * best_ring
            gain = best_ring / t[ops.GEMM_WS]
            wins += gain >= 1.03
            s = (f"{m:6d} {n:6d} {k:6d} {bits:4d} {t[ops.GEMM_RING]:8.3f} {t[ops.GEMM_RING128]:8.3f} {t[ops.GEMM_WS]:8.3f} {t[ops.GEMM_AUTO]:8.3f}   {gain:6.3f}"
                 f"{'  <- auto = ws' if auto_is_ws else ''}   {2.0 * m * n * k / min(t.values()) / 1e9:7.0f} TF")
            print(s, flush=True); lines.append(s)
s = f"shapes where the wave-specialised kernel beats the better ring tile by >= 3 %: {wins}"
print(s); lines.append(s)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
