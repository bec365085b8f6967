#!/usr/bin/env python3
"""repeat one GEMM shape many times through the split-K and the single-pass entry points; report mismatching launches"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops
ops.LIB_GEMM_ROWS = 0        # time / check the fused unpack + MFMA kernel itself, _lib
from amq_amd.hqq_format import random_hqq
dev = torch.device("cuda:0")
lib = _lib.load()
for (m, n, k, bits) in ((64, 4096, 11008, 4), (64, 4096, 4096, 4), (100, 1280, 2048, 2), (512, 4096, 4096, 3), (4096, 5120, 5120, 4)):
    h = random_hqq(n, k, bits, seed=1).to(dev)
    qn, mn = ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), bits, n, k)
    x = torch.randn(m, k, device=dev).half()
    w = ops.dequantize(qn, mn, bits, 0, n, k)
    ref = (x.float() @ w.float().t())
    rms = ref.pow(2).mean().sqrt()
    for name in ("auto", "single"):
        bad = 0; first = None; worst = 0.0
        for it in range(60):
            y = torch.empty(m, n, dtype=torch.float16, device=dev)
            if name == "auto":
                ops.gemm(x, qn, mn, bits, 0, n, k, out=y)
            else:
                _lib.check(lib.amq_gemm_f16(bits, 0, _lib.ptr(x), _lib.ptr(qn), _lib.ptr(mn), None, _lib.ptr(y), m, n, k, 128, 0, 0, _lib.current_stream()))
            err = ((y.float() - ref).abs() / (1e-3 * ref.abs() + 1e-3 * rms)).max().item()
            worst = max(worst, err)
            if first is None:
                first = y.clone()
            elif not torch.equal(first, y):
                bad += 1
                if bad == 1:
                    d = (first != y).nonzero()
                    print("   first mismatch: %d elements, rows %s cols %s" % (d.shape[0], sorted(set(d[:, 0].tolist()))[:8], sorted(set(d[:, 1].tolist()))[:8]))
        print(f"{m}x{n}x{k} b{bits} {name:6s} splitk_ws={lib.amq_gemm_splitk_workspace_bytes(m, n, k)}: {bad}/59 launches differ from the first; worst err/bar {worst:.2f}", flush=True)
