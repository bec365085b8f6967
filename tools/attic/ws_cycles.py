#!/usr/bin/env python3
"""Shader cycles per half-tile (64 k) of the wave-specialised GEMM's consumer waves, from a -DAMQ_WS_CYCLES build:
    python tools/with_variant.py <tag> tools/attic/ws_cycles.py
64 MFMAs per half-tile and consumer = 1024 cycles at a full matrix pipe.  Cycle counts do not depend on the data or on the
clock the chip holds (wall-clock A/B of timing ablations does: they change the operand values, hence power, hence clock)."""
import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import _lib, ops
from amq_amd.llama import _synthetic_linear
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
lib = _lib.load()
M = 32768
for (n, k) in ((13824, 5120), (5120, 13824)):
    for bits in (4, 3, 2):
        l = _synthetic_linear(n, k, bits, gen, dev)
        x = (torch.randn(M, k, device=dev, generator=gen) * 0.5).half()
        y = torch.empty(M, n, device=dev, dtype=torch.float16)
        ts = []
        for i in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm(x, l.qn, l.mn, bits, l.mode, n, k, out=y, route=ops.GEMM_WS); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        T = min(8192, (M // 256) * ((n + 127) // 128))
        buf = (ctypes.c_ulonglong * T)()
        rc = lib.amq_debug_ws_cycles(buf, T)
        assert rc == 0, rc
        c = sorted(buf)
        pb = (ctypes.c_ulonglong * (4 * T))()
        prod = None
        if hasattr(lib, "amq_debug_ws_pcycles") and lib.amq_debug_ws_pcycles(pb, T) == 0:
            prod = [round(sorted(pb[4 * i + j] for i in range(T))[T // 2] / (2 * (k // 128)), 1) for j in range(3)]
        nh = 2 * (k // 128)
        t = sorted(ts[1:])[1]
        print(json.dumps({"N": n, "K": k, "bits": bits, "TF": round(2.0 * M * n * k / t / 1e9, 1), "cycles_per_half_median": round(c[T // 2] / nh, 1),
                          "p10": round(c[T // 10] / nh, 1), "p90": round(c[9 * T // 10] / nh, 1), "pipe_frac": round(1024.0 * nh / c[T // 2], 3),
                          "producer_issue_stores|x_wait|barrier_wait": prod}), flush=True)
