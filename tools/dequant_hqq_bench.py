#!/usr/bin/env python3
"""amq_dequantize_hqq_f16 (HQQ Format A -> fp16 W[N,K]) against HBM: us, GB/s of fp16 written and of all bytes moved.
Warm-up launches first, then `reps` timed launches rotating over `copies` distinct input / output buffer sets (so that no
launch finds its lines in the Infinity Cache).  Bit-exactness against the oracle is checked in tests/ (golden W_deq); here:
against the native-layout dequantize kernel.  Used by bench.py (`dequant_hqq` rows of the result line)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops
from amq_amd.hqq_format import random_hqq

SHAPES = ((4096, 4096), (11008, 4096), (8192, 28672))


def measure(dev, shapes=SHAPES, reps=48, budget_bytes=768 << 20):
    rows = []
    for n, k in shapes:
        for bits in (4, 3, 2):
            h = random_hqq(n, k, bits, seed=bits).to(dev)
            s, z = h.scale.reshape(-1).contiguous(), h.zero.reshape(-1).contiguous()
            w = ops.dequantize_hqq(h.W_q, s, z, bits, n, k)
            qn, mn = ops.repack_from_hqq(h.W_q, s, z, bits, n, k)
            assert torch.equal(w, ops.dequantize(qn, mn, bits, ops.MODE_HQQ, n, k))
            del w, qn, mn
            per = 2 * n * k + h.W_q.numel() * h.W_q.element_size()
            copies = max(2, min(16, budget_bytes // per))
            wq = [h.W_q.clone() for _ in range(copies)]
            lib = ops._lib.load()
            outs = [torch.empty(n, k, dtype=torch.float16, device=dev) for _ in range(copies)]

            def call(i):
                ops._lib.check(lib.amq_dequantize_hqq_f16(bits, ops._lib.ptr(wq[i]), ops._lib.ptr(s), ops._lib.ptr(z), n, k, 128,
                                                          ops._lib.ptr(outs[i]), ops._lib.current_stream()))
            for i in range(copies):
                call(i)                                 # warm-up (first touch of every buffer)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(reps):
                call(r % copies)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            wr, rd = 2.0 * n * k, h.W_q.numel() * h.W_q.element_size() + 4.0 * n * k / 128
            rows.append({"N": n, "K": k, "bits": bits, "us": round(us, 1), "write_GBps": round(wr / us / 1e3, 1),
                         "total_GBps": round((wr + rd) / us / 1e3, 1), "write_frac_of_8TBps": round(wr / us / 1e3 / 8000, 3),
                         "buffer_sets": copies})
            del wq, outs
            torch.cuda.empty_cache()
    return rows


if __name__ == "__main__":
    for row in measure(torch.device("cuda:0")):
        print(json.dumps(row), flush=True)
