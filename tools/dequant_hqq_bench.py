#!/usr/bin/env python3
"""amq_dequantize_hqq_f16 (HQQ Format A -> fp16 W[N,K]) against HBM: us, GB/s of fp16 written and of all bytes moved.
Bit-exactness against the oracle is checked in tests/ (golden W_deq); here: against the native-layout dequantize kernel."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops
from amq_amd.hqq_format import random_hqq

dev = torch.device("cuda:0")
for n, k in ((4096, 4096), (11008, 4096), (8192, 28672)):
    for bits in (4, 3, 2):
        h = random_hqq(n, k, bits, seed=bits).to(dev)
        s, z = h.scale.reshape(-1).contiguous(), h.zero.reshape(-1).contiguous()
        w = ops.dequantize_hqq(h.W_q, s, z, bits, n, k)
        qn, mn = ops.repack_from_hqq(h.W_q, s, z, bits, n, k)
        assert torch.equal(w, ops.dequantize(qn, mn, bits, ops.MODE_HQQ, n, k))
        reps = 50
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.dequantize_hqq(h.W_q, s, z, bits, n, k)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        wr, rd = 2.0 * n * k, h.W_q.numel() * h.W_q.element_size() + 4.0 * n * k / 128
        print(json.dumps({"N": n, "K": k, "bits": bits, "us": round(us, 1), "write_GBps": round(wr / us / 1e3, 1),
                          "total_GBps": round((wr + rd) / us / 1e3, 1), "write_frac_of_8TBps": round(wr / us / 1e3 / 8000, 3)}), flush=True)
