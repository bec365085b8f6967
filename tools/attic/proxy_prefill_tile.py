import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
dev = torch.device("cuda:0")
for T in (2048, 8192, 32768):
    nh = 8
    S = 64
    L = 16
    kcs = [torch.randn(1, nh, T + 64, 128, device=dev).half() for _ in range(L)]
    vcs = [torch.randn(1, nh, T + 64, 128, device=dev).half() for _ in range(L)]
    q = torch.randn(S, nh * 128, device=dev).half()
    out = torch.empty_like(q)
    def run():
        for l in range(L):
            ops.attn_prefill(q, kcs[l], vcs[l], out, S, nh, nh, batch=1, pos0=T - S, kv_cache=True)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5 / L
    print(f"T {T}: {us:.1f} us per launch, {us / (T / 64):.3f} us per 64-key tile (one workgroup per head walking all keys)")
