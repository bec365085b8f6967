#!/usr/bin/env python3
"""decode tokens/s against the context length (graph replay, batch 1, Llama-2-7B avg-3 synthetic weights): the attention
kernel's share of a token grows with the cached keys.  usage: decode_context_sweep.py [contexts, comma separated] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import arch, ops
from amq_amd.llama import QuantLlama
if os.environ.get("ATTN_CHUNK"):
    ops.ATTN_CHUNK = int(os.environ["ATTN_CHUNK"])          # A/B: keys per workgroup at a full cache

ctxs = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "64,512,1024,2048,4000".split(","))]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
name = os.environ.get("SWEEP_MODEL", "Llama-2-7b-hf")
dev = torch.device("cuda:0")
cfg = arch.MODEL_CONFIGS[name]
a, usage = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=arch.PINNED_7B if "7b" in name else ())
m = QuantLlama(cfg, a["linear"], device=dev, max_seq=max(ctxs) + steps + 40, seed=0)
for S in ctxs:
    m.reset()
    ids = torch.randint(0, m.vocab - 1, (S,), generator=torch.Generator().manual_seed(0)).to(dev)
    m.prefill(ids, use_graph=False)
    m.capture()
    for _ in range(8):
        m.decode_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.decode_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name} context {S:5d}: {1/dt:7.1f} tokens/s  {dt*1e3:.3f} ms/token", flush=True)
