#!/usr/bin/env python3
"""decode throughput against the batch (graph replay, 64-token prompts, Llama-2-7B avg-3 synthetic weights): sequences per step
share one pass over the weights.  usage: decode_batch.py [batches, comma separated] [steps]
(NORM_SUMS=0: the 5 .. 8-row steps with one rmsnorm launch per norm instead of the partial-sum RMSNorm, A/B)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import arch, ops
from amq_amd.llama import QuantLlama
if os.environ.get("GEMV_WAVES") or os.environ.get("GEMV_DEPTH") or os.environ.get("GEMV_DOT"):
    ops.DEFAULT_GEMV_OPTS = ops.GemvOpts(waves=int(os.environ.get("GEMV_WAVES", "0")), depth=int(os.environ.get("GEMV_DEPTH", "0")),
                                         dot=int(os.environ.get("GEMV_DOT", "0")))   # A/B: waves per workgroup, ring depth, v_dot2 body at one row

if os.environ.get("NORM_SUMS") == "0":
    QuantLlama.NORM_SUMS = False
batches = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,8".split(","))]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
name = os.environ.get("SWEEP_MODEL", "Llama-2-7b-hf")
dev = torch.device("cuda:0")
cfg = arch.MODEL_CONFIGS[name]
a, usage = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=arch.PINNED_7B if "7b" in name else ())
for B in batches:
    m = QuantLlama(cfg, a["linear"], device=dev, max_seq=64 + steps + 24, seed=0, batch=B)
    ids = torch.randint(0, m.vocab - 1, (B, 64), generator=torch.Generator().manual_seed(0)).to(dev)
    m.prefill(ids if B > 1 else ids[0], use_graph=False)
    m.capture()
    for _ in range(8):
        m.decode_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.decode_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m.check()
    print(f"{name} batch {B}: {dt*1e3:.3f} ms/step  {1/dt:7.1f} steps/s  {B/dt:8.1f} tokens/s aggregate", flush=True)
    del m
    torch.cuda.empty_cache()
