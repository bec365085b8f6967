#!/usr/bin/env python3
"""prefill wall time (hipGraph replay, median of 5) for 13B / 70B shapes at 64 / 256 / 2048 prompt rows, with the fused pass
checked against the framework-glue pass on the last-token logits.  usage: prefill_models.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import arch
from amq_amd.llama import QuantLlama

dev = torch.device("cuda:0")
for name in ("Llama-2-13b-hf", "Llama-2-70b-hf"):
    cfg = arch.MODEL_CONFIGS[name]
    a, usage = arch.synthesize_arch(cfg, 3.0, seed=0)
    m = QuantLlama(cfg, a["linear"], device=dev, max_seq=2100, seed=0)
    for S in (64, 256, 2048):
        ids = torch.randint(0, m.vocab - 1, (S,), generator=torch.Generator().manual_seed(0)).to(dev)
        ref = m._prefill_unfused(ids).float().clone()
        got = m.prefill(ids, use_graph=False).float().clone()
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        for _ in range(2):
            m.prefill(ids)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); m.prefill(ids); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{name} S={S}: {ts[2]*1e3:.2f} ms ({S/ts[2]:.0f} tokens/s)  fused-vs-glue logits distance {err:.2e}", flush=True)
    del m
    torch.cuda.empty_cache()
