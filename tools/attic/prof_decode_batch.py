#!/usr/bin/env python3
"""usage: prof_decode_batch.py B [steps] -- eager batched decode steps of the 7B avg-3 bench model for rocprofv3 --kernel-trace --stats"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import arch
from amq_amd.llama import QuantLlama
B = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = arch.MODEL_CONFIGS["Llama-2-7b-hf"]
a, _ = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=arch.PINNED_7B)
m = QuantLlama(cfg, a["linear"], device="cuda:0", max_seq=64 + steps + 24, seed=0, batch=B)
ids = torch.randint(0, m.vocab - 1, (B, 64), generator=torch.Generator().manual_seed(0)).to(m.dev)
m.prefill(ids if B > 1 else ids[0], use_graph=False)
for _ in range(steps):
    m.decode_step(use_graph=False)
torch.cuda.synchronize()
print("done")
