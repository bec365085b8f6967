#!/usr/bin/env python3
"""usage: prof_attn.py B S heads kv_heads [iters] -- repeated prompt-attention launches for rocprofv3"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
B, S, nh, nkv = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dev = torch.device("cuda:0")
q = torch.randn(B * S, nh * 128, device=dev).half(); k = torch.randn(B * S, nkv * 128, device=dev).half(); v = torch.randn(B * S, nkv * 128, device=dev).half()
out = torch.empty_like(q)
for _ in range(iters):
    ops.attn_prefill(q, k, v, out, S, nh, nkv, batch=B)
torch.cuda.synchronize()
print("done")
