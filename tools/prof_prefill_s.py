#!/usr/bin/env python3
"""usage: prof_prefill_s.py S [iters] -- eager prompt passes of S rows on the 7B avg-3 bench model, for rocprofv3 --kernel-trace --stats"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
S = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m, a, usage = bench.build_model(torch.device('cuda:0'), seed=0, max_seq=S + 8)
ids = torch.randint(0, m.vocab - 1, (S,), generator=torch.Generator().manual_seed(0)).to(m.dev)
for _ in range(iters):
    m.prefill(ids, use_graph=False)
torch.cuda.synchronize()
print("done")
