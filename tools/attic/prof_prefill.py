#!/usr/bin/env python3
"""eager prefill launches of the bench model for rocprofv3 --kernel-trace --stats.  usage: prof_prefill.py [S]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m, a, usage = bench.build_model(torch.device("cuda:0"), seed=0, max_seq=S + 8)
ids = torch.randint(0, m.vocab - 1, (S,), generator=torch.Generator().manual_seed(0)).to(m.dev)
for _ in range(3):
    m.prefill(ids, use_graph=False)
torch.cuda.synchronize()
print("done")
