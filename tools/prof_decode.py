#!/usr/bin/env python3
"""Eager (no hipGraph) decode steps of the bench workload, for rocprofv3 --pmc passes
(counter collection segfaults on graph replays with ROCm 7.2).  usage: prof_decode.py [tokens] [gs]   (gs: the opt-in AMQ_MATH_GROUPSCALE arithmetic)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if len(sys.argv) > 2 and sys.argv[2] == "gs":
    from amq_amd import ops
    ops.DEFAULT_GEMV_OPTS = ops.GemvOpts(math=ops.MATH_GROUPSCALE)
m, a, usage = bench.build_model(torch.device("cuda:0"), seed=0, max_seq=256)
ids = torch.randint(0, m.vocab - 1, (64,), generator=torch.Generator().manual_seed(0)).to(m.dev)
m.prefill(ids)
for _ in range(tokens):
    m.decode_step(use_graph=False)
torch.cuda.synchronize()
print("done", tokens, "tokens; linear bytes/token", m.linear_bytes_per_token())
