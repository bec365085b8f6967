#!/usr/bin/env python3
"""Does splitting the decode attention over several workgroups per head pay at the SHORT contexts of the headline benchmark
(64 .. 340 cached keys)?  Run against a variant built with a smaller minimum chunk:
    make -C amq_amd/csrc tuvariant TU=amq_decode TAG=chunk64 EXTRA=-DAMQ_ATT_MIN_CHUNK=64
    python tools/with_variant.py chunk64 tools/attic/exp_attn_split_short.py
Times the bench's decode loop with 1 (product), 2, 3 and 4 workgroups per head."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from amq_amd import ops

dev = torch.device("cuda:0")
steps, warm = 256, 16
for splits in (1, 2, 3, 4, 1):
    max_seq = bench.PROMPT + warm + steps + 8
    if splits == 1:
        ops.ATTN_SPLIT_FROM, ops.ATTN_CHUNK = 512, 384
    else:
        ops.ATTN_SPLIT_FROM, ops.ATTN_CHUNK = 0, (max_seq + splits - 1) // splits
    m, _, _ = bench.build_model(dev, seed=0, max_seq=max_seq)
    ids = torch.randint(0, m.vocab - 1, (bench.PROMPT,), generator=torch.Generator().manual_seed(0)).to(dev)
    m.prefill(ids)
    m.capture()
    for _ in range(warm):
        m.decode_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.decode_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m.check()
    print(f"workgroups per head {splits} (chunk {ops.ATTN_CHUNK}): {steps / dt:7.1f} tokens/s   token {int(m.token.item())}", flush=True)
    del m
    torch.cuda.empty_cache()
