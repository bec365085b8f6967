#!/usr/bin/env python3
"""64-token prompt pass of the 7B avg-3-bit runner (hipGraph replay, HIP events): the harness's GeMM / TTFT shape.  usage: prompt64_time.py [rows]
A/B (environment, this tool only): FUSE_DOWN_NORM=0 -- down_proj's split-K reduce and the next block's RMSNorm as two launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import arch
from amq_amd.llama import QuantLlama
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if "FUSE_DOWN_NORM" in os.environ:
    QuantLlama.FUSE_DOWN_NORM = os.environ["FUSE_DOWN_NORM"] != "0"
dev = torch.device("cuda:0")
cfg = arch.MODEL_CONFIGS["Llama-2-7b-hf"]
a, usage = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=arch.PINNED_7B)
m = QuantLlama(cfg, a["linear"], device=dev, max_seq=rows + 64, seed=0)
ids = torch.randint(0, m.vocab - 1, (rows,), generator=torch.Generator().manual_seed(0)).to(dev)
for _ in range(3):
    m.reset(); m.prefill(ids)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    m.reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); m.prefill(ids); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"{rows}-row prompt pass (FUSE_DOWN_NORM={int(QuantLlama.FUSE_DOWN_NORM)}): median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f} ms")
