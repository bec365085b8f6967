#!/usr/bin/env python3
"""The decode attention launch alone at a long cache: L layers' worth of distinct K / V caches (HBM-cold: L x 2 x heads x T x 256 B >> the 256 MiB
Infinity Cache), one launch per layer replayed from a hipGraph, HIP events.  usage: attn_decode_bench.py [T] [n_splits,...] [heads] [kv_heads] [layers]
prints us per launch and the K + V bytes it read per second."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
splits = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0".split(","))]
nh = int(sys.argv[3]) if len(sys.argv) > 3 else 32
nkv = int(sys.argv[4]) if len(sys.argv) > 4 else nh
L = int(sys.argv[5]) if len(sys.argv) > 5 else 32
if os.environ.get("GQA_KEYS"):
    ops.ATTN_GQA_KEYS = int(os.environ["GQA_KEYS"])        # A/B: keys per workgroup of the grouped-query kernel (a multiple of 128)
dev = torch.device("cuda:0")
max_seq = int(os.environ.get("MAX_SEQ", T + 80))      # (MAX_SEQ: the cache's size, e.g. a power of two just above T)
g = torch.Generator(device=dev).manual_seed(0)
kc = [torch.randn(1, nkv, max_seq, 128, device=dev, generator=g).half() for _ in range(L)]
vc = [torch.randn(1, nkv, max_seq, 128, device=dev, generator=g).half() for _ in range(L)]
q = torch.randn(1, nh * 128, device=dev, generator=g).half()
k = torch.randn(1, nkv * 128, device=dev, generator=g).half()
v = torch.randn(1, nkv * 128, device=dev, generator=g).half()
out = torch.empty(1, nh * 128, device=dev, dtype=torch.float16)
pos = torch.tensor([T], dtype=torch.int32, device=dev)
table = ops.rope_table(max_seq, 10000.0, dev)
for ns in splits:
    def launches():
        for l in range(L):
            ops.attn_decode(q, k, v, kc[l], vc[l], out, pos, nh, nkv, table=table, n_splits=ns)
    launches()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            launches()
    torch.cuda.current_stream(dev).wait_stream(side)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps / L
    byts = 2 * nkv * (T + 1) * 256
    eff = ns if ns else ops.attn_decode_splits(max_seq, nh, 1, nkv)
    print(f"T {T} heads {nh}/{nkv} n_splits {eff:3d}: {us:7.2f} us per launch   {byts / us / 1e3:7.1f} GB/s of K+V ({byts / 1e6:.1f} MB)", flush=True)
