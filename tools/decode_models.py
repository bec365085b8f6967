#!/usr/bin/env python3
"""decode tokens/s (graph replay, batch 1, 64-token prompt) for the model shapes of BASELINE.json's configs 2-5 on ONE GPU:
uniform 4-bit 7B, avg-3 7B / 13B / 70B (synthetic weights), plus configs[1] in the reference's own format: uniform 4-bit 7B whose
weights are IMPORTED from AWQ (FT_QuantLinear) buffers -- MODE_FMA arithmetic, w = fma(q, s, c) -- instead of native HQQ payloads.
usage: decode_models.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import arch
from amq_amd.llama import QuantLlama

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
cases = [("Llama-2-7b-hf", 4.0, True), ("Llama-2-7b-hf awq-import", 4.0, True), ("Llama-2-7b-hf", 3.0, False), ("Llama-2-13b-hf", 3.0, False),
         ("Llama-2-70b-hf", 3.0, False)]


def awq_import(m):
    """replace every linear of the runner by one imported from AWQ-format buffers (amq_repack_from_awq): same integers as a random
    4-bit layer, scales / scaled zeros as FT_QuantLinear stores them (fp16 [K/G, N]); the runner then decodes with MODE_FMA"""
    from amq_amd import ops
    gen = torch.Generator(device=dev).manual_seed(7)
    for blk in m.blocks:
        for name in m.cfg["linear"]:
            l = blk[name]
            n, k = l.N, l.K
            qweight = torch.randint(-2 ** 15, 2 ** 15 - 1, (n // 4, k), dtype=torch.int16, device=dev, generator=gen)
            s = ((0.75 + 0.5 * torch.rand(k // 128, n, device=dev, generator=gen)) * 0.5 / (k ** 0.5 * 4.6)).half()
            sz = (-(7.5 + torch.rand(k // 128, n, device=dev, generator=gen) - 0.5) * s.float()).half()
            l.qn, l.mn = ops.repack_from_awq(qweight, s, sz, n, k)
            l.mode = ops.fma_mode_for(l.mn, l.bits) if os.environ.get("DECODE_FMA1", "1") != "0" else ops.MODE_FMA

if os.environ.get("DECODE_MODELS"):
    cases = [c for c in cases if any(k in c[0] for k in os.environ["DECODE_MODELS"].split(","))]
for name, bits, uniform in cases:
    cfg = arch.MODEL_CONFIGS[name.split()[0]]
    if uniform:
        a, usage = arch.uniform_arch(cfg, int(bits)), bits + 0.25
    else:
        a, usage = arch.synthesize_arch(cfg, bits, seed=0, pinned=arch.PINNED_7B if "7b" in name else ())
    group = int(os.environ.get("DECODE_GROUP", "128"))          # 64 / 32: layers with two / four (scale, zero) pairs per tile row
    m = QuantLlama(cfg, a["linear"], device=dev, max_seq=64 + steps + 24, seed=0, group=group)
    if group != 128:
        name += f" group {group}"
        usage += 4.0 * (1.0 / group - 1.0 / 128) * 8
    if "awq-import" in name:
        awq_import(m)
    ids = torch.randint(0, m.vocab - 1, (64,), generator=torch.Generator().manual_seed(0)).to(dev)
    m.prefill(ids, use_graph=False)
    m.capture()
    for _ in range(16):
        m.decode_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.decode_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gb = m.linear_bytes_per_token() / 1e9
    print(f"{name} bits_usage {usage:.3f}: {1/dt:7.1f} tokens/s  {dt*1e3:.3f} ms/token  linears {gb:.2f} GB/token -> {gb/dt/1e3:.2f} TB/s", flush=True)
    del m
    torch.cuda.empty_cache()
