#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its workload.

metric   decode tokens/s (whole job), Llama-2-7B shapes, AMQ mixed 2/3/4-bit
         per-layer config at avg 3.0 bits (configs[2] of BASELINE.json: the
         config the metric is quoted on), batch 1, one decode stream per GPU.
step     one decode token through the whole model: 32 x (fused-RMSNorm q/k/v
         GEMV, RoPE+KV+attention, o_proj GEMV+residual, fused-RMSNorm gate/up
         GEMV, SiLU*mul down GEMV+residual) + final norm + fp16 lm_head + argmax,
         replayed from a hipGraph.  A 64-token prefill runs un-timed first
         (reference GeMV mode: amq/utils/speed.py:50-89).
data     synthetic: random native 2/3/4-bit payloads of the real layer shapes
         and a synthesized arch (SearchSpace.sample recipe, seed 0; no searched
         .stats file or checkpoint ships with the reference, and there is no network).
roofline the dominant kernel is the weight-streaming GEMV (amq::gemv_kernel);
         `achieved` = algorithmic bytes of all its launches in one token
         (packed weights + fp16 scale/zero + x + y) divided by
         the replay time of a hipGraph holding exactly those launches, timed with
         HIP events on the launch stream -- i.e. it INCLUDES the ~1.3-1.5 us
         device-side boundary between dependent kernels (conservative; the
         rocprofv3 kernel-only averages are in profiles/).
cpu_baseline  the reference's CPU path (nn.Linear on dequantized weights) ported
         to torch CPU ops, timed on the host cores on a bounded sample (oracle/cpu_baseline.py).

N > 1: independent replicas (one decode stream per GPU, no data-path
collective); RCCL is used only for the start barrier and the max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = "Llama-2-7b-hf"
TARGET_BITS = 3.0
PROMPT = 64


def build_model(device, seed=0, max_seq=1024):
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = arch.MODEL_CONFIGS[MODEL]
    a, usage = arch.synthesize_arch(cfg, TARGET_BITS, seed=0, pinned=arch.PINNED_7B)
    m = QuantLlama(cfg, a["linear"], device=device, max_seq=max_seq, seed=seed)
    return m, a, usage


def gemv_roofline(m, reps=20):
    """time a graph holding only the GEMV launches of one token (4 per block)"""
    from amq_amd import ops
    EPS = 1e-5
    dev = m.dev
    H, I = m.H, m.I

    def launches():
        for blk in m.blocks:
            ops.gemv_grouped(m.x, [blk["self_attn.q_proj"].seg(m.q), blk["self_attn.k_proj"].seg(m.k),
                                   blk["self_attn.v_proj"].seg(m.v)], H, prologue=ops.PRO_RMSNORM, gamma=blk["ln1"], eps=EPS)
            ops.gemv_grouped(m.att, [blk["self_attn.o_proj"].seg(m.q)], H)
            ops.gemv_grouped(m.x, [blk["mlp.gate_proj"].seg(m.gate), blk["mlp.up_proj"].seg(m.up)], H,
                             prologue=ops.PRO_RMSNORM, gamma=blk["ln2"], eps=EPS)
            ops.gemv_grouped(m.gate, [blk["mlp.down_proj"].seg(m.q)], I, prologue=ops.PRO_SILU_MUL, x2=m.up)

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        launches()
        side.synchronize()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            launches()
    torch.cuda.current_stream(dev).wait_stream(side)
    g.replay()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize(dev)
    span_s = e0.elapsed_time(e1) * 1e-3 / reps
    n_launch = 4 * m.nb
    # algorithmic bytes: packed weights + fp16 scale/zero (the native buffers are exactly that) + x + y per launch
    wbytes = m.linear_bytes_per_token()
    xy = m.nb * (2 * H + 2 * (H + 2 * m.kvd)) + m.nb * (2 * H + 2 * H) + m.nb * (2 * H + 2 * 2 * I) + m.nb * (2 * 2 * I + 2 * H)
    alg = wbytes + xy
    return {"bytes_per_launch": alg / n_launch, "us_per_launch": span_s / n_launch * 1e6, "launches_per_token": n_launch,
            "gbps": alg / span_s / 1e9}


def cpu_baseline(seed=0):
    from amq_amd import arch
    from amq_amd.hqq_format import random_hqq
    from oracle import cpu_baseline as cb
    cfg = arch.MODEL_CONFIGS[MODEL]
    bits_cycle = [4, 3, 2, 3, 3, 2, 4]          # one block at ~avg 3 bits
    layers = []
    for name, bits in zip(cfg["linear"], bits_cycle):
        n, k = cfg["linear_shape"][name]
        h = random_hqq(n, k, bits, seed=seed + len(layers))
        layers.append({"W_q": h.W_q, "scale": h.scale, "zero": h.zero, "nbits": bits, "shape": (n, k)})
    lm_head = torch.randn(cfg["vocab_size"], cfg["hidden_size"]).to(torch.float16)
    return cb.time_decode_linears(layers, cfg["n_block"], tokens=8, extra_dense=lm_head)


def load_traffic():
    """HBM bytes per GEMV launch from the committed rocprofv3 PMC pass (profiles/), or None"""
    p = os.path.join(ROOT, "profiles", "r01_gemv_pmc.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    from amq_amd.replicas import Replicas
    rep = Replicas(backend="nccl", device=dev)        # nccl == RCCL on ROCm; no-op for one process
    rank, n_gpus = rep.rank, rep.world

    max_seq = PROMPT + args.warmup + args.steps + 8
    m, a, usage = build_model(dev, seed=rank, max_seq=max_seq)
    ids = torch.randint(0, m.vocab - 1, (PROMPT,), generator=torch.Generator().manual_seed(rank)).to(dev)
    m.prefill(ids)                                  # un-timed (GeMV-mode protocol)
    m.capture()
    for _ in range(args.warmup):
        m.decode_step()
    elapsed = rep.timed(m.decode_step, args.steps, sync=lambda: torch.cuda.synchronize(dev))
    ok = bool(torch.isfinite(m.logits.float()).all().item())

    if rank == 0:
        roof = gemv_roofline(m)
        peak = 8000.0
        out = {
            "metric": "decode tokens/s, Llama-2-7B AMQ mixed 2/3/4-bit avg-3-bit, batch 1",
            "value": n_gpus * args.steps / elapsed,
            "unit": "tokens/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "Llama-2-7B shapes, synthesized avg-3.0-bit per-layer arch (bits_usage %.3f), "
                                   "batch 1 decode after a 64-token prefill, one stream per GPU" % usage,
                       "parallelism": "replicas x%d" % n_gpus, "prompt": PROMPT, "group_size": 128},
            "roofline": {"bound": "hbm", "achieved": roof["gbps"], "peak": peak, "unit": "GB/s",
                         "frac": roof["gbps"] / peak, "traffic": load_traffic(),
                         "kernel": "amq::gemv_kernel (grouped 2/3/4-bit weight-streaming GEMV)",
                         "bytes_per_launch": roof["bytes_per_launch"], "us_per_launch": roof["us_per_launch"],
                         "launches_per_token": roof["launches_per_token"]},
            "finite_logits": ok,
            "linear_gb_per_token": m.linear_bytes_per_token() / 1e9,
            "model_gbps": m.total_bytes_per_token(PROMPT + args.warmup + args.steps // 2) / (elapsed / args.steps) / 1e9,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            out["cpu_baseline"] = {"value": cb["tokens_per_s_predequantized"], "unit": "tokens/s", "cores": cb["cores"],
                                   "kind": "port", "sample": cb["sample"],
                                   "dequant_every_call_tokens_per_s": cb["tokens_per_s_dequant_every_call"]}
        print(json.dumps(out), flush=True)
    rep.close()


if __name__ == "__main__":
    main()
