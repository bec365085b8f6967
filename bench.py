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
         rocprofv3 kernel-only averages are in profiles/).  `gemv_layers` is the second
         half of the metric: the same measurement per (shape, bit-width) of the 7B
         linears over cold weight copies.
cpu_baseline  the reference's CPU path (nn.Linear on dequantized weights) ported
         to torch CPU ops, timed on the host cores on a bounded sample with the
         thread count swept (oracle/cpu_baseline.py).

--gpus N (N > 1) without a launcher (no WORLD_SIZE in the environment): this process
spawns N fresh children, one rank per GPU, BEFORE touching the GPU itself; under
`python -m torch.distributed.run` the ranks already exist and are used as they are.
Ranks are independent replicas (one decode stream per GPU, no data-path collective);
RCCL is used for the start/stop barrier, the max-over-ranks time and the per-rank rates.

--config 5: BASELINE.json configs[4] -- Llama-2-70B shapes, avg-3-bit arch, one independent decode stream per rank (N = 1: one
replica on one GPU, the case the driver can record on a one-GPU box; N = 8 is configs[4] itself): same line schema, roofline on the
70B GEMV launches.  The 7B-specific extras of the default line (per-layer table, MFMA roofline, CPU baseline) are not repeated.

--config 4: BASELINE.json configs[3] instead -- Llama-2-13B avg-3-bit, 16 x 2048 prompt
rows in one batched prompt pass (the harness' GeMM mode, amq/utils/speed.py:61-71); a step
is one pass, the roofline object is the MFMA one (linears' flop / time of the linears).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = "Llama-2-7b-hf"
TARGET_BITS = 3.0
PROMPT = 64
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 measured with a plain copy)
MFMA_PEAK_TFLOPS = 2500.0       # dense fp16/bf16


def build_model(device, seed=0, max_seq=1024, model=MODEL, pinned=None, batch=1, engine=None):
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = arch.MODEL_CONFIGS[model]
    a, usage = arch.synthesize_arch(cfg, TARGET_BITS, seed=0, pinned=arch.PINNED_7B if pinned is None else pinned)
    m = QuantLlama(cfg, a["linear"], device=device, max_seq=max_seq, seed=seed, batch=batch, engine=engine)
    return m, a, usage


def _decode_rate(dev, m, prompt, warm, steps, recapture=False):
    """decode steps per second of runner ``m`` after a ``prompt``-token prefill (None: keep its context): capture, ``warm`` replays, ``steps`` timed"""
    if prompt is not None:
        ids = torch.randint(0, m.vocab - 1, (m.B, prompt) if m.B > 1 else (prompt,), generator=torch.Generator().manual_seed(0)).to(dev)
        m.prefill(ids, use_graph=False)
    if recapture:
        m.graph = None
    m.capture()
    for _ in range(warm):
        m.decode_step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.decode_step()
    torch.cuda.synchronize(dev)
    rate = steps / (time.perf_counter() - t0)
    m.check()
    return rate


class _gemv_math:
    """run a block with the GEMV's opt-in arithmetic as the tools' default option (ops.DEFAULT_GEMV_OPTS; restored on exit)"""

    def __init__(self, math):
        self.math = math

    def __enter__(self):
        from amq_amd import ops
        self.old = ops.DEFAULT_GEMV_OPTS
        ops.DEFAULT_GEMV_OPTS = ops.GemvOpts(math=self.math)

    def __exit__(self, *exc):
        from amq_amd import ops
        ops.DEFAULT_GEMV_OPTS = self.old


def beyond_the_metric(dev):
    """Figures next to the headline (NOT part of `value`; rank 0 at N = 1 only), each a short graph-replayed run whose failure is reported as a
    string, never raised: the headline workload over a long cache (2048 cached keys: the attention step split over several workgroups per head),
    with 8 sequences decoded together (one pass over the weights per step), with its weights in the reference kernels' arithmetic, and under
    the GEMV's opt-in arithmetic; one replica of BASELINE.json configs[4] (Llama-2-70B) and one prompt pass of configs[3] (Llama-2-13B, 16 x 2048)."""
    from amq_amd import ops
    out = {}

    def leg(key, fn):
        try:
            out[key] = fn()
        except Exception as e:      # noqa: BLE001
            out[key] = "failed: %r" % (e,)
        torch.cuda.empty_cache()

    def long_cache():
        m, _, _ = build_model(dev, seed=0, max_seq=2048 + 64 + 16)
        return round(_decode_rate(dev, m, 2048, 8, 48), 1)

    def eight_sequences():
        m, _, _ = build_model(dev, seed=0, max_seq=PROMPT + 96, batch=8)
        return round(8 * _decode_rate(dev, m, PROMPT, 8, 64), 1)

    def reference_format():
        # the same model with its weights in the REFERENCE's kernel arithmetic, w = fma(q, s, c) with c = -(z s) (one rounding: what its GPTQ / AWQ
        # cache files decode to, auto_gptq_kernel.cu:206, gemv_cuda.cu:151), as a swapped model loaded from those files runs: AMQ_MODE_FMA1, one packed
        # fma per weight pair in the GEMV kernel (bit-identical to AMQ_MODE_FMA; HISTORY.md 7 item 5).  NOT `value`: the headline is the HQQ arithmetic
        # (two roundings) that the parity gate's W_deq is
        m, _, _ = build_model(dev, seed=0, max_seq=PROMPT + 96)
        modes = set()
        for blk in m.blocks:
            for name in m.cfg["linear"]:
                l = blk[name]
                mt = l.mn.view(-1, 2)
                mt[:, 1] = (-(mt[:, 1].float() * mt[:, 0].float())).to(torch.float16)
                l.mode = ops.fma_mode_for(l.mn, l.bits)
                modes.add(l.mode)
        rate = _decode_rate(dev, m, PROMPT, 8, 64)
        assert bool(torch.isfinite(m.logits.float()).all().item())
        out["reference_format_modes"] = sorted(modes)
        return round(rate, 1)

    def groupscale():
        # the headline workload under the opt-in GROUPSCALE arithmetic (include/amq_hip.h AMQ_MATH_GROUPSCALE: the first fp16 rounding per weight exact,
        # the scale applied per 128-group in fp32 after the MFMAs; 3.2e-4 rms(y) from the oracle, up to 0.93 of the parity bar -- and past it behind a
        # bias add: why it is not the default)
        with _gemv_math(ops.MATH_GROUPSCALE):
            m, _, _ = build_model(dev, seed=0, max_seq=PROMPT + 96)
            return round(_decode_rate(dev, m, PROMPT, 8, 64), 1)

    def llama70b():
        # BASELINE.json configs[4], one replica (what `--config 5 --gpus 1` prints as its own line): Llama-2-70B shapes, avg-3-bit arch, batch-1 decode
        m, _, usage = build_model(dev, seed=0, max_seq=PROMPT + 72, model="Llama-2-70b-hf", pinned=())
        tps = _decode_rate(dev, m, PROMPT, 8, 24)
        roof = gemv_roofline(m, reps=5)
        with _gemv_math(ops.MATH_GROUPSCALE):           # (the same replica under the opt-in arithmetic: a fresh capture over the same buffers)
            tps_gs = _decode_rate(dev, m, None, 4, 16, recapture=True)
        return {"tokens_per_s": round(tps, 2), "roofline_frac": round(roof["gbps"] / HBM_PEAK_GBPS, 3),
                "tokens_per_s_groupscale_math": round(tps_gs, 2), "us_per_launch": round(roof["us_per_launch"], 2),
                "gbps": round(roof["gbps"], 1), "linear_gb_per_token": round(m.linear_bytes_per_token() / 1e9, 2), "bits_usage": round(usage, 3),
                "steps": 24, "warmup": 8, "config": "BASELINE.json configs[4], one of its 8 independent streams"}

    def family(model):
        # the reference's other model families (README.md:90-92; amq/configs/llama.json:82+, mistral.json, qwen2.json) through the same runner: avg-3-bit
        # arch, batch-1 decode after a 64-token prompt.  Their vocabularies make the fp16 lm_head a fifth of a token's bytes (128,256 x 4096 x 2 B =
        # 1.05 GB; 152,064 x 3584 x 2 B = 1.09 GB): reported, and counted in whole_step_gbps
        def run():
            m, _, usage = build_model(dev, seed=0, max_seq=PROMPT + 96, model=model, pinned=())
            tps = _decode_rate(dev, m, PROMPT, 8, 64)
            roof = gemv_roofline(m, reps=10)
            step_bytes = m.total_bytes_per_token(PROMPT + 40)
            long_ctx = None
            if m.nh != m.nkv:       # grouped-query heads: the decode attention over a long cache (attn_decode_gqa_kernel: one workgroup per kv head and chunk)
                del m
                torch.cuda.empty_cache()
                m, _, _ = build_model(dev, seed=0, max_seq=8192, model=model, pinned=())
                long_ctx = round(_decode_rate(dev, m, 8000, 8, 48), 1)
            return {"tokens_per_s": round(tps, 1), "tokens_per_s_at_8000_cached_keys": long_ctx, "roofline_frac": round(roof["gbps"] / HBM_PEAK_GBPS, 3), "us_per_launch": round(roof["us_per_launch"], 2),
                    "linear_gb_per_token": round(m.linear_bytes_per_token() / 1e9, 3), "lm_head_gb_per_token": round(m.lm_head.numel() * 2 / 1e9, 3),
                    "whole_step_gbps": round(step_bytes * tps / 1e9, 1), "whole_step_frac": round(step_bytes * tps / 1e9 / HBM_PEAK_GBPS, 3),
                    "bits_usage": round(usage, 3), "finite_logits": bool(torch.isfinite(m.logits.float()).all().item()),
                    "shape": "hidden %d, mlp %d, %d q / %d kv heads, vocab %d%s%s" % (m.H, m.I, m.nh, m.nkv, m.vocab, ", q/k/v bias" if m.has_bias else "",
                                                                                       ", llama3 rope scaling" if m.inv_freq is not None else "")}
        return run

    def llama13b():
        # BASELINE.json configs[3] end to end (what `--config 4` prints as its own line): Llama-2-13B, one batched prompt pass of 16 x 2048 rows
        B, S = 16, 2048
        m, _, usage = build_model(dev, seed=0, max_seq=S, model="Llama-2-13b-hf", pinned=())
        ids = torch.randint(0, m.vocab - 1, (B, S), generator=torch.Generator().manual_seed(0)).to(dev)
        with torch.inference_mode():
            m.prefill_batch(ids)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(2):
                logits = m.prefill_batch(ids)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / 2
        total_flops = sum(2.0 * B * S * blk[name].N * blk[name].K for blk in m.blocks for name in m.cfg["linear"])
        return {"passes_per_s": round(1 / dt, 3), "ms_per_pass": round(dt * 1e3, 1), "whole_pass_tflops": round(total_flops / dt / 1e12, 1),
                "prompt_tokens_per_s": round(B * S / dt), "finite_logits": bool(torch.isfinite(logits.float()).all().item()),
                "bits_usage": round(usage, 3),
                "config": "BASELINE.json configs[3]: 16 x 2048 rows, whole model (linears + attention + norms + lm_head)"}

    def bf16_variant():
        # the optional bfloat16 entry points (include/amq_hip.h "bfloat16 variants"): a 4096 x 4096 3-bit layer with bf16 (scale, zero) --
        # weights against the oracle bit for bit, y against torch's CPU F.linear on them (bar: one bf16 ulp), and what a launch costs: the few-row kernel
        # over weights cold in HBM (72 copies in one graph), the batched end (dequantize once + the bf16 MFMA GEMM) at 8192 rows of the 13B gate/up shape
        import numpy as np
        from amq_amd.hqq_format import random_hqq
        from oracle import hqq_ref
        bits, n, k = 3, 4096, 4096
        h = random_hqq(n, k, bits, seed=11)
        sb, zb = h.scale.float().to(torch.bfloat16), h.zero.float().to(torch.bfloat16)
        as_bits = lambda t: t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)
        w_ref = hqq_ref.dequantize_bf16(h.W_q.numpy(), as_bits(sb), as_bits(zb), bits, (n, k), 128)
        qn, mn = ops.repack_from_hqq(h.W_q.to(dev), sb.reshape(-1).to(dev), zb.reshape(-1).to(dev), bits, n, k)
        exact = bool(np.array_equal(as_bits(ops.dequantize_bf16(qn, mn, bits, n, k)), w_ref))
        x = torch.randn(5, k, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16)
        y = ops.linear_bf16(x.to(dev), qn, mn, bits, n, k).float().cpu().double()
        ref = torch.nn.functional.linear(x, torch.from_numpy(w_ref.view(np.int16)).view(torch.bfloat16)).double()
        bar = 2.0 ** -7 * ref.abs() + 2.0 ** -8 * ref.pow(2).mean().sqrt()
        worst = float(((y - ref).abs() / bar).max())
        copies = [qn.clone() for _ in range(72)]                       # 72 x 6.3 MB: past the 256 MB last-level cache
        xd = x[:1].to(dev).contiguous()
        yd = torch.empty(1, n, dtype=torch.bfloat16, device=dev)
        t = _graph_time(dev, lambda: [ops.linear_bf16(xd, c, mn, bits, n, k, out=yd) for c in copies], 5) / len(copies)
        del copies
        M, N, K = 8192, 13824, 5120
        hb = random_hqq(N, K, bits, seed=12).to(dev)
        qb, mb = ops.repack_from_hqq(hb.W_q, hb.scale.float().to(torch.bfloat16).reshape(-1), hb.zero.float().to(torch.bfloat16).reshape(-1), bits, N, K)
        xb = torch.randn(M, K, device=dev).to(torch.bfloat16)
        yb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        tg = _graph_time(dev, lambda: ops.linear_bf16(xb, qb, mb, bits, N, K, out=yb), 5)
        return {"weights_bit_exact": exact, "max_err_over_bar": round(worst, 3), "ok": bool(exact and worst <= 1.0),
                "bar": "|y - y_cpu| <= 2^-7*|y_cpu| + 2^-8*rms(y_cpu) (one bf16 ulp)", "gemv_us_per_launch_4096x4096_3bit": round(t * 1e6, 2),
                "gemm_tflops_8192x13824x5120": round(2.0 * M * N * K / tg / 1e12, 1),
                "kernels": "amq::gemv_bf16_kernel; amq::dequant_native_bf16_kernel + amq::gemm_bf16_pp_kernel"}

    leg("decode_tokens_per_s_at_2048_cached_keys", long_cache)
    leg("decode_tokens_per_s_8_sequences", eight_sequences)
    leg("decode_tokens_per_s_reference_format_weights", reference_format)
    leg("decode_tokens_per_s_groupscale_math", groupscale)
    leg("llama70b_one_replica", llama70b)
    leg("llama13b_prompt_pass", llama13b)
    leg("llama31_8b", family("Llama-3.1-8B"))
    leg("qwen25_7b", family("Qwen2.5-7B"))
    leg("mistral_7b_v03", family("Mistral-7B-v0.3"))
    leg("bf16_variant", bf16_variant)
    return out


def _graph_time(dev, launches, reps):
    """seconds per replay of a hipGraph holding `launches()` (HIP events on the replay stream)"""
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
    return e0.elapsed_time(e1) * 1e-3 / reps


def gemv_roofline(m, reps=20):
    """time a graph holding only the GEMV launches of one token (4 per block)"""
    from amq_amd import ops
    H, I = m.H, m.I

    def launches():
        for blk in m.blocks:
            ops.gemv_grouped(m.x, [blk["self_attn.q_proj"].seg(m.q), blk["self_attn.k_proj"].seg(m.k),
                                   blk["self_attn.v_proj"].seg(m.v)], H, prologue=ops.PRO_RMSNORM, gamma=blk["ln1"], eps=m.eps)
            ops.gemv_grouped(m.att, [blk["self_attn.o_proj"].seg(m.q)], H)
            ops.gemv_grouped(m.x, [blk["mlp.gate_proj"].seg(m.gate), blk["mlp.up_proj"].seg(m.up)], H,
                             prologue=ops.PRO_RMSNORM, gamma=blk["ln2"], eps=m.eps)
            ops.gemv_grouped(m.gate, [blk["mlp.down_proj"].seg(m.q)], I, prologue=ops.PRO_SILU_MUL, x2=m.up)

    span_s = _graph_time(m.dev, launches, reps)
    n_launch = 4 * m.nb
    # algorithmic bytes: packed weights + fp16 scale/zero (the native buffers are exactly that) + x + y per launch
    wbytes = m.linear_bytes_per_token()
    xy = m.nb * (2 * H + 2 * (H + 2 * m.kvd)) + m.nb * (2 * H + 2 * H) + m.nb * (2 * H + 2 * 2 * I) + m.nb * (2 * 2 * I + 2 * H)
    alg = wbytes + xy
    return {"bytes_per_launch": alg / n_launch, "us_per_launch": span_s / n_launch * 1e6, "launches_per_token": n_launch,
            "gbps": alg / span_s / 1e9}


def gemv_layer_table(dev, iters=96):
    """Per-layer dequant-GEMV GB/s (the metric's second half): every decode launch shape of Llama-2-7B at uniform 2 / 3 /
    4 bit, M = 1, replayed from a hipGraph over rotating cold weight copies (>= 512 MB per case, beyond the Infinity Cache).
    us includes the inter-kernel boundary; bytes = N*K*b/8 + 4*N*K/128 + 2*K + 2*N."""
    from amq_amd import ops
    from amq_amd.llama import _synthetic_linear
    gen = torch.Generator(device=dev).manual_seed(123)
    H, I = 4096, 11008
    shapes = [("q/k/v", [(H, H)] * 3, ops.PRO_RMSNORM), ("o_proj", [(H, H)], ops.PRO_NONE),
              ("gate/up", [(I, H)] * 2, ops.PRO_RMSNORM), ("down_proj", [(H, I)], ops.PRO_SILU_MUL)]
    rows = []
    for name, segs, pro in shapes:
        K = segs[0][1]
        ntot = sum(n for n, _ in segs)
        for bits in (4, 3, 2):
            per = ntot * K * bits // 8 + 4 * ntot * K // 128
            copies = max(2, min(48, (512 << 20) // per + 1))
            w = [[_synthetic_linear(n, k, bits, gen, dev) for n, k in segs] for _ in range(copies)]
            x = torch.randn(1, K, device=dev, generator=gen).half()
            x2 = torch.randn(1, K, device=dev, generator=gen).half()
            gamma = torch.ones(K, device=dev, dtype=torch.float16)
            ys = [torch.empty(1, n, device=dev, dtype=torch.float16) for n, _ in segs]

            def launches():
                for i in range(iters):
                    ops.gemv_grouped(x, [l.seg(y) for l, y in zip(w[i % copies], ys)], K, prologue=pro, x2=x2, gamma=gamma, eps=1e-5)

            us = _graph_time(dev, launches, 3) / iters * 1e6
            b = per + 2 * K * (2 if pro == ops.PRO_SILU_MUL else 1) + 2 * ntot
            rows.append({"launch": name, "N": ntot, "K": K, "bits": bits, "us": round(us, 2), "GBps": round(b / us / 1e3, 1),
                         "frac": round(b / us / 1e3 / HBM_PEAK_GBPS, 3)})
            del w
    return rows


def mfma_roofline(dev, seed=0):
    """The batched path's roofline, driver-visible (north_star: "MFMA utilisation on the batched path against MI355X peak";
    reference counterpart of the measurement: GeMM mode, amq/utils/speed.py:61-71,95-105): the 28 linears of 4 Llama-2-13B
    blocks (synthesized avg-3-bit arch) at M = 16 x 2048 = 32768 rows -- BASELINE.json configs[3]'s row count -- through
    ops.gemm (the product dispatch), timed with HIP events on the launch stream after one warm-up pass; next to it the opt-in
    dequantize + library GEMM route on the same launches (comparison only)."""
    from amq_amd import arch, ops
    from amq_amd.llama import _synthetic_linear
    cfg = arch.MODEL_CONFIGS["Llama-2-13b-hf"]
    a, usage = arch.synthesize_arch(cfg, TARGET_BITS, seed=0, pinned=())
    gen = torch.Generator(device=dev).manual_seed(seed)
    M, nblk = 16 * 2048, 4
    lins = []
    for b in range(nblk):
        for name in cfg["linear"]:
            n, k = cfg["linear_shape"][name]
            lins.append(_synthetic_linear(n, k, arch.arch_bits(a["linear"], name, b), gen, dev))
    xs = {k: torch.randn(M, k, device=dev, dtype=torch.float16) * 0.05 for k in {l.K for l in lins}}
    ys = {n: torch.empty(M, n, device=dev, dtype=torch.float16) for n in {l.N for l in lins}}
    flops = sum(2.0 * M * l.N * l.K for l in lins)

    def linears():
        for l in lins:
            ops.gemm(xs[l.K], l.qn, l.mn, l.bits, l.mode, l.N, l.K, out=ys[l.N])

    def timed(fn=linears):
        fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e-3

    def linears_fused():                               # comparison: the fused unpack + MFMA ring kernel on the same launches
        for l in lins:
            ops.gemm(xs[l.K], l.qn, l.mn, l.bits, l.mode, l.N, l.K, out=ys[l.N], route=ops.GEMM_RING)

    with torch.inference_mode():
        t = timed()
        ops.LIB_GEMM_ROWS = 1024
        try:
            t_lib = timed()
        finally:
            ops.LIB_GEMM_ROWS = 0
        t_fused = timed(linears_fused)
    tf = flops / t / 1e12
    return {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
            "traffic": None, "kernel": ops.gemm_route_name(M), "launches": len(lins), "ms": t * 1e3, "rows": M,
            "comparison_fused_ring_kernel_tflops": flops / t_fused / 1e12,
            "workload": "the 28 linears of 4 Llama-2-13B blocks, avg-3-bit arch (bits_usage %.3f), M = 16 x 2048 rows "
                        "(BASELINE.json configs[3]); HIP events on the launch stream" % usage,
            "comparison_library_route_tflops": flops / t_lib / 1e12}


def dequant_hqq_table(dev):
    """f-4: the standalone HQQ Format A -> fp16 dequantize kernel on three layer shapes x 2/3/4 bit (tools/dequant_hqq_bench.py:
    warm-up, rotating buffer sets, HIP events)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dequant_hqq_bench", os.path.join(ROOT, "tools", "dequant_hqq_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.measure(dev)


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
    # the thread sweep on a 4-block sample, then -- at the best thread count -- 8 greedy tokens of the WHOLE 32-block model
    return cb.time_decode_linears(layers, cfg["n_block"], tokens=3, extra_dense=lm_head, sample_blocks=4,
                                  full_model=(cfg["num_heads"], cfg["num_kv_heads"]), full_tokens=8)


def host_cpu_info():
    """model string and PHYSICAL core count of the host the CPU baseline ran on ("core count + CPU model string")"""
    model, cores = None, set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or None)


def parity_gate(dev, seed=7):
    """The same-run parity gate (north_star: "outputs match the reference CPU nn.Linear-on-dequantized-weights path
    within 1e-3 relative fp16 tolerance ... in the same run"): for one 4096 x 4096 group-128 layer per bit-width (configs[0]'s shape),
    y_cpu = F.linear(x, W_deq) on the host with the ORACLE's dequantized weights (hqq/utils/patching.py:95-100 semantics) against
    y_gpu = the product GEMV / GEMM on the repacked payload of the very same layer, 1 and 5 rows through amq::gemv_kernel and 64 rows
    through the few-row MFMA kernel.  The bar is the tests' bar; the run FAILS (non-zero exit) outside it."""
    from amq_amd import ops
    from amq_amd.hqq_format import random_hqq
    from oracle import cpu_baseline as cb, hqq_ref
    n = k = 4096
    worst_bar, worst_rel, worst_rms, cases, worst_gs, rms_gs = 0.0, 0.0, 0.0, 0, 0.0, 0.0
    for bits in (2, 3, 4):
        h = random_hqq(n, k, bits, seed=seed + bits)
        w_deq = cb.dequantize_torch(h.W_q, h.scale, h.zero, bits, (n, k))                 # the oracle port (checked against the numpy
        w_np = hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), bits, (n, k))   # restatement right here)
        assert np.array_equal(w_deq.numpy().view(np.uint16), w_np.view(np.uint16))
        hd = h.to(dev)
        qn, mn = ops.repack_from_hqq(hd.W_q, hd.scale.reshape(-1), hd.zero.reshape(-1), bits, n, k)
        assert torch.equal(ops.dequantize(qn, mn, bits, ops.MODE_HQQ, n, k).cpu(), w_deq)          # weights: bit for bit
        for rows in (1, 5, 64):
            x = torch.randn(rows, k, generator=torch.Generator().manual_seed(seed + rows)).to(torch.float16)
            with torch.inference_mode():
                y_cpu = torch.nn.functional.linear(x.float(), w_deq.float()).to(torch.float16).float()   # fp32 accumulate, one fp16 rounding
                xg = x.to(dev)
                y_gpu = (ops.gemv(xg, qn, mn, bits, ops.MODE_HQQ, n, k) if rows <= 16 else
                         ops.gemm(xg, qn, mn, bits, ops.MODE_HQQ, n, k)).float().cpu()
                if rows <= 16:      # (information: the same launch under the opt-in GROUPSCALE arithmetic)
                    y_gs = ops.gemv(xg, qn, mn, bits, ops.MODE_HQQ, n, k, opts=ops.GemvOpts(math=ops.MATH_GROUPSCALE)).float().cpu()
            rms = float(y_cpu.pow(2).mean().sqrt())
            err = (y_gpu - y_cpu).abs()
            worst_bar = max(worst_bar, float((err / (1e-3 * y_cpu.abs() + 1e-3 * rms)).max()))
            big = y_cpu.abs() >= rms                                                       # relative error where "relative" means something
            worst_rel = max(worst_rel, float((err[big] / y_cpu.abs()[big]).max()))
            worst_rms = max(worst_rms, float(err.max()) / rms)
            if rows <= 16:
                e_gs = (y_gs - y_cpu).abs()
                worst_gs = max(worst_gs, float((e_gs / (1e-3 * y_cpu.abs() + 1e-3 * rms)).max()))
                rms_gs = max(rms_gs, float(e_gs.pow(2).mean().sqrt()) / rms)
            cases += 1
    return {"ok": worst_bar <= 1.0, "max_err_over_bar": worst_bar, "max_rel_err": worst_rel, "max_err_over_rms": worst_rms,
            "bits": [2, 3, 4], "rows": [1, 5, 64], "shape": [n, k], "cases": cases, "weights_bit_exact": True,
            "gemv_math": gemv_math_name(),
            "optin_groupscale_math": {"max_err_over_bar": worst_gs, "rms_err_over_rms": rms_gs, "rows": [1, 5],
                                      "note": "information only: AMQ_MATH_GROUPSCALE is opt-in because of this margin (and 2-ulp misses behind a bias add)"},
            "bar": "|y_gpu - y_cpu| <= 1e-3*|y_cpu| + 1e-3*rms(y_cpu); max_rel_err over outputs with |y_cpu| >= rms",
            "cpu_side": "torch CPU F.linear (fp32 accumulate, one fp16 rounding) on the oracle's dequantized fp16 weights"}


def gemv_math_name():
    """the GEMV arithmetic this build of the library runs by default (include/amq_hip.h)"""
    from amq_amd import _lib
    return {_lib.MATH_EXACT: "EXACT (the reference's two fp16 roundings per weight, bit-identical weights)",
            _lib.MATH_GROUPSCALE: "GROUPSCALE (first fp16 rounding per weight exact, scale applied per 128-group in fp32 after the MFMAs)"}[_lib.load().amq_default_gemv_math()]


def load_traffic():
    """HBM bytes per GEMV launch from this round's committed rocprofv3 PMC pass over the CURRENT kernels
    (profiles/r06_gemv_pmc.json, tools/collect_round.sh r06), or None -- never a stale constant."""
    p = os.path.join(ROOT, "profiles", "r06_gemv_pmc.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def run_decode(args, rep, dev):
    rank, n_gpus = rep.rank, rep.world
    max_seq = PROMPT + args.warmup + args.steps + 8
    big = args.config == 5                      # BASELINE.json configs[4]: Llama-2-70B, one decode stream per GPU
    model = "Llama-2-70b-hf" if big else MODEL
    m, a, usage = build_model(dev, seed=rank, max_seq=max_seq, model=model, pinned=() if big else None,
                              engine=False if args.five_launch else (True if args.engine else None))
    if args.fuse_qkv_attn:
        m.fuse_qkv_attn = True
    ids = torch.randint(0, m.vocab - 1, (PROMPT,), generator=torch.Generator().manual_seed(rank)).to(dev)
    m.prefill(ids)                                  # un-timed (GeMV-mode protocol)
    m.capture()
    for _ in range(args.warmup):
        m.decode_step()
    elapsed = rep.timed(m.decode_step, args.steps, sync=lambda: torch.cuda.synchronize(dev))
    per_rank = rep.gather(args.steps / rep.last_local)
    m.check()                                       # no step ran past the KV cache
    ok = bool(torch.isfinite(m.logits.float()).all().item())
    if rank != 0:
        return
    roof = gemv_roofline(m)
    out = {
        "metric": "decode tokens/s, Llama-2-%s AMQ mixed 2/3/4-bit avg-3-bit, batch 1%s" % (
            "70B" if big else "7B", ", %d independent streams" % n_gpus if big else ""),
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
        "config": {"workload": "Llama-2-%s shapes, synthesized avg-3.0-bit per-layer arch (bits_usage %.3f), "
                               "batch 1 decode after a 64-token prefill, one stream per GPU%s" % (
                                   "70B" if big else "7B", usage, " (BASELINE.json configs[4])" if big else ""),
                   "parallelism": "replicas x%d" % n_gpus, "prompt": PROMPT, "group_size": 128},
        "roofline": {"bound": "hbm", "achieved": roof["gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": roof["gbps"] / HBM_PEAK_GBPS, "traffic": None if big else load_traffic(),
                     "kernel": "amq::gemv_kernel (grouped 2/3/4-bit weight-streaming GEMV; arithmetic: %s)" % gemv_math_name(),
                     "bytes_per_launch": roof["bytes_per_launch"], "us_per_launch": roof["us_per_launch"],
                     "launches_per_token": roof["launches_per_token"]},
        "finite_logits": ok,
        # what "parity-green" means in tests/ (north_star: 1e-3 relative fp16 tolerance): element-wise
        # |y - y_ref| <= 1e-3 |y_ref| + 1e-3 rms(y_ref) against the CPU nn.Linear-on-dequantized-weights oracle -- the rms term is
        # a floor for outputs that cancel to ~0; dequantized weights and repacks are compared bit for bit
        "parity_bar": "|y - y_ref| <= 1e-3*|y_ref| + 1e-3*rms(y_ref) vs CPU nn.Linear on oracle-dequantized weights; weights/repack bit-exact",
        # read from the LIVE process group (not from the command line): what the barrier / all-reduce / all-gather spanned
        "rccl_world_size": rep.live_world_size(),
        "rccl_backend": rep.live_backend(),
        "per_rank_tokens_per_s": [round(v, 2) for v in per_rank],
        "linear_gb_per_token": m.linear_bytes_per_token() / 1e9,
        "model_gbps": m.total_bytes_per_token(PROMPT + args.warmup + args.steps // 2) / (elapsed / args.steps) / 1e9,
    }
    if big:
        args.no_layer_table = args.no_mfma = args.no_cpu_baseline = True      # 7B / 13B-specific extras of the default line
    if n_gpus == 1 and not args.no_layer_table:
        del m
        torch.cuda.empty_cache()
        out["gemv_layers"] = gemv_layer_table(dev)
        out["beyond_the_metric"] = beyond_the_metric(dev)
    if n_gpus == 1 and not args.no_mfma:
        try:
            out["mfma_roofline"] = mfma_roofline(dev)
        except Exception as e:      # noqa: BLE001  (an extra of the line: never costs the headline)
            out["mfma_roofline"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            out["dequant_hqq"] = dequant_hqq_table(dev)
        except Exception as e:      # noqa: BLE001
            out["dequant_hqq"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    gate_ok = True
    if n_gpus == 1 and not args.no_cpu_baseline:
        out["parity"] = parity_gate(dev)
        gate_ok = out["parity"]["ok"]
        cb = cpu_baseline()
        cpu_model, physical = host_cpu_info()
        out["cpu_baseline"] = {"value": cb["tokens_per_s_full_model"], "unit": "tokens/s", "cores": cb["full_model_threads"],
                               "best_threads": cb["full_model_threads"], "whole_model_thread_probe_tokens_per_s": cb["full_model_thread_probe"], "physical_cores": physical, "cpu_model": cpu_model,
                               "kind": "port", "sample": cb["full_model_sample"], "host_threads": cb["host_threads"],
                               "seconds_per_token": cb["full_model_seconds_per_token"],
                               "linears_only_sample_tokens_per_s": cb["tokens_per_s_predequantized"], "thread_sweep_sample": cb["sample"],
                               "thread_sweep_tokens_per_s": cb["thread_sweep"],
                               "dequant_every_call_tokens_per_s": cb["tokens_per_s_dequant_every_call"]}
    print(json.dumps(out), flush=True)
    if not gate_ok:
        raise SystemExit("parity gate failed: GPU output outside the bar against the CPU reference path (see \"parity\" in the line above)")


def run_gemm_mode(args, rep, dev):
    """BASELINE.json configs[3]: Llama-2-13B avg-3-bit, 16 x 2048 prompt rows per pass (GeMM mode)."""
    from amq_amd import ops
    rank, n_gpus = rep.rank, rep.world
    B, S = 16, 2048
    m, a, usage = build_model(dev, seed=rank, max_seq=S, model="Llama-2-13b-hf", pinned=())
    ids = torch.randint(0, m.vocab - 1, (B, S), generator=torch.Generator().manual_seed(rank)).to(dev)
    with torch.inference_mode():
        for _ in range(args.warmup):
            m.prefill_batch(ids)
        elapsed = rep.timed(lambda: m.prefill_batch(ids), args.steps, sync=lambda: torch.cuda.synchronize(dev))
        logits = m.prefill_batch(ids)
    ok = bool(torch.isfinite(logits.float()).all().item())
    if rank != 0:
        return
    # the dominant kernel family: the seven linears of every block at M = 32768 (what ops.gemm dispatches to),
    # timed alone with HIP events on the launch stream
    M = B * S
    xs = {k: torch.randn(M, k, device=dev, dtype=torch.float16) * 0.05 for k in (m.H, m.I)}
    flops = 0
    with torch.inference_mode():
        def linears():
            for blk in m.blocks[:4]:
                for name in m.cfg["linear"]:
                    l = blk[name]
                    ops.gemm(xs[l.K], l.qn, l.mn, l.bits, l.mode, l.N, l.K)
        linears()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        linears()
        e1.record()
        torch.cuda.synchronize(dev)
    for blk in m.blocks[:4]:
        for name in m.cfg["linear"]:
            flops += 2.0 * M * blk[name].N * blk[name].K
    tf = flops / (e0.elapsed_time(e1) * 1e-3) / 1e12
    # comparison only (not the product path): the opt-in dequantize + library GEMM route on the same launches
    ops.LIB_GEMM_ROWS = 1024
    with torch.inference_mode():
        linears()
        torch.cuda.synchronize(dev)
        e0.record()
        linears()
        e1.record()
        torch.cuda.synchronize(dev)
    ops.LIB_GEMM_ROWS = 0
    tf_lib = flops / (e0.elapsed_time(e1) * 1e-3) / 1e12
    total_flops = sum(2.0 * M * blk[name].N * blk[name].K for blk in m.blocks for name in m.cfg["linear"])
    out = {
        "metric": "batched prompt passes/s (GeMM mode), Llama-2-13B AMQ mixed 2/3/4-bit avg-3-bit, 16 x 2048 rows",
        "value": n_gpus * args.steps / elapsed, "unit": "passes/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "Llama-2-13B shapes, synthesized avg-3.0-bit arch (bits_usage %.3f), one batched prompt pass "
                               "of 16 x 2048 rows (BASELINE.json configs[3])" % usage,
                   "parallelism": "replicas x%d" % n_gpus, "group_size": 128, "gemm_route": ops.gemm_route_name(M)},
        "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
                     "traffic": None, "kernel": ops.gemm_route_name(M),
                     "note": "the seven linears of 4 blocks at M = 32768 through ops.gemm, HIP events on the launch stream"},
        "prompt_tokens_per_s": n_gpus * args.steps * M / elapsed,
        "linear_tflop_per_pass": total_flops / 1e12,
        "whole_pass_tflops": total_flops * args.steps / elapsed / 1e12,
        "comparison_library_route_tflops": tf_lib,
        "finite_logits": ok,
    }
    print(json.dumps(out), flush=True)


def rehearse_ranks(args):
    """``--rehearse-ranks``: everything of an N-rank run EXCEPT the GPU work, on host cores with the gloo backend standing in for RCCL -- the launch path
    (launch_local or an outer launcher), each rank's CPU placement, the rendezvous, ``rep.timed`` (barrier, max over ranks), the gather of per-rank
    rates, rank 0 alone printing, the closing barrier.  The step is a 2 ms sleep; the line says so (``rehearsal``: true, ``value``: null): it is a test
    of the plumbing an 8-GPU node will run (tests/test_replicas_cpu.py), never a number."""
    import time
    from amq_amd.replicas import Replicas
    rep = Replicas(backend="gloo")
    try:
        for _ in range(args.warmup):
            time.sleep(0.002)
        elapsed = rep.timed(lambda: time.sleep(0.002 * (1 + rep.rank % 2)), args.steps)
        per_rank = rep.gather(args.steps / rep.last_local)
        cpus = rep.gather(float(len(os.sched_getaffinity(0))))
        if rep.rank == 0:
            print(json.dumps({
                "metric": "REHEARSAL of the %d-rank plumbing (no GPU work): not a measurement" % rep.world, "value": None, "unit": "tokens/s",
                "rehearsal": True, "n_gpus": rep.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "scaling": "weak", "data": "none (a 2 ms sleep per step)", "config": {"workload": "BASELINE.json configs[%d] launch path" % (args.config - 1),
                                                                                         "parallelism": "replicas x%d" % rep.world},
                "rccl_world_size": rep.live_world_size(), "rccl_backend": rep.live_backend() + " (stand-in for nccl = RCCL)",
                "per_rank_tokens_per_s": [round(v, 2) for v in per_rank], "per_rank_host_cores": [int(c) for c in cpus]}), flush=True)
        rep.barrier()
    finally:
        rep.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=(3, 4, 5),
                    help="3 (default): BASELINE.json configs[2], the headline decode metric; 4: configs[3], GeMM mode; "
                         "5: configs[4], Llama-2-70B decode streams (one per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-layer-table", action="store_true")
    ap.add_argument("--engine", action="store_true", help="A/B: decode steps through the one-launch-per-token engine")
    ap.add_argument("--fuse-qkv-attn", action="store_true", help="A/B: q/k/v + attention of a block as one launch (4 launches per block)")
    ap.add_argument("--five-launch", action="store_true",
                    help="A/B: decode steps as five launches per block instead of the one-launch-per-token engine")
    ap.add_argument("--no-mfma", action="store_true", help="skip the batched-path (MFMA) roofline and the dequantize rows")
    ap.add_argument("--rehearse-ranks", action="store_true",
                    help="NOT a measurement: rehearse the N-rank plumbing of this script on host cores -- launch, CPU placement, gloo rendezvous, barrier / "
                         "max-over-ranks / gather, rank 0's line -- with a sleep in place of the decode step (no model, no kernels; value is null)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {3: 256, 5: 64}.get(args.config, 3)
    if args.warmup is None:
        args.warmup = {3: 16, 5: 8}.get(args.config, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing above has touched the GPU (`import torch` does not), and the children are
        # fresh processes -- never a re-exec of a process that holds a device context.
        from amq_amd.replicas import launch_local
        sys.exit(launch_local(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")

    if world > 1 and "AMQ_RENDEZVOUS_FILE" not in os.environ:
        # ranks started by torch.distributed.run are not placed by anyone: bind this one to its GPU's NUMA node (its share of those cores)
        # before the first GPU call creates the runtime's threads.  launch_local's children (AMQ_RENDEZVOUS_FILE set) arrive already bound:
        # computing the share again over the narrowed mask would divide it among the node's peers a second time
        from amq_amd.replicas import pin_rank_cpus
        pin_rank_cpus()
    if args.rehearse_ranks:
        return rehearse_ranks(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    from amq_amd.replicas import Replicas
    rep = Replicas(backend="nccl", device=dev)        # nccl == RCCL on ROCm; no-op for one process
    try:
        if args.config == 4:
            run_gemm_mode(args, rep, dev)
        else:
            run_decode(args, rep, dev)
        rep.barrier()               # rank 0 finishes its extra measurements before anyone tears the process group down
    finally:
        rep.close()


if __name__ == "__main__":
    main()
