"""CPU baseline leg of bench.py: the reference's CPU path timed on host cores.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never on the
product path.

The reference's CPU forward for a quantized linear is
``torch.matmul(x, self.dequantize().T)`` (hqq/utils/patching.py:95-100 with
Quantizer.dequantize, hqq/core/quantize.py:184-199).  The reference's Python
cannot travel to the GPU box, so this is the port: the same two steps with
torch CPU ops on all host threads --
  (i)  ``F.linear(x, W_deq)`` on pre-dequantized fp16 weights (the fastest the
       reference's CPU path can possibly be: dequantization hoisted out), and
  (ii) unpack + ``(W_r - zero) * scale`` + matmul on every call (what
       ``forward_hqq_inferece`` literally does).
A bounded sample is timed per thread count (the 7 linears of ``blocks`` decoder blocks
for ``tokens`` decode tokens, extrapolated: the thread sweep); at the best thread count
the WHOLE model then decodes ``full_tokens`` greedy tokens (">= 8
greedy decode tokens of the synthetic Llama-2-7B, pre-dequantized fp16, ~13 GB host
RAM"): every block's seven matrices in their own memory, RMSNorm, rotary embedding, KV
cache, softmax attention, SiLU-gated MLP, final norm, lm_head, arg-max -- the forward the
reference's CPU model runs -- and that is the value reported.
"""
import os
import time

import numpy as np
import torch

from . import hqq_ref


def _unpack_torch(W_q, nbits, rows):
    """BitPack.unpack_* with torch CPU ops (bitpack.py:30-110), fp16 result like the reference."""
    if nbits == 4:
        return torch.cat([(W_q & 0xF0) >> 4, W_q & 0x0F], 0).to(torch.float16)[:rows]
    if nbits == 2:
        return torch.cat([(W_q >> 6) & 3, (W_q >> 4) & 3, (W_q >> 2) & 3, W_q & 3], 0).to(torch.float16)[:rows]
    return torch.cat([(W_q >> (27 - 3 * c)) & 7 for c in range(10)], 0).to(torch.float16)[:rows]


def dequantize_torch(W_q, scale, zero, nbits, shape):
    n, k = shape
    w_r = _unpack_torch(W_q, nbits, n * k // 128)
    return ((w_r - zero) * scale).reshape(n, k)


def _time_tokens(mats, xs, tokens):
    """seconds per pass over ``mats`` (median over ``tokens`` passes after one warm-up pass)"""
    with torch.inference_mode():
        for w, x in zip(mats, xs):
            torch.nn.functional.linear(x, w)
        t = []
        for _ in range(tokens):
            t0 = time.perf_counter()
            for w, x in zip(mats, xs):
                torch.nn.functional.linear(x, w)
            t.append(time.perf_counter() - t0)
    return float(np.median(t))


def _rmsnorm(x, eps):
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(torch.float16)        # (gamma = 1: synthetic)


class FullModel:
    """A whole synthetic decoder on the host: ``n_block`` blocks, each with its OWN copy of the seven pre-dequantized fp16 matrices of ``block`` (forward
    order q, k, v, o, gate, up, down; distinct memory: a token's weights fit no cache), gamma = 1, the embedding = the lm_head's rows.  ``step`` is the
    forward the reference's CPU model runs for one new token: RMSNorm, q / k / v, rotary embedding, cache append, softmax attention over the cache,
    o_proj + residual, RMSNorm, SiLU-gated MLP + residual; ``next_token`` adds the final norm, the lm_head and the arg-max."""

    def __init__(self, block, n_block, n_heads, n_kv_heads, lm_head, max_ctx=64, eps=1e-5, theta=10000.0):
        self.blocks = [block] + [[w.clone() for w in block] for _ in range(n_block - 1)]
        self.nh, self.nkv, self.lm_head, self.eps = n_heads, n_kv_heads, lm_head, eps
        self.H = block[0].shape[1]
        self.d = self.H // n_heads
        self.kc = [torch.zeros(n_kv_heads, max_ctx, self.d, dtype=torch.float16) for _ in range(n_block)]
        self.vc = [torch.zeros(n_kv_heads, max_ctx, self.d, dtype=torch.float16) for _ in range(n_block)]
        self.inv = 1.0 / (theta ** (torch.arange(0, self.d, 2, dtype=torch.float32) / self.d))
        self.pos, self.tok = 0, 1

    def _rope(self, t, pos):
        fr = pos * self.inv
        cos, sin = torch.cat([fr, fr]).cos().to(torch.float16), torch.cat([fr, fr]).sin().to(torch.float16)
        return t * cos + torch.cat([-t[..., self.d // 2:], t[..., :self.d // 2]], -1) * sin

    def next_token(self):
        F = torch.nn.functional
        pos, d, rep = self.pos, self.d, self.nh // self.nkv
        x = self.lm_head[self.tok:self.tok + 1].clone()
        for b, (wq, wk, wv, wo, wg, wu, wd) in enumerate(self.blocks):
            h = _rmsnorm(x, self.eps)
            q = self._rope(F.linear(h, wq).view(self.nh, d), pos)
            self.kc[b][:, pos] = self._rope(F.linear(h, wk).view(self.nkv, d), pos)
            self.vc[b][:, pos] = F.linear(h, wv).view(self.nkv, d)
            k = self.kc[b][:, :pos + 1].repeat_interleave(rep, 0)
            v = self.vc[b][:, :pos + 1].repeat_interleave(rep, 0)
            p = torch.softmax(torch.einsum("hd,htd->ht", q.float(), k.float()) / d ** 0.5, -1).to(torch.float16)
            a = torch.einsum("ht,htd->hd", p.float(), v.float()).to(torch.float16).reshape(1, self.H)
            x = x + F.linear(a, wo)
            h2 = _rmsnorm(x, self.eps)
            x = x + F.linear(F.silu(F.linear(h2, wg)) * F.linear(h2, wu), wd)
        self.tok = int(F.linear(_rmsnorm(x, self.eps), self.lm_head).float().argmax())
        self.pos += 1
        return self.tok

    def timed_tokens(self, n):
        """seconds of each of ``n`` greedy tokens"""
        out = []
        with torch.inference_mode():
            for _ in range(n):
                t0 = time.perf_counter()
                self.next_token()
                out.append(time.perf_counter() - t0)
        return out


def decode_full_model(block, n_block, n_heads, n_kv_heads, lm_head, tokens=8, thread_counts=None, probe_tokens=2):
    """``tokens`` greedy decode tokens of the whole model at the best of ``thread_counts`` (each probed with ``probe_tokens`` tokens after one un-timed
    token that touches every page: the whole forward -- small normalisation / attention ops between the GEMVs -- does not peak where a GEMV alone does).
    Returns (seconds per token: list, threads used, {threads: probe tokens/s})."""
    thread_counts = list(thread_counts or [torch.get_num_threads()])
    m = FullModel(block, n_block, n_heads, n_kv_heads, lm_head, max_ctx=2 + len(thread_counts) * probe_tokens + tokens)
    torch.set_num_threads(thread_counts[0])
    m.timed_tokens(1)                                   # un-timed: first touch of every page
    probe = {}
    for nt in thread_counts:
        torch.set_num_threads(nt)
        probe[nt] = 1.0 / float(np.median(m.timed_tokens(probe_tokens)))
    best = max(probe, key=probe.get)
    torch.set_num_threads(best)
    return m.timed_tokens(tokens), best, probe


def time_decode_linears(layers, n_block_total, tokens=3, extra_dense=None, sample_blocks=4, thread_counts=None,
                        budget_s=25.0, full_model=None, full_tokens=8):
    """layers: list of dicts {W_q, scale, zero, nbits, shape} (torch CPU tensors) = ONE decoder block's seven linears in
    forward order, real HQQ payloads (the port is checked against the numpy oracle on them).  ``sample_blocks`` blocks
    are timed per token: block 0 holds the dequantized weights of ``layers``; the others are distinct fp16 matrices of the
    same shapes (values do not affect the speed of a GEMV; distinct memory so that the CPU caches do not help), and the
    result is extrapolated to ``n_block_total`` blocks.  The thread count is SWEPT (an M = 1 fp16 GEMV does not scale to
    hundreds of threads: pinning os.cpu_count() threads on it measured 29 s per token on a 256-thread EPYC); the best
    setting is reported, the whole sweep returned.  extra_dense: optional fp16 [N,K] (lm_head), added un-scaled.
    full_model: (n_heads, n_kv_heads) -> at the best thread count the WHOLE model (``n_block_total`` blocks, each its own copy of the
    dequantized block, + extra_dense as lm_head / embedding) decodes ``full_tokens`` greedy tokens (decode_full_model): ``tokens_per_s_full_model``."""
    ncpu = os.cpu_count() or 1
    if thread_counts is None:
        thread_counts = sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu})
    # check the port against the numpy oracle on the first layer (the checker checks itself)
    l0 = layers[0]
    w_t = dequantize_torch(l0["W_q"], l0["scale"], l0["zero"], l0["nbits"], l0["shape"])
    w_o = hqq_ref.dequantize(l0["W_q"].numpy(), l0["scale"].numpy(), l0["zero"].numpy(), l0["nbits"], l0["shape"])
    assert np.array_equal(w_t.numpy().view(np.uint16), w_o.view(np.uint16)), "CPU port disagrees with the oracle"

    torch.set_num_threads(min(ncpu, 32))
    deq = [dequantize_torch(l["W_q"], l["scale"], l["zero"], l["nbits"], l["shape"]) for l in layers]
    g = torch.Generator().manual_seed(1)
    mats = list(deq)
    for _ in range(sample_blocks - 1):
        mats += [(torch.randn(w.shape, generator=g, dtype=torch.float32) * 0.02).to(torch.float16) for w in deq]
    xs = [torch.randn(1, w.shape[1], generator=g).to(torch.float16) for w in mats]
    xd = torch.randn(1, extra_dense.shape[1], generator=g).to(torch.float16) if extra_dense is not None else None
    scale_up = n_block_total / float(sample_blocks)

    sweep, t_start = {}, time.perf_counter()
    for nt in thread_counts:
        if sweep and time.perf_counter() - t_start > budget_s:
            break                                       # bounded: the default bench run must stay within minutes
        torch.set_num_threads(nt)
        t_pre = _time_tokens(mats, xs, tokens)
        t_dense = _time_tokens([extra_dense], [xd], tokens) if extra_dense is not None else 0.0
        sweep[nt] = 1.0 / (t_pre * scale_up + t_dense)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    with torch.inference_mode():                        # (ii) dequantize every call: one block, one token, best thread count
        t0 = time.perf_counter()
        for l, x in zip(layers, xs):
            torch.matmul(x, dequantize_torch(l["W_q"], l["scale"], l["zero"], l["nbits"], l["shape"]).T)
        t_deq = time.perf_counter() - t0
        t_dense = _time_tokens([extra_dense], [xd], 1) if extra_dense is not None else 0.0
    full = {}
    if full_model is not None and extra_dense is not None:
        del mats, xs                                    # (the sample's matrices: 1.6 GB the whole model does not need)
        cand = sorted(sweep, key=sweep.get, reverse=True)[:3]
        t_tok, full_threads, probe = decode_full_model(deq, n_block_total, full_model[0], full_model[1], extra_dense, tokens=full_tokens, thread_counts=cand)
        torch.set_num_threads(best)
        full = {"tokens_per_s_full_model": 1.0 / float(np.median(t_tok)), "full_model_tokens": len(t_tok), "full_model_threads": full_threads,
                "full_model_thread_probe": {str(k): round(v, 3) for k, v in probe.items()},
                "full_model_seconds_per_token": [round(t, 5) for t in t_tok],
                "full_model_sample": f"{n_block_total} of {n_block_total} blocks, {len(t_tok)} greedy tokens (pre-dequantized fp16 weights, "
                                     f"{sum(w.numel() for w in deq) * 2 * n_block_total / 1e9:.1f} GB; whole decoder forward incl. norms, rotary embedding, attention over the "
                                     f"cache, lm_head, arg-max) at {full_threads} threads = the best of {sorted(probe)} probed on the whole model (2 tokens each, after one "
                                     f"un-timed token); tokens/s = 1 / median token time"}
    return {
        **full,
        "tokens_per_s_predequantized": sweep[best],
        "tokens_per_s_dequant_every_call": 1.0 / (t_deq * n_block_total + t_dense),
        "cores": best,
        "host_threads": ncpu,
        "thread_sweep": {str(k): round(v, 4) for k, v in sweep.items()},
        "sample": f"{sample_blocks} of {n_block_total} decoder blocks x 7 linears (block 0: HQQ weights dequantized by the "
                  f"oracle port; the rest: fp16 matrices of the same shapes), median of {tokens} tokens per thread count, "
                  f"+ lm_head; extrapolated x{scale_up:.0f}; threads swept over {sorted(sweep)} of {ncpu}, best reported",
    }
