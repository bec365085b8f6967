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
A bounded sample is timed (the 7 linears of ``blocks`` decoder blocks for
``tokens`` decode tokens) and extrapolated to the whole model per token.
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


def time_decode_linears(layers, n_block_total, tokens=8, extra_dense=None):
    """layers: list of dicts {W_q, scale, zero, nbits, shape} (torch CPU tensors) = the sampled blocks'
    linears, in forward order.  Returns dict with tokens/s extrapolated to ``n_block_total`` blocks.
    extra_dense: optional fp16 [N,K] (lm_head) timed once per token and added un-scaled."""
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    # check the port against the numpy oracle on the first layer (the checker checks itself)
    l0 = layers[0]
    w_t = dequantize_torch(l0["W_q"], l0["scale"], l0["zero"], l0["nbits"], l0["shape"])
    w_o = hqq_ref.dequantize(l0["W_q"].numpy(), l0["scale"].numpy(), l0["zero"].numpy(), l0["nbits"], l0["shape"])
    assert np.array_equal(w_t.numpy().view(np.uint16), w_o.view(np.uint16)), "CPU port disagrees with the oracle"

    deq = [dequantize_torch(l["W_q"], l["scale"], l["zero"], l["nbits"], l["shape"]) for l in layers]
    xs = [torch.randn(1, l["shape"][1]).to(torch.float16) for l in layers]
    with torch.inference_mode():
        for w, x in zip(deq, xs):                       # warm-up
            torch.nn.functional.linear(x, w)
        t = []
        for _ in range(tokens):
            t0 = time.perf_counter()
            for w, x in zip(deq, xs):
                torch.nn.functional.linear(x, w)
            t.append(time.perf_counter() - t0)
        t_pre = float(np.median(t))
        t0 = time.perf_counter()
        for l, x in zip(layers, xs):                    # (ii) dequantize every call, one token
            torch.matmul(x, dequantize_torch(l["W_q"], l["scale"], l["zero"], l["nbits"], l["shape"]).T)
        t_deq = time.perf_counter() - t0
        t_dense = 0.0
        if extra_dense is not None:
            xd = torch.randn(1, extra_dense.shape[1]).to(torch.float16)
            torch.nn.functional.linear(xd, extra_dense)
            t0 = time.perf_counter()
            for _ in range(3):
                torch.nn.functional.linear(xd, extra_dense)
            t_dense = (time.perf_counter() - t0) / 3
    n_sampled = len(layers) / 7.0
    scale_up = n_block_total / n_sampled
    return {
        "tokens_per_s_predequantized": 1.0 / (t_pre * scale_up + t_dense),
        "tokens_per_s_dequant_every_call": 1.0 / (t_deq * scale_up + t_dense),
        "cores": threads,
        "sample": f"{int(n_sampled)} of {n_block_total} decoder blocks x 7 linears, {tokens} tokens (median), "
                  f"+ lm_head; linears only, extrapolated x{scale_up:.0f}",
    }
