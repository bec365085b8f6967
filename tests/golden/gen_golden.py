#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference on CPU.

Run only in the build container (needs /root/reference).  It imports the
vendored HQQ package that AMQ ships (amq/kernel/hqq) with two stub modules for
missing optional deps, drives the reference's own classes, and writes small
``.npz`` fixtures next to this file.  The fixtures are data only (inputs and
the reference's outputs); no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

What is captured (per case ``hqq_b{bits}_{N}x{K}.npz``):
  W            fp16 [N,K]   the weight that was quantized
  W_q          HQQLinear.W_q          (Format A payload)
  scale, zero  HQQLinear.meta[...]    fp16 [N*K/G, 1]
  W_deq        HQQLinear.dequantize() fp16 [N,K]       <- the parity weight
  gptq_qweight/gptq_scales/gptq_zeros   GPTQLinear buffers after patch_hqq_to_gptq
  gptq_x, gptq_y   x[128,K] and GPTQLinear.forward(x) (torch fallback branch)
  awq_qweight/awq_scales/awq_scaled_zeros  FT_QuantLinear buffers (4-bit only)
  x, y_ref     x[3,K] and torch.matmul(x, W_deq.T) (+bias) on CPU, fp16
  bias         fp16 [N] or absent
and ``bitpack.npz``: BitPack.pack_* outputs for random integer matrices,
``pack_intweight.npz``: ft.pack_intweight on random 4-bit matrices.
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference/amq/kernel/hqq"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stubs():
    d = tempfile.mkdtemp(prefix="amq_stubs_")
    os.makedirs(f"{d}/termcolor")
    os.makedirs(f"{d}/faster_transformer")
    with open(f"{d}/termcolor/__init__.py", "w") as f:
        f.write("def colored(s,*a,**k): return s\n")
    with open(f"{d}/faster_transformer/__init__.py", "w") as f:
        f.write("def gemv_4bit(*a,**k): raise NotImplementedError\n"
                "def gemm_4bit(*a,**k): raise NotImplementedError\n")
    return d


def main():
    sys.dont_write_bytecode = True
    sys.path[:0] = [_stubs(), REF]
    import torch
    from hqq.core.quantize import HQQLinear, BaseQuantizeConfig
    from hqq.core.bitpack import BitPack
    from hqq.backends.autogptq import patch_hqq_to_gptq
    from hqq.backends.ft import patch_hqq_to_ft, pack_intweight
    import copy

    torch.manual_seed(1234)
    cases = [(2, 128, 512, False), (3, 128, 512, True), (4, 128, 512, False),
             (2, 64, 384, True), (3, 64, 384, False), (4, 64, 384, True)]
    for bits, n, k, with_bias in cases:
        lin = torch.nn.Linear(k, n, bias=with_bias)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(n, k) * 0.02)
            # a few outliers so groups have different ranges
            lin.weight[::7, ::13] *= 4.0
            if with_bias:
                lin.bias.copy_(torch.randn(n) * 0.1)
        lin = lin.half()
        w = lin.weight.data.clone()
        cfg = BaseQuantizeConfig(nbits=bits, group_size=128, axis=1)
        def make():
            hh = HQQLinear(copy.deepcopy(lin), cfg, compute_dtype=torch.float16, device="cpu", del_orig=False)
            hh.name = "golden"
            return hh
        h = make()
        out = {
            "W": w.numpy(), "W_q": h.W_q.numpy(),
            "scale": h.meta["scale"].numpy(), "zero": h.meta["zero"].numpy(),
            "W_deq": h.dequantize().numpy(),
            "nbits": np.int32(bits), "group_size": np.int32(128),
            "shape": np.array([n, k], np.int32),
        }
        assert h.meta["scale"].dtype == torch.float16 and h.meta["packing"] in ("2bit_u8", "3bit_32", "4bit_u8")
        if with_bias:
            out["bias"] = h.bias.detach().numpy()
        x = (torch.randn(3, k) * 1.0).half()
        y = torch.matmul(x, h.dequantize().T)
        if h.bias is not None:
            y = y + h.bias
        out["x"], out["y_ref"] = x.numpy(), y.detach().numpy()

        # reference quirk: GPTQLinear/FT_QuantLinear.__init__ do `if bias:` on the
        # bias *tensor* (autogptq.py:79, ft.py:92) -> RuntimeError for biased
        # layers; only bias-free layers (all Llama linears) can be patched.
        h2 = make()
        assert torch.equal(h2.W_q, h.W_q) and torch.equal(h2.meta["zero"], h.meta["zero"])
        if with_bias:
            h2.bias = None
        g = patch_hqq_to_gptq(h2, None)
        out["gptq_qweight"] = g.qweight.numpy()
        out["gptq_scales"] = g.scales.numpy()
        out["gptq_zeros"] = g.zeros.numpy()
        gx = torch.randn(128, k).half()
        out["gptq_x"] = gx.numpy()
        out["gptq_y"] = g(gx).detach().numpy()          # M=128 -> torch fallback branch

        if bits == 4:
            h3 = make()
            assert torch.equal(h3.W_q, h.W_q) and torch.equal(h3.meta["scale"], h.meta["scale"])
            if with_bias:
                h3.bias = None
            f = patch_hqq_to_ft(h3, None)
            out["awq_qweight"] = f.qweight.numpy()
            out["awq_scales"] = f.scales.numpy()
            out["awq_scaled_zeros"] = f.scaled_zeros.numpy()
        np.savez_compressed(f"{HERE}/hqq_b{bits}_{n}x{k}.npz", **out)
        print("wrote", f"hqq_b{bits}_{n}x{k}.npz")

    # BitPack known-answer captures (reference tests/test_bitpack.py style)
    rng = np.random.default_rng(42)
    bp = {}
    for bits, fn in ((4, BitPack.pack_4bit_u8), (2, BitPack.pack_2bit_u8), (3, BitPack.pack_3bit_32)):
        for rows, cols in ((32, 32), (60, 128), (52, 128)):
            if bits == 4 and rows % 2:
                continue
            if bits == 2 and rows % 4:
                continue
            q = rng.integers(0, 2 ** bits, size=(rows, cols), dtype=np.int64)
            tq = torch.from_numpy(q.astype(np.uint8 if bits != 3 else np.int32))
            bp[f"q_b{bits}_{rows}x{cols}"] = q.astype(np.uint8)
            bp[f"packed_b{bits}_{rows}x{cols}"] = fn(tq).numpy()
    np.savez_compressed(f"{HERE}/bitpack.npz", **bp)
    print("wrote bitpack.npz")

    pi = {}
    for n, k in ((8, 64), (16, 256), (64, 128)):
        q = rng.integers(0, 16, size=(n, k), dtype=np.int64)
        pi[f"q_{n}x{k}"] = q.astype(np.uint8)
        pi[f"packed_{n}x{k}"] = pack_intweight(torch.from_numpy(q.astype(np.int32)), 4, 64).numpy()
    np.savez_compressed(f"{HERE}/pack_intweight.npz", **pi)
    print("wrote pack_intweight.npz")


if __name__ == "__main__":
    main()
