#!/usr/bin/env python3
"""Golden vectors for the bfloat16 entry points, from the REAL reference on CPU.

Run only in the build container (needs /root/reference); same recipe as
gen_golden.py (the vendored HQQ package + two stub modules).  HQQLinear is
constructed with compute_dtype=torch.bfloat16: scale / zero are then bf16 and
Quantizer.dequantize runs in bf16 (hqq/core/quantize.py:184-199, 516).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_bf16.py

Per case ``bf16_b{bits}_{N}x{K}.npz`` (bf16 tensors stored as their uint16
bit patterns -- numpy has no bfloat16):
  W_q              HQQLinear.W_q (Format A payload)
  scale, zero      HQQLinear.meta[...]     bf16 bits [N*K/G, 1]
  W_deq            HQQLinear.dequantize()  bf16 bits [N,K]     <- the parity weight
  x, y_ref         x[3,K] bf16 bits and torch.matmul(x, W_deq.T) (+ bias) on CPU, bf16 bits
  x16, y16_ref     x[16,K] / y (the 16-row case of the few-row kernel)
  bias             bf16 bits [N] or absent
"""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import REF, _stubs  # noqa: E402


def bits_of(t):
    import torch
    assert t.dtype == torch.bfloat16
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def main():
    sys.dont_write_bytecode = True
    sys.path[:0] = [_stubs(), REF]
    import torch
    from hqq.core.quantize import HQQLinear, BaseQuantizeConfig

    torch.manual_seed(4321)
    cases = [(2, 128, 512, False), (3, 128, 512, True), (4, 128, 512, False),
             (2, 64, 384, True), (3, 64, 384, False), (4, 64, 384, True)]
    for bits, n, k, with_bias in cases:
        lin = torch.nn.Linear(k, n, bias=with_bias)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(n, k) * 0.02)
            lin.weight[::7, ::13] *= 4.0
            if with_bias:
                lin.bias.copy_(torch.randn(n) * 0.1)
        lin = lin.to(torch.bfloat16)
        cfg = BaseQuantizeConfig(nbits=bits, group_size=128, axis=1)
        h = HQQLinear(copy.deepcopy(lin), cfg, compute_dtype=torch.bfloat16, device="cpu", del_orig=False)
        assert h.meta["scale"].dtype == torch.bfloat16 and h.meta["zero"].dtype == torch.bfloat16
        wd = h.dequantize()
        assert wd.dtype == torch.bfloat16
        out = {"W_q": h.W_q.numpy(), "scale": bits_of(h.meta["scale"]), "zero": bits_of(h.meta["zero"]),
               "W_deq": bits_of(wd), "nbits": np.int32(bits), "group_size": np.int32(128),
               "shape": np.array([n, k], np.int32)}
        if with_bias:
            out["bias"] = bits_of(h.bias.detach())
        for name, rows in (("", 3), ("16", 16)):
            x = torch.randn(rows, k).to(torch.bfloat16)
            y = torch.matmul(x, wd.T)
            if h.bias is not None:
                y = y + h.bias
            out["x" + name], out["y" + name + "_ref"] = bits_of(x), bits_of(y.detach())
        np.savez_compressed(f"{HERE}/bf16_b{bits}_{n}x{k}.npz", **out)
        print("wrote", f"bf16_b{bits}_{n}x{k}.npz")


if __name__ == "__main__":
    main()
