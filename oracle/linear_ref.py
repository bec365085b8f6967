"""Oracle: the reference's CPU forward -- nn.Linear on dequantized weights.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates ``forward_hqq_inferece`` (utils/patching.py:95-100):
    out = torch.matmul(x, self.dequantize().T) (+ bias)
with fp16 inputs.  torch's CPU Half matmul accumulates in fp32 and rounds the
result to fp16 once; ``matmul_f16`` does the same with numpy (fp32 GEMM of the
exactly-representable fp16 operands, one rounding to fp16).  Summation order
inside an fp32 GEMM is implementation-defined, so parity with the reference is
"to fp16 output rounding" (tests use rtol 1e-3), not bitwise.
"""
import numpy as np


def matmul_f16(a, b):
    """fp16 [M,K] @ fp16 [K,N] -> fp16 [M,N], fp32 accumulate, one rounding."""
    a32 = np.asarray(a, dtype=np.float16).astype(np.float32)
    b32 = np.asarray(b, dtype=np.float16).astype(np.float32)
    return (a32 @ b32).astype(np.float16)


def matmul_f64(a, b):
    """Exact-ish reference (fp64 accumulate), returned as fp64 -- used to bound
    the error of both the oracle and the HIP kernels in tests."""
    return np.asarray(a, np.float16).astype(np.float64) @ np.asarray(b, np.float16).astype(np.float64)


def linear_f16(x, w_deq, bias=None):
    """F.linear(x, W_deq, bias) in fp16: x[..., K], W_deq[N, K] -> [..., N]."""
    x = np.asarray(x, dtype=np.float16)
    lead = x.shape[:-1]
    y = matmul_f16(x.reshape(-1, x.shape[-1]), np.asarray(w_deq, np.float16).T)
    if bias is not None:
        y = (y + np.asarray(bias, np.float16)).astype(np.float16)
    return y.reshape(*lead, -1)


def hqq_forward(x, wq, scale, zero, nbits, shape, group_size=128, bias=None):
    """The full reference CPU path: dequantize-every-call + matmul
    (patching.py:95-100 with quantize.py:184-199)."""
    from .hqq_ref import dequantize
    return linear_f16(x, dequantize(wq, scale, zero, nbits, shape, group_size), bias)


def linear_bf16(x_bits, w_bits, bias_bits=None):
    """F.linear on bfloat16 tensors given as uint16 bit patterns: fp32 accumulate,
    one rounding of y to bf16, bias as a separate bf16 add (what torch's CPU
    bf16 matmul + add do; summation order is implementation-defined, so parity
    is to bf16 output rounding, not bitwise)."""
    from .hqq_ref import bf16_bits_to_f32, f32_to_bf16_bits
    x = bf16_bits_to_f32(x_bits)
    w = bf16_bits_to_f32(w_bits)
    y = f32_to_bf16_bits(x.reshape(-1, x.shape[-1]) @ w.T)
    if bias_bits is not None:
        y = f32_to_bf16_bits(bf16_bits_to_f32(y) + bf16_bits_to_f32(bias_bits))
    return y.reshape(*x.shape[:-1], -1)
