"""Oracle: GPTQLinear "Format B" (AutoGPTQ cuda-old int32 pack).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in numpy:
  * GPTQLinear.pack                       (backends/autogptq.py:111-156)
  * the torch fallback branch of forward  (backends/autogptq.py:245-283)
  * the dequant arithmetic of VecQuant{2,3,4}MatMulKernelFaster_old
    (amq/kernel/AutoGPTQ/auto_gptq_kernel.cu:197-218): w = fma(q, s, -zeros)

Format B:  qweight int32 [K/32*bits, N]; scales fp32 [K/G, N] = s;
zeros fp32 [K/G, N] = fp16(z*s) widened.  Bit order along K (column n):
  2-bit: (qweight[k//16, n] >> 2*(k%16)) & 3
  4-bit: (qweight[k//8,  n] >> 4*(k%8))  & 15
  3-bit: 32 values per 3 words r0,r1,r2 (autogptq.py:133-151):
     j<10 : (r0 >> 3j) & 7
     j=10 : (r0 >> 30) | ((r1 & 1) << 2)
     11..20: (r1 >> (3(j-11)+1)) & 7
     j=21 : (r1 >> 31) | ((r2 & 3) << 1)
     22..31: (r2 >> (3(j-22)+2)) & 7
"""
import numpy as np


def recover_int(w_deq, scales, zeros, group_size=128):
    """autogptq.py:112-120: intweight = round((W + z*s) / s), all in fp16.

    ``scales``/``zeros`` are the HQQ meta reshaped to [N, K/G] (fp16).
    Returns (q[N,K] int32, scale_zeros[N,K/G] fp16)."""
    w = np.asarray(w_deq, dtype=np.float16)
    s = np.asarray(scales, dtype=np.float16)
    z = np.asarray(zeros, dtype=np.float16)
    sz = (z * s).astype(np.float16)
    s_rep = np.repeat(s, group_size, axis=1)
    sz_rep = np.repeat(sz, group_size, axis=1)
    t = (w + sz_rep).astype(np.float16)
    t = (t / s_rep).astype(np.float16)
    q = np.rint(t.astype(np.float32)).astype(np.int32)   # torch.round = half-to-even
    return q, sz


def pack_qweight(q_nk, bits):
    """autogptq.py:121-156 -- q[N,K] -> qweight int32 [K/32*bits, N]."""
    q = np.ascontiguousarray(np.asarray(q_nk).T).astype(np.uint32)   # [K, N]
    k, n = q.shape
    out = np.zeros((k // 32 * bits, n), dtype=np.uint32)
    if bits in (2, 4, 8):
        per = 32 // bits
        for j in range(per):
            out |= q[j::per] << np.uint32(bits * j)
    elif bits == 3:
        blk = q.reshape(k // 32, 32, n)
        r0 = np.zeros((k // 32, n), np.uint32)
        r1 = np.zeros_like(r0)
        r2 = np.zeros_like(r0)
        for j in range(10):
            r0 |= blk[:, j] << np.uint32(3 * j)
        r0 |= blk[:, 10] << np.uint32(30)
        r1 |= (blk[:, 10] >> np.uint32(2)) & np.uint32(1)
        for j in range(11, 21):
            r1 |= blk[:, j] << np.uint32(3 * (j - 11) + 1)
        r1 |= blk[:, 21] << np.uint32(31)
        r2 |= (blk[:, 21] >> np.uint32(1)) & np.uint32(3)
        for j in range(22, 32):
            r2 |= blk[:, j] << np.uint32(3 * (j - 22) + 2)
        out = np.stack([r0, r1, r2], axis=1).reshape(k // 32 * 3, n)
    else:
        raise NotImplementedError(bits)
    return out.astype(np.uint32).view(np.int32)


def pack(w_deq, scales, zeros, bits, group_size=128):
    """GPTQLinear.pack (autogptq.py:111-156).  Returns the three buffers
    (qweight int32 [K/32*bits,N], scales fp32 [K/G,N], zeros fp32 [K/G,N])."""
    q, sz = recover_int(w_deq, scales, zeros, group_size)
    qweight = pack_qweight(q, bits)
    sc = np.ascontiguousarray(np.asarray(scales, np.float16).T).astype(np.float32)
    zr = np.ascontiguousarray(sz.T).astype(np.float32)
    return qweight, sc, zr


def unpack_qweight(qweight, bits):
    """Inverse of pack_qweight -> q[N,K] uint8 (the bit layout in the module
    docstring; matches autogptq.py:249-277)."""
    qw = np.asarray(qweight).view(np.uint32)
    rows, n = qw.shape
    k = rows * 32 // bits
    q = np.zeros((k, n), dtype=np.uint32)
    if bits in (2, 4, 8):
        per = 32 // bits
        mask = np.uint32((1 << bits) - 1)
        for j in range(per):
            q[j::per] = (qw >> np.uint32(bits * j)) & mask
    elif bits == 3:
        w3 = qw.reshape(rows // 3, 3, n)
        r0, r1, r2 = w3[:, 0], w3[:, 1], w3[:, 2]
        blk = np.zeros((rows // 3, 32, n), dtype=np.uint32)
        for j in range(10):
            blk[:, j] = (r0 >> np.uint32(3 * j)) & np.uint32(7)
        blk[:, 10] = (r0 >> np.uint32(30)) | ((r1 & np.uint32(1)) << np.uint32(2))
        for j in range(11, 21):
            blk[:, j] = (r1 >> np.uint32(3 * (j - 11) + 1)) & np.uint32(7)
        blk[:, 21] = (r1 >> np.uint32(31)) | ((r2 & np.uint32(3)) << np.uint32(1))
        for j in range(22, 32):
            blk[:, j] = (r2 >> np.uint32(3 * (j - 22) + 2)) & np.uint32(7)
        q = blk.reshape(k, n)
    else:
        raise NotImplementedError(bits)
    return np.ascontiguousarray(q.T).astype(np.uint8)


def dequant_fallback(qweight, scales, zeros, bits, group_size=128):
    """Weight of the torch fallback branch (autogptq.py:249-282):
    ``weight = scales.half() * q - zeros.half()`` -> fp16 [K, N]
    (fp16 multiply, then fp16 subtract: two roundings)."""
    q = unpack_qweight(qweight, bits).T.astype(np.float16)          # [K, N]
    s = np.repeat(np.asarray(scales, np.float32).astype(np.float16), group_size, axis=0)
    z = np.repeat(np.asarray(zeros, np.float32).astype(np.float16), group_size, axis=0)
    return ((s * q).astype(np.float16) - z).astype(np.float16)


def dequant_kernel(qweight, scales, zeros, bits, group_size=128):
    """Weight as the CUDA kernels form it (auto_gptq_kernel.cu:197-218):
    ``hfma2(lut(q), half2(scale), half2(-zero))`` = one fused fp16 rounding.
    Returns fp16 [N, K] (nn.Linear orientation)."""
    q = unpack_qweight(qweight, bits).astype(np.float64)            # [N, K]
    s = np.repeat(np.asarray(scales, np.float32).astype(np.float16).T, group_size, axis=1)
    z = np.repeat(np.asarray(zeros, np.float32).astype(np.float16).T, group_size, axis=1)
    # products/sums of fp16 values are exact in fp64 -> single rounding to fp16
    return (q * s.astype(np.float64) - z.astype(np.float64)).astype(np.float16)


def forward_fallback(x, qweight, scales, zeros, bits, group_size=128, bias=None):
    """GPTQLinear.forward, M >= kernel_switch_threshold branch
    (autogptq.py:245-287): fp16 matmul(x, weight[K,N]) (+ bias)."""
    from .linear_ref import matmul_f16
    w_kn = dequant_fallback(qweight, scales, zeros, bits, group_size)
    y = matmul_f16(np.asarray(x, np.float16).reshape(-1, w_kn.shape[0]), w_kn)
    if bias is not None:
        y = (y + np.asarray(bias, np.float16)).astype(np.float16)
    return y
