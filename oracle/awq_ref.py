"""Oracle: FT_QuantLinear "Format C" (llm-awq / TinyChat v2 int16 interleave).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in numpy:
  * pack_intweight(q, interleave=4, kstride=64)   (backends/ft.py:15-55)
  * FT_QuantLinear.pack                           (backends/ft.py:103-126)
  * the decode the CUDA GEMV performs: dequantize_s4_to_fp16x2
    (amq/kernel/ft/quantization_new/dequantize.cuh:14-77) followed by the
    register shuffle of gemv_cuda.cu:142-158, and its dequant arithmetic
    ``w = fma(q, s, scaled_zeros)`` (gemv_cuda.cu:151).
There is no CPU decode of Format C anywhere in the reference; the decode here
follows the kernel and is pinned by pack->decode round trips plus the golden
``pack_intweight`` captures.

Format C:  qweight int16 [N/4, K]; scales fp16 [K/G, N] = s;
scaled_zeros fp16 [K/G, N] = -(z*s).

The pack is a pure permutation of 4-bit values.  For one output row n and
input index k, write k = 32*b + i (i in 0..31):
  step 1 (ft.py:21-24)  position within the 32-block:
         i = 8*p + 2*a + e   ->   i1 = 8*a + 2*p + e
  step 2 (ft.py:27-30)  position within each 8:
         i1 = 8*g + 2*c + e  ->   i2 = 8*g + 4*e + c
  step 3 (ft.py:33-41)  rows are grouped by 4 and K by 64: the 256 values of
         (row group, k-chunk) are laid out [r][kk] (r = n%4, kk = position in
         the 64-chunk after steps 1-2) and cut into 64 int16 of 4 nibbles:
         flat v = 64*r + kk; int16 index v//4, nibble v%4 (bit 4*(v%4)).
"""
import numpy as np


def _perm32():
    """dst position (after steps 1+2) for each source index i of a 32-block."""
    dst = np.zeros(32, dtype=np.int64)
    for i in range(32):
        p, a, e = i // 8, (i % 8) // 2, i % 2
        i1 = 8 * a + 2 * p + e
        g, c, e1 = i1 // 8, (i1 % 8) // 2, i1 % 2
        dst[i] = 8 * g + 4 * e1 + c
    return dst


_DST32 = _perm32()


def pack_intweight(q_nk, interleave=4, kstride=64):
    """ft.py:15-55 -- q[N,K] (values 0..15) -> int16 [N/4, K]."""
    assert interleave == 4 and kstride == 64
    q = np.asarray(q_nk).astype(np.uint16)
    n, k = q.shape
    # steps 1+2: permute inside every 32-block
    blk = q.reshape(n, k // 32, 32)
    perm = np.empty_like(blk)
    perm[:, :, _DST32] = blk
    perm = perm.reshape(n, k)
    # step 3: [N/4][K/64] chunks of 4 rows x 64 values, row-major by row
    chunks = perm.reshape(n // 4, 4, k // 64, 64).transpose(0, 2, 1, 3)
    flat = chunks.reshape(n // 4, k // 64, 64, 4)          # [.., int16 idx, nibble]
    packed = (flat[..., 0] | (flat[..., 1] << 4) | (flat[..., 2] << 8)
              | (flat[..., 3] << 12)).astype(np.uint16)
    return packed.reshape(n // 4, k).view(np.int16)


def unpack_intweight(qweight, n, k):
    """Decode Format C as the GEMV kernel does.

    Per (row group of 4, 64-k chunk): 64 int16 = 32 uint32; row r owns uint32
    #8r..8r+7 (= int16 #16r..16r+15).  Each uint32's nibbles n0..n7 are emitted
    by dequantize_s4_to_fp16x2 in the order [n0,n4,n1,n5,n2,n6,n3,n7]
    (dequantize.cuh:31-47: elt_01, elt_23, elt_45, elt_67 of the interleaved
    register).  The resulting 32 halves hw[0..31] (4 uint32 = one thread's
    float4) are then re-ordered by gemv_cuda.cu:142-158:
        out[(i*4 + j)*2 + e] = hw[(i + 4*j)*2 + e],  i,j in 0..3, e in 0..1
    which yields the 32 consecutive k of that block."""
    qw = np.asarray(qweight).view(np.uint16).reshape(n // 4, k // 64, 4, 16)
    # 16 int16 -> 8 uint32 (little endian: int16 #2t is the low half)
    u32 = qw[..., 0::2].astype(np.uint32) | (qw[..., 1::2].astype(np.uint32) << 16)
    nib = np.stack([(u32 >> np.uint32(4 * i)) & np.uint32(15) for i in range(8)], axis=-1)
    hw = nib[..., [0, 4, 1, 5, 2, 6, 3, 7]]                 # [N/4,K/64,4,8(u32),8]
    hw = hw.reshape(n // 4, k // 64, 4, 2, 32)              # two 32-blocks per row
    out = np.empty_like(hw)
    for i in range(4):
        for j in range(4):
            for e in range(2):
                out[..., (i * 4 + j) * 2 + e] = hw[..., (i + 4 * j) * 2 + e]
    out = out.reshape(n // 4, k // 64, 4, 64).transpose(0, 2, 1, 3)
    return np.ascontiguousarray(out.reshape(n, k)).astype(np.uint8)


def pack(w_deq, scales, zeros, group_size=128):
    """FT_QuantLinear.pack (ft.py:103-126), sym=False.
    Returns (qweight int16 [N/4,K], scales fp16 [K/G,N], scaled_zeros fp16)."""
    from .gptq_ref import recover_int
    q, sz = recover_int(w_deq, scales, zeros, group_size)
    qweight = pack_intweight(q)
    sc = np.ascontiguousarray(np.asarray(scales, np.float16).T)
    szt = (-np.ascontiguousarray(sz.T)).astype(np.float16)
    return qweight, sc, szt


def dequant_kernel(qweight, scales, scaled_zeros, group_size=128):
    """Weight as gemv_kernel forms it (gemv_cuda.cu:151):
    ``__hfma2(w, scale, scaled_zeros)`` -> fp16 [N,K], one fused rounding."""
    kg, n = np.asarray(scales).shape
    k = kg * group_size
    q = unpack_intweight(qweight, n, k).astype(np.float64)
    s = np.repeat(np.asarray(scales, np.float16).T.astype(np.float64), group_size, axis=1)
    z = np.repeat(np.asarray(scaled_zeros, np.float16).T.astype(np.float64), group_size, axis=1)
    return (q * s + z).astype(np.float16)
