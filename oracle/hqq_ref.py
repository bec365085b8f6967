"""Oracle: HQQ on-disk weight format ("Format A") -- bit packing and dequantize.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in numpy:
  * BitPack.pack/unpack_{4bit_u8,2bit_u8,3bit_32}  (core/bitpack.py:24-110)
  * Quantizer.dequantize                            (core/quantize.py:184-199)
  * the axis=1 grouping of Quantizer.quantize       (core/quantize.py:106-111)

Layout facts being restated (axis=1, group_size=G):
  W[N,K] is viewed as Wg[R,G] with R = N*K/G (row-major reshape), each row is
  one quantization group with its own fp16 ``scale`` (the *dequant multiplier*,
  quantize.py:154) and fp16 ``zero``.  The packers split Wg's rows into
  equal "row chunks" and put chunk c at a fixed bit offset of every word.
"""
import numpy as np

PACKING = {4: "4bit_u8", 3: "3bit_32", 2: "2bit_u8"}


# ----------------------------------------------------------------- packing
def pack_4bit_u8(wg):
    """core/bitpack.py:24-29 -- uint8[R,G] -> uint8[R/2,G]; first half in the
    high nibble, second half in the low nibble."""
    wg = np.asarray(wg).astype(np.uint8)
    step = wg.shape[0] // 2
    return ((wg[:step] << 4) | wg[step:2 * step]).astype(np.uint8)


def unpack_4bit_u8(wq):
    """core/bitpack.py:31-39."""
    wq = np.asarray(wq).astype(np.uint8)
    return np.concatenate([(wq & 0xF0) >> 4, wq & 0x0F], axis=0)


def pack_2bit_u8(wg):
    """core/bitpack.py:43-53 -- four row-quarters at shifts 6,4,2,0."""
    wg = np.asarray(wg).astype(np.uint8)
    s = wg.shape[0] // 4
    return ((wg[:s] << 6) | (wg[s:2 * s] << 4) | (wg[2 * s:3 * s] << 2)
            | wg[3 * s:4 * s]).astype(np.uint8)


def unpack_2bit_u8(wq):
    """core/bitpack.py:55-65."""
    wq = np.asarray(wq).astype(np.uint8)
    return np.concatenate([(wq >> 6) & 3, (wq >> 4) & 3, (wq >> 2) & 3, wq & 3],
                          axis=0)


def pack_3bit_32(wg):
    """core/bitpack.py:69-92 -- rows zero-padded to a multiple of 10, ten
    row-chunks at shifts 27,24,...,0 of an int32 word."""
    wg = np.asarray(wg)
    rows = int(10 * np.ceil(wg.shape[0] / 10.0))
    pad = np.zeros((rows, wg.shape[1]), dtype=np.int32)
    pad[:wg.shape[0]] = wg
    s = rows // 10
    out = np.zeros((s, wg.shape[1]), dtype=np.int32)
    for c in range(10):
        out |= pad[c * s:(c + 1) * s] << (27 - 3 * c)
    return out


def unpack_3bit_32(wq):
    """core/bitpack.py:95-110 -- returns all 10*step rows (caller slices)."""
    wq = np.asarray(wq).astype(np.int32)
    return np.concatenate([(wq >> (27 - 3 * c)) & 7 for c in range(10)],
                          axis=0).astype(np.uint8)


_PACK = {4: pack_4bit_u8, 3: pack_3bit_32, 2: pack_2bit_u8}
_UNPACK = {4: unpack_4bit_u8, 3: unpack_3bit_32, 2: unpack_2bit_u8}


def pack(q_nk, nbits, group_size=128):
    """q[N,K] integers -> HQQ ``W_q`` (quantize.py:106-111 reshape + the
    packer selected by Quantizer.bit_to_packing, quantize.py:40-49)."""
    q = np.asarray(q_nk)
    wg = q.reshape(-1, group_size)
    return _PACK[nbits](wg)


def unpack(wq, nbits, shape, group_size=128):
    """HQQ ``W_q`` -> q[N,K] uint8.  3-bit rows are sliced back to R rows
    exactly as Quantizer.dequantize does (quantize.py:190-195)."""
    n, k = shape
    r = n * k // group_size
    wg = _UNPACK[nbits](wq)[:r]
    return wg.reshape(n, k)


# -------------------------------------------------------------- dequantize
def dequantize(wq, scale, zero, nbits, shape, group_size=128):
    """Quantizer.dequantize (quantize.py:184-199), compute_dtype = fp16:

        W_r = unpack(W_q).to(fp16)[:R]
        W   = ((W_r - zero) * scale).reshape(shape)      # two fp16 roundings

    ``scale``/``zero`` are fp16 [R,1].  numpy float16 arithmetic rounds each
    elementwise op to fp16 (via fp32), which is what torch's CPU Half kernels
    do as well, so this is bit-identical to the reference on CPU.
    """
    n, k = shape
    r = n * k // group_size
    wr = _UNPACK[nbits](wq)[:r].astype(np.float16)
    scale = np.asarray(scale, dtype=np.float16).reshape(r, 1)
    zero = np.asarray(zero, dtype=np.float16).reshape(r, 1)
    d = (wr - zero).astype(np.float16)
    w = (d * scale).astype(np.float16)
    return w.reshape(n, k)


def dequantize_from_q(q_nk, scale, zero, group_size=128):
    """Same arithmetic starting from already-unpacked integers q[N,K]."""
    n, k = q_nk.shape
    r = n * k // group_size
    wr = np.asarray(q_nk).reshape(r, group_size).astype(np.float16)
    scale = np.asarray(scale, dtype=np.float16).reshape(r, 1)
    zero = np.asarray(zero, dtype=np.float16).reshape(r, 1)
    return ((wr - zero).astype(np.float16) * scale).astype(np.float16).reshape(n, k)


# ------------------------------------------------- bfloat16 compute dtype
# HQQLinear(compute_dtype=torch.bfloat16) keeps scale / zero in bf16 and runs
# the same two-op dequantize in bf16 (quantize.py:184-199, 396-407, 516).
# numpy has no bfloat16: values travel as uint16 bit patterns; each torch bf16
# op is "widen to fp32, operate, round to nearest even", restated below.
def bf16_bits_to_f32(bits):
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x):
    """round-to-nearest-even fp32 -> bf16 bit patterns (finite inputs; NaN kept quiet)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    rounded = ((u + (np.uint32(0x7FFF) + ((u >> 16) & 1))) >> 16).astype(np.uint16)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    return np.where(nan, np.uint16(0x7FC0), rounded).astype(np.uint16)


def dequantize_bf16(wq, scale_bits, zero_bits, nbits, shape, group_size=128):
    """Quantizer.dequantize with compute_dtype = bfloat16 -> bf16 bits [N,K]:
    W = bf16(bf16(W_r - zero) * scale), scale / zero given as bf16 bits [R,1]."""
    n, k = shape
    r = n * k // group_size
    wr = _UNPACK[nbits](wq)[:r].astype(np.float32)
    s = bf16_bits_to_f32(scale_bits).reshape(r, 1)
    z = bf16_bits_to_f32(zero_bits).reshape(r, 1)
    d = bf16_bits_to_f32(f32_to_bf16_bits(wr - z))
    return f32_to_bf16_bits(d * s).reshape(n, k)
