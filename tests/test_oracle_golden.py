"""Pin the CPU oracle against golden vectors captured from the real reference
(tests/golden/gen_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import hqq_ref, gptq_ref, awq_ref, linear_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "hqq_b*.npz")))


def _load(path):
    d = np.load(path)
    return {k: d[k] for k in d.files}


def test_fixture_inventory():
    assert len(CASES) == 6
    assert os.path.exists(os.path.join(GOLDEN, "bitpack.npz"))
    assert os.path.exists(os.path.join(GOLDEN, "pack_intweight.npz"))


def test_bitpack_known_answers():
    """reference tests/test_bitpack.py semantics: pack == reference, unpack(pack(q)) == q."""
    d = _load(os.path.join(GOLDEN, "bitpack.npz"))
    seen = 0
    for key in d:
        if not key.startswith("q_b"):
            continue
        bits = int(key[3])
        tag = key[2:]
        q, packed = d[key], d["packed_" + tag]
        mine = {4: hqq_ref.pack_4bit_u8, 2: hqq_ref.pack_2bit_u8, 3: hqq_ref.pack_3bit_32}[bits](q)
        assert mine.dtype == packed.dtype and np.array_equal(mine, packed), key
        un = {4: hqq_ref.unpack_4bit_u8, 2: hqq_ref.unpack_2bit_u8, 3: hqq_ref.unpack_3bit_32}[bits](packed)
        assert np.array_equal(un[: q.shape[0]], q), key
        seen += 1
    assert seen >= 7


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_hqq_dequantize_bit_exact(path):
    g = _load(path)
    bits, shape = int(g["nbits"]), tuple(int(v) for v in g["shape"])
    w = hqq_ref.dequantize(g["W_q"], g["scale"], g["zero"], bits, shape, 128)
    assert w.dtype == np.float16
    assert np.array_equal(w.view(np.uint16), g["W_deq"].view(np.uint16))
    # pack(unpack(W_q)) reproduces the payload
    q = hqq_ref.unpack(g["W_q"], bits, shape, 128)
    assert q.max() < 2 ** bits
    assert np.array_equal(hqq_ref.pack(q, bits, 128), g["W_q"])
    assert np.array_equal(hqq_ref.dequantize_from_q(q, g["scale"], g["zero"]).view(np.uint16),
                          g["W_deq"].view(np.uint16))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_reference_forward(path):
    """y = matmul(x, W_deq.T) (+bias): fp32-accumulate, one fp16 rounding."""
    g = _load(path)
    bits, shape = int(g["nbits"]), tuple(int(v) for v in g["shape"])
    y = linear_ref.hqq_forward(g["x"], g["W_q"], g["scale"], g["zero"], bits, shape, 128, g.get("bias"))
    ref = g["y_ref"].astype(np.float32)
    err = np.abs(y.astype(np.float32) - ref)
    # same math up to fp32 summation order: at most 1 fp16 ulp of the output
    assert np.all(err <= 1e-3 * np.abs(ref) + 1e-4), err.max()


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_gptq_pack_bit_exact(path):
    g = _load(path)
    bits, (n, k) = int(g["nbits"]), tuple(int(v) for v in g["shape"])
    sc = g["scale"].reshape(n, -1)
    zr = g["zero"].reshape(n, -1)
    qweight, scales, zeros = gptq_ref.pack(g["W_deq"], sc, zr, bits, 128)
    assert qweight.dtype == np.int32 and qweight.shape == g["gptq_qweight"].shape
    assert np.array_equal(qweight, g["gptq_qweight"])
    assert np.array_equal(scales, g["gptq_scales"]) and scales.dtype == np.float32
    assert np.array_equal(zeros, g["gptq_zeros"]) and zeros.dtype == np.float32
    # the recovered integers are the original HQQ integers
    q = gptq_ref.unpack_qweight(qweight, bits)
    assert np.array_equal(q, hqq_ref.unpack(g["W_q"], bits, (n, k), 128))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_gptq_fallback_forward(path):
    g = _load(path)
    bits = int(g["nbits"])
    y = gptq_ref.forward_fallback(g["gptq_x"], g["gptq_qweight"], g["gptq_scales"], g["gptq_zeros"], bits, 128)
    ref = g["gptq_y"].astype(np.float32)
    err = np.abs(y.astype(np.float32) - ref)
    assert np.all(err <= 1e-3 * np.abs(ref) + 2e-4), err.max()
    # the kernel-form weight (single fma rounding) stays within fp16 rounding of the fallback weight
    wk = gptq_ref.dequant_kernel(g["gptq_qweight"], g["gptq_scales"], g["gptq_zeros"], bits, 128)
    wf = gptq_ref.dequant_fallback(g["gptq_qweight"], g["gptq_scales"], g["gptq_zeros"], bits, 128).T
    assert np.max(np.abs(wk.astype(np.float32) - wf.astype(np.float32))) <= 2.0 ** -10 * np.max(np.abs(wf.astype(np.float32)))


def test_pack_intweight_known_answers():
    d = _load(os.path.join(GOLDEN, "pack_intweight.npz"))
    seen = 0
    for key in d:
        if not key.startswith("q_"):
            continue
        q, packed = d[key], d["packed_" + key[2:]]
        mine = awq_ref.pack_intweight(q)
        assert mine.dtype == np.int16 and np.array_equal(mine, packed), key
        assert np.array_equal(awq_ref.unpack_intweight(packed, *q.shape), q), key
        seen += 1
    assert seen == 3


@pytest.mark.parametrize("path", [c for c in CASES if "_b4_" in c], ids=lambda c: os.path.basename(c))
def test_awq_pack_bit_exact(path):
    g = _load(path)
    n, k = (int(v) for v in g["shape"])
    qweight, scales, szeros = awq_ref.pack(g["W_deq"], g["scale"].reshape(n, -1), g["zero"].reshape(n, -1), 128)
    assert np.array_equal(qweight, g["awq_qweight"])
    assert np.array_equal(scales.view(np.uint16), g["awq_scales"].view(np.uint16))
    assert np.array_equal(szeros.view(np.uint16), g["awq_scaled_zeros"].view(np.uint16))
    q = awq_ref.unpack_intweight(qweight, n, k)
    assert np.array_equal(q, hqq_ref.unpack(g["W_q"], 4, (n, k), 128))
    # kernel-form weight is within fp16 rounding of the HQQ weight
    wk = awq_ref.dequant_kernel(qweight, scales, szeros, 128).astype(np.float32)
    wd = g["W_deq"].astype(np.float32)
    assert np.max(np.abs(wk - wd)) <= 2.0 ** -9 * np.max(np.abs(wd))


# ---- group sizes other than 128 (256; 64 and 32), captured from the real reference by tests/golden/gen_golden_groups.py
GCASES = sorted(glob.glob(os.path.join(GOLDEN, "hqq_g*_b*.npz")))


def test_group_fixture_inventory():
    assert len(GCASES) == 9


@pytest.mark.parametrize("path", GCASES, ids=[os.path.basename(c) for c in GCASES])
def test_coarser_group_oracle_matches_reference(path):
    """the oracle's group_size parameter against the reference at groups 256, 64 and 32: Format A dequant and pack / unpack bit-exact (the
    packing geometry depends on the group), forward within one fp16 ulp, GPTQ pack bit-exact and its fallback forward"""
    g = _load(path)
    bits, (n, k), G = int(g["nbits"]), tuple(int(v) for v in g["shape"]), int(g["group_size"])
    assert G in (256, 64, 32)
    w = hqq_ref.dequantize(g["W_q"], g["scale"], g["zero"], bits, (n, k), G)
    assert np.array_equal(w.view(np.uint16), g["W_deq"].view(np.uint16))
    q = hqq_ref.unpack(g["W_q"], bits, (n, k), G)
    assert np.array_equal(hqq_ref.pack(q, bits, G), g["W_q"])
    y = linear_ref.hqq_forward(g["x"], g["W_q"], g["scale"], g["zero"], bits, (n, k), G, None)
    ref = g["y_ref"].astype(np.float32)
    assert np.all(np.abs(y.astype(np.float32) - ref) <= 1e-3 * np.abs(ref) + 1e-4)
    qweight, scales, zeros = gptq_ref.pack(g["W_deq"], g["scale"].reshape(n, -1), g["zero"].reshape(n, -1), bits, G)
    assert np.array_equal(qweight, g["gptq_qweight"]) and np.array_equal(scales, g["gptq_scales"]) and np.array_equal(zeros, g["gptq_zeros"])
    yf = gptq_ref.forward_fallback(g["gptq_x"], g["gptq_qweight"], g["gptq_scales"], g["gptq_zeros"], bits, G)
    rf = g["gptq_y"].astype(np.float32)
    assert np.all(np.abs(yf.astype(np.float32) - rf) <= 1e-3 * np.abs(rf) + 2e-4)


# ---- bfloat16 compute dtype (tests/golden/gen_golden_bf16.py: HQQLinear(compute_dtype=torch.bfloat16) on CPU)
BF16_CASES = sorted(glob.glob(os.path.join(GOLDEN, "bf16_b*.npz")))


def bf16_close(y_bits, ref_bits, ulps=1.0, floor=2.0 ** -8):
    """|y - ref| <= ulps * 2^-7 * |ref| + floor * rms(ref): one bf16 ulp of the output (bf16 keeps 8 significant bits, so the
    1e-3 bar of the fp16 path is below its rounding step) plus a floor for outputs that cancel."""
    y, ref = hqq_ref.bf16_bits_to_f32(y_bits).astype(np.float64), hqq_ref.bf16_bits_to_f32(ref_bits).astype(np.float64)
    bar = ulps * 2.0 ** -7 * np.abs(ref) + floor * np.sqrt(np.mean(ref ** 2))
    return bool(np.all(np.abs(y - ref) <= bar)), float(np.max(np.abs(y - ref) / bar))


def test_bf16_fixture_inventory():
    assert len(BF16_CASES) == 6


def test_bf16_rounding_helper():
    import torch
    x = torch.randn(4096, dtype=torch.float32) * torch.logspace(-8, 8, 4096)
    x[:4] = torch.tensor([1.00390625, 1.01171875, -1.00390625, 3.0e38])      # ties: to even both ways; near the top of the range
    got = hqq_ref.f32_to_bf16_bits(x.numpy())
    want = x.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(got, want)
    assert np.array_equal(hqq_ref.bf16_bits_to_f32(want), x.to(torch.bfloat16).float().numpy())


@pytest.mark.parametrize("path", BF16_CASES, ids=[os.path.basename(c) for c in BF16_CASES])
def test_bf16_dequantize_bit_exact(path):
    g = _load(path)
    bits, shape = int(g["nbits"]), tuple(int(v) for v in g["shape"])
    w = hqq_ref.dequantize_bf16(g["W_q"], g["scale"], g["zero"], bits, shape, 128)
    assert w.dtype == np.uint16 and np.array_equal(w, g["W_deq"])


@pytest.mark.parametrize("path", BF16_CASES, ids=[os.path.basename(c) for c in BF16_CASES])
def test_bf16_reference_forward(path):
    g = _load(path)
    for tag in ("", "16"):
        y = linear_ref.linear_bf16(g["x" + tag], g["W_deq"], g.get("bias"))
        ok, worst = bf16_close(y, g["y" + tag + "_ref"])
        assert ok, (tag, worst)
