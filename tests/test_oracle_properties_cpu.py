"""Size-independent properties of the oracle's packing restatements (hypothesis): pack -> unpack round trips of the three weight
formats the reference keeps (HQQ Format A bitpack.py:23-110, GPTQ int32 autogptq.py:111-156, AWQ int16 ft.py:15-55) on random shapes
and contents, and linearity / group structure of the dequantizers.  The golden vectors (test_oracle_golden.py) pin the restatements to the
reference at fixed sizes; these properties extend that to ragged sizes the fixtures do not hold."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import awq_ref, gptq_ref, hqq_ref

SET = dict(max_examples=25, deadline=None)


@settings(**SET)
@given(bits=st.sampled_from([2, 3, 4]), n=st.integers(1, 9).map(lambda v: 16 * v), g=st.integers(1, 5), seed=st.integers(0, 2 ** 31 - 1))
def test_hqq_pack_unpack_roundtrip(bits, n, g, seed):
    k = 128 * g
    q = np.random.default_rng(seed).integers(0, 2 ** bits, size=(n, k), dtype=np.uint8)
    wq = hqq_ref.pack(q, bits)
    assert np.array_equal(hqq_ref.unpack(wq, bits, (n, k)), q)
    # the dequantizer is affine in the integers, per (row, group): (q - z) * s with two fp16 roundings
    rng = np.random.default_rng(seed + 1)
    s = (rng.random((n * g, 1), dtype=np.float32) * 0.02 + 0.001).astype(np.float16)
    z = (rng.random((n * g, 1), dtype=np.float32) * (2 ** bits - 1)).astype(np.float16)
    w = hqq_ref.dequantize(wq, s, z, bits, (n, k))
    want = ((q.reshape(-1, 128).astype(np.float16) - z).astype(np.float16) * s).astype(np.float16).reshape(n, k)
    assert np.array_equal(w.view(np.uint16), want.view(np.uint16))
    assert np.array_equal(hqq_ref.dequantize_from_q(q, s, z).view(np.uint16), w.view(np.uint16))


@settings(**SET)
@given(bits=st.sampled_from([2, 3, 4]), n=st.integers(1, 6).map(lambda v: 32 * v), g=st.integers(1, 4), seed=st.integers(0, 2 ** 31 - 1))
def test_gptq_pack_unpack_roundtrip(bits, n, g, seed):
    k = 128 * g
    q = np.random.default_rng(seed).integers(0, 2 ** bits, size=(n, k)).astype(np.int32)
    qweight = gptq_ref.pack_qweight(q, bits)
    assert qweight.shape == (k // 32 * bits, n) and qweight.dtype == np.int32
    assert np.array_equal(gptq_ref.unpack_qweight(qweight, bits), q)


@settings(**SET)
@given(n=st.integers(1, 6).map(lambda v: 64 * v), g=st.integers(1, 4), seed=st.integers(0, 2 ** 31 - 1))
def test_awq_pack_unpack_roundtrip(n, g, seed):
    k = 128 * g
    q = np.random.default_rng(seed).integers(0, 16, size=(n, k)).astype(np.int32)
    packed = awq_ref.pack_intweight(q)
    assert packed.shape == (n // 4, k) and packed.dtype == np.int16
    assert np.array_equal(awq_ref.unpack_intweight(packed, n, k), q)


@settings(**SET)
@given(bits=st.sampled_from([2, 3, 4]), seed=st.integers(0, 2 ** 31 - 1))
def test_gptq_recovers_the_integers_it_was_given(bits, seed):
    """GPTQLinear.pack recovers q = round((W + z s) / s) from the dequantized weights (autogptq.py:120): exact for weights that
    came from integers, whatever the (positive) scale"""
    n, k = 32, 256
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 2 ** bits, size=(n, k)).astype(np.float32)
    s = (rng.random((n, k // 128), dtype=np.float32) * 0.05 + 0.002).astype(np.float16).astype(np.float32)
    z = rng.integers(0, 2 ** bits, size=(n, k // 128)).astype(np.float32)
    w = ((q - np.repeat(z, 128, 1)).astype(np.float16) * np.repeat(s, 128, 1).astype(np.float16)).astype(np.float16)   # HQQ's two roundings
    got, sz = gptq_ref.recover_int(w, s, z)
    assert np.array_equal(got, q.astype(got.dtype))
    assert np.array_equal(sz.view(np.uint16), (z.astype(np.float16) * s.astype(np.float16)).astype(np.float16).view(np.uint16))
