"""GPU parity of the optional bfloat16 entry points (csrc/amq_bf16.hip, the bf16 instantiation of amq_gemm_f16.hip) through the C ABI.

Oracle: oracle/hqq_ref.dequantize_bf16 + oracle/linear_ref.linear_bf16, pinned bit-exactly (weights) by golden vectors captured from the
reference's HQQLinear(compute_dtype=torch.bfloat16) on CPU (tests/golden/gen_golden_bf16.py, tests/test_oracle_golden.py).

Bar: dequantized weights bit-exact (also the weights the GEMV kernel forms in registers: recovered with x = unit rows); matmul outputs within
ONE bf16 ulp of the output, |y - ref| <= 2^-7 * |ref| + 2^-8 * rms(ref) -- bfloat16 keeps 8 significant bits, so the fp16 path's 1e-3 bar is
below its rounding step; the reference side is torch's CPU F.linear on the oracle's bf16 weights."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import hqq_ref, linear_ref

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "bf16_b*.npz")))


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _load(path):
    d = np.load(path)
    return {k: d[k] for k in d.files}


def _bf(bits_arr):
    """uint16 bit patterns -> bfloat16 tensor on the GPU"""
    return torch.from_numpy(np.ascontiguousarray(bits_arr).view(np.int16)).view(torch.bfloat16).to(_dev())


def _bits(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def _assert_bf16_close(y_bits, ref_bits, what=""):
    y = hqq_ref.bf16_bits_to_f32(y_bits).astype(np.float64)
    ref = hqq_ref.bf16_bits_to_f32(ref_bits).astype(np.float64)
    bar = 2.0 ** -7 * np.abs(ref) + 2.0 ** -8 * np.sqrt(np.mean(ref ** 2))
    bad = np.abs(y - ref) > bar
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} out of tolerance, worst {np.max(np.abs(y - ref) / bar):.3f} of the bar"


def _random_bf16_layer(n, k, bits, seed, bias=False):
    """synthetic HQQ layer with bfloat16 meta: hqq_format.random_hqq's recipe, scale / zero rounded to bf16"""
    from amq_amd.hqq_format import random_hqq, HQQWeights
    h = random_hqq(n, k, bits, seed=seed, bias=bias)
    return HQQWeights(h.W_q, h.scale.float().to(torch.bfloat16), h.zero.float().to(torch.bfloat16), bits, (n, k), 128,
                      None if h.bias is None else h.bias.float().to(torch.bfloat16))


def _native(h):
    from amq_amd import ops
    n, k = h.shape
    hd = h.to(_dev())
    return ops.repack_from_hqq(hd.W_q, hd.scale.reshape(-1).contiguous(), hd.zero.reshape(-1).contiguous(), h.nbits, n, k)


def _oracle_w_bits(h):
    return hqq_ref.dequantize_bf16(h.W_q.numpy(), _bits(h.scale), _bits(h.zero), h.nbits, h.shape, 128)


def _cpu_linear_bits(x, w_bits, bias=None, residual=None):
    """the reference side: torch CPU F.linear on bf16 tensors (fp32 accumulate, one bf16 rounding), bias / residual as separate bf16 adds"""
    w = torch.from_numpy(w_bits.view(np.int16)).view(torch.bfloat16)
    y = torch.nn.functional.linear(x.cpu(), w)
    if bias is not None:
        y = y + bias.cpu()
    if residual is not None:
        y = residual.cpu() + y
    return _bits(y)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_bf16_golden(path):
    """the reference's own bf16 layer: weights bit for bit, y within one bf16 ulp (3 rows: the reference's capture; 16 rows: the kernel's row limit)"""
    from amq_amd import ops
    g = _load(path)
    bits, (n, k) = int(g["nbits"]), tuple(int(v) for v in g["shape"])
    qn, mn = ops.repack_from_hqq(torch.from_numpy(g["W_q"]).to(_dev()), _bf(g["scale"].reshape(-1)), _bf(g["zero"].reshape(-1)), bits, n, k)
    assert mn.dtype == torch.bfloat16
    w = ops.dequantize_bf16(qn, mn, bits, n, k)
    assert np.array_equal(_bits(w), g["W_deq"])
    # ... and straight from Format A (the standalone dequantize)
    w2 = ops.dequantize_hqq(torch.from_numpy(g["W_q"]).to(_dev()), _bf(g["scale"].reshape(-1)), _bf(g["zero"].reshape(-1)), bits, n, k)
    assert w2.dtype == torch.bfloat16 and np.array_equal(_bits(w2), g["W_deq"])
    bias = _bf(g["bias"]) if "bias" in g else None
    for tag in ("", "16"):
        y = ops.linear_bf16(_bf(g["x" + tag]), qn, mn, bits, n, k, bias=bias)
        assert y.dtype == torch.bfloat16
        _assert_bf16_close(_bits(y), g["y" + tag + "_ref"], f"{os.path.basename(path)} rows{tag or 3}")
        # ... and the oracle's own restatement of the forward agrees with what was captured
        _assert_bf16_close(linear_ref.linear_bf16(g["x" + tag], g["W_deq"], g.get("bias")), g["y" + tag + "_ref"])


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_bf16_gemv_weights_equal_oracle_weights(bits):
    """x = unit rows: every output is ONE weight (one exact product, exact sums of zeros), so the few-row kernel's in-register dequantize is read
    back bit for bit -- against the oracle, not against this build's dequantize kernel"""
    from amq_amd import ops
    n, k = 64, 384
    h = _random_bf16_layer(n, k, bits, seed=10 + bits)
    qn, mn = _native(h)
    want = _oracle_w_bits(h)                                   # [n, k]
    got = np.zeros((n, k), np.uint16)
    for k0 in range(0, k, 16):
        x = torch.zeros(16, k, dtype=torch.bfloat16, device=_dev())
        x[torch.arange(16), k0 + torch.arange(16)] = 1.0
        y = ops.linear_bf16(x, qn, mn, bits, n, k)             # y[m, n] = W[n, k0 + m]
        got[:, k0:k0 + 16] = _bits(y).T
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,k", [(256, 512), (4096, 4096), (11008, 4096), (4096, 11008)])
@pytest.mark.parametrize("bits", [2, 3, 4])
def test_bf16_few_rows_vs_oracle(bits, n, k):
    """7B layer shapes, 1 .. 16 rows (LDS-staged x, and -- 16 rows of K = 11008 -- x read per tile from global memory), bias + residual"""
    from amq_amd import ops
    h = _random_bf16_layer(n, k, bits, seed=100 * bits + n % 97, bias=True)
    qn, mn = _native(h)
    wb = _oracle_w_bits(h)
    assert np.array_equal(_bits(ops.dequantize_bf16(qn, mn, bits, n, k)), wb)
    g = torch.Generator().manual_seed(bits + n + k)
    bias = h.bias.to(_dev())
    for m in (1, 2, 5, 8, 16):
        x = torch.randn(m, k, generator=g).to(torch.bfloat16)
        y = ops.linear_bf16(x.to(_dev()), qn, mn, bits, n, k)
        _assert_bf16_close(_bits(y), _cpu_linear_bits(x, wb), f"{bits} bit {n}x{k} rows {m}")
    x = torch.randn(5, k, generator=g).to(torch.bfloat16)
    res = torch.randn(5, n, generator=g).to(torch.bfloat16)
    y = ops.linear_bf16(x.to(_dev()), qn, mn, bits, n, k, bias=bias, residual=res.to(_dev()))
    ref = _cpu_linear_bits(x, wb, bias=h.bias, residual=res)
    # (three bf16 roundings on each side: an ulp per add can differ)
    yf, rf = hqq_ref.bf16_bits_to_f32(_bits(y)).astype(np.float64), hqq_ref.bf16_bits_to_f32(ref).astype(np.float64)
    assert np.all(np.abs(yf - rf) <= 3 * 2.0 ** -7 * np.abs(rf) + 2.0 ** -7 * np.sqrt(np.mean(rf ** 2)))


@pytest.mark.parametrize("bits,m,n,k", [(4, 17, 256, 512), (3, 300, 1040, 384), (2, 1024, 4096, 4096), (3, 2500, 512, 1024)])
def test_bf16_many_rows_vs_oracle(bits, m, n, k):
    """beyond 16 rows: dequantize once + the bf16 instantiation of the MFMA-bound GEMM kernel (ragged M and N tiles)"""
    from amq_amd import ops
    h = _random_bf16_layer(n, k, bits, seed=7 * bits + m, bias=True)
    qn, mn = _native(h)
    wb = _oracle_w_bits(h)
    g = torch.Generator().manual_seed(m)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16)
    y = ops.linear_bf16(x.to(_dev()), qn, mn, bits, n, k)
    _assert_bf16_close(_bits(y), _cpu_linear_bits(x, wb), "plain")
    yb = ops.linear_bf16(x.to(_dev()), qn, mn, bits, n, k, bias=h.bias.to(_dev()))
    # the bias is a SEPARATE bf16 add of the plain result on both sides: exact given y
    want = (y.float() + h.bias.to(_dev()).float()).to(torch.bfloat16)
    assert torch.equal(yb, want)
    # 3-D input, leading dims kept
    y3 = ops.linear_bf16(x.to(_dev()).reshape(2, m // 2, k) if m % 2 == 0 else x.to(_dev()).reshape(1, m, k), qn, mn, bits, n, k)
    assert y3.shape[-1] == n and torch.equal(y3.reshape(m, n), y)


@pytest.mark.parametrize("group", [32, 64, 128, 256])
@pytest.mark.parametrize("bits", [2, 3, 4])
def test_bf16_dequantize_hqq_groups(bits, group):
    """Format A -> bf16 W at every group size the fp16 kernel takes, against the oracle bit for bit (3 bit: the zero-padded last chunks)"""
    from amq_amd import ops
    from amq_amd.hqq_format import random_hqq
    n, k = 80, 512
    h = random_hqq(n, k, bits, seed=bits + group, group=group)
    sb, zb = h.scale.float().to(torch.bfloat16), h.zero.float().to(torch.bfloat16)
    want = hqq_ref.dequantize_bf16(h.W_q.numpy(), _bits(sb), _bits(zb), bits, (n, k), group)
    got = ops.dequantize_hqq(h.W_q.to(_dev()), sb.reshape(-1).to(_dev()), zb.reshape(-1).to(_dev()), bits, n, k, group=group)
    assert np.array_equal(_bits(got), want)


def test_bf16_rows_agree_across_the_kernel_boundary():
    """row r of a 16-row launch (few-row kernel) and of a 17-row launch (GEMM kernel) see the same weights: both within the bar of the reference"""
    from amq_amd import ops
    bits, n, k = 3, 512, 1024
    h = _random_bf16_layer(n, k, bits, seed=5)
    qn, mn = _native(h)
    wb = _oracle_w_bits(h)
    x = torch.randn(17, k, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16)
    y16 = ops.linear_bf16(x[:16].to(_dev()), qn, mn, bits, n, k)
    y17 = ops.linear_bf16(x.to(_dev()), qn, mn, bits, n, k)
    ref = _cpu_linear_bits(x, wb)
    _assert_bf16_close(_bits(y16), ref[:16])
    _assert_bf16_close(_bits(y17), ref)
    # deterministic: the same launch twice gives the same bits
    assert torch.equal(y16, ops.linear_bf16(x[:16].to(_dev()), qn, mn, bits, n, k))


def test_bf16_argument_checks():
    from amq_amd import ops, _lib
    bits, n, k = 4, 64, 256
    h = _random_bf16_layer(n, k, bits, seed=2)
    qn, mn = _native(h)
    x = torch.randn(2, k).to(torch.bfloat16).to(_dev())
    with pytest.raises(ValueError):                              # fp16 x into the bf16 entry point
        ops.linear_bf16(x.half(), qn, mn, bits, n, k)
    with pytest.raises(ValueError):                              # an fp16-meta buffer into the bf16 entry point
        ops.linear_bf16(x, qn, mn.view(torch.float16), bits, n, k)
    with pytest.raises(ValueError):                              # ... and a bf16-meta buffer into the fp16 one
        ops.gemv(x.half(), qn, mn, bits, ops.MODE_HQQ, n, k)
    assert ops.linear_bf16(x[:0], qn, mn, bits, n, k).shape == (0, n)
    lib = _lib.load()
    y = torch.empty(17, n, dtype=torch.bfloat16, device=_dev())
    x17 = torch.randn(17, k).to(torch.bfloat16).to(_dev())
    assert lib.amq_gemm_bf16_workspace_bytes(17, n, k) == n * k * 2 and lib.amq_gemm_bf16_workspace_bytes(16, n, k) == 0
    rc = lib.amq_gemm_bf16(bits, _lib.ptr(x17), _lib.ptr(qn), _lib.ptr(mn), None, None, _lib.ptr(y), 17, n, k, 128, 0, 0, None, 0, None)
    assert rc == -1 and b"workspace" in lib.amq_last_error()
    rc = lib.amq_gemv_bf16(bits, _lib.ptr(x17), _lib.ptr(qn), _lib.ptr(mn), None, None, _lib.ptr(y), 17, n, k, 128, 0, 0, None)
    assert rc == -2
    rc = lib.amq_gemv_bf16(bits, _lib.ptr(x17), _lib.ptr(qn), _lib.ptr(mn), None, None, _lib.ptr(y), 2, n, k, 64, 0, 0, None)
    assert rc == -2                                              # groups of 64 / 32: fp16 entry points only


def test_bf16_strided_rows_through_the_c_abi():
    """x_stride / y_stride: rows taken out of wider buffers"""
    from amq_amd import ops, _lib
    bits, n, k = 4, 128, 512
    h = _random_bf16_layer(n, k, bits, seed=3)
    qn, mn = _native(h)
    wb = _oracle_w_bits(h)
    lib = _lib.load()
    for xs in (k + 64, k + 8):
        xbuf = torch.randn(4, xs, generator=torch.Generator().manual_seed(xs)).to(torch.bfloat16).to(_dev())
        ybuf = torch.zeros(4, n + 32, dtype=torch.bfloat16, device=_dev())
        rc = lib.amq_gemv_bf16(bits, _lib.ptr(xbuf), _lib.ptr(qn), _lib.ptr(mn), None, None, _lib.ptr(ybuf), 4, n, k, 128, xs, n + 32,
                               _lib.current_stream())
        assert rc == 0, lib.amq_last_error()
        torch.cuda.synchronize()
        _assert_bf16_close(_bits(ybuf[:, :n]), _cpu_linear_bits(xbuf[:, :k].cpu().contiguous(), wb))
        assert torch.count_nonzero(ybuf[:, n:]) == 0           # nothing written past a row's N outputs
    rc = lib.amq_gemv_bf16(bits, _lib.ptr(xbuf), _lib.ptr(qn), _lib.ptr(mn), None, None, _lib.ptr(ybuf), 4, n, k, 128, k + 4, n + 32, None)
    assert rc == -2                                              # rows are read in 16-byte pieces


def test_bf16_module_drop_in():
    """HIPQuantLinear built from a bf16 HQQ layer: bf16 buffers, forward in bf16 (fp32 / fp16 callers cast, as the reference's modules do),
    state_dict round trip, deepcopy; prepare_for_inference leaves bf16 modules ungrouped"""
    import copy
    from amq_amd.quant_linear import HIPQuantLinear
    bits, n, k = 3, 256, 512
    h = _random_bf16_layer(n, k, bits, seed=9, bias=True)
    mod = HIPQuantLinear.from_hqq(h, device=_dev())
    assert mod.is_bf16 and mod.meta.dtype == torch.bfloat16 and mod.bias.dtype == torch.bfloat16
    wb = _oracle_w_bits(h)
    x = torch.randn(2, 3, k, generator=torch.Generator().manual_seed(4)).to(torch.bfloat16)
    y = mod(x.to(_dev()))
    assert y.shape == (2, 3, n) and y.dtype == torch.bfloat16
    ref = _cpu_linear_bits(x.reshape(6, k), wb, bias=h.bias)
    yf, rf = hqq_ref.bf16_bits_to_f32(_bits(y.reshape(6, n))).astype(np.float64), hqq_ref.bf16_bits_to_f32(ref).astype(np.float64)
    assert np.all(np.abs(yf - rf) <= 2 * 2.0 ** -7 * np.abs(rf) + 2.0 ** -7 * np.sqrt(np.mean(rf ** 2)))
    assert mod(x.to(_dev()).float()).dtype == torch.float32
    mod2 = HIPQuantLinear(bits, 128, k, n, bias=True, weight_dtype=torch.bfloat16).to(_dev())
    mod2.load_state_dict(mod.state_dict())
    assert mod2.is_bf16 and torch.equal(mod2(x.to(_dev())), y)
    assert torch.equal(copy.deepcopy(mod)(x.to(_dev())), y)
    assert mod.to_kernel_arithmetic() is mod and mod.mode == 0   # (a no-op for bf16 modules)
    from amq_amd import patching
    assert not patching._fusable(mod)


def test_bf16_layers_through_prepare_for_inference(tmp_path):
    """a model quantized with compute_dtype = bfloat16 becomes fp16 modules under prepare_for_inference by default -- what the
    reference's patch_hqq_to_gptq / patch_hqq_to_ft make of every layer (ft.py:62) -- so it keeps the grouped / fused fp16 launches, and layers
    with groups of 64 (which the bf16 kernels do not serve) convert at all; ``keep_bf16=True`` keeps HQQ's bf16 arithmetic where the kernels serve
    the layer, through the cache file as well; dequantize() of a bfloat16 module returns the bf16 weights"""
    from amq_amd import patching
    from amq_amd.hqq_format import random_hqq, HQQWeights
    from amq_amd.quant_linear import HIPQuantLinear

    def model(group):
        m = torch.nn.Module()
        att = torch.nn.Module()
        for i, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            h = random_hqq(256, 512, 4, seed=20 + i, group=group)
            hb = HQQWeights(h.W_q, h.scale.float().to(torch.bfloat16), h.zero.float().to(torch.bfloat16), 4, (256, 512), group, None, nm)
            setattr(att, nm, patching.HQQWeightsModule(hb.to(_dev())))
        m.self_attn = att
        return m

    x = torch.randn(1, 512, generator=torch.Generator().manual_seed(1)).to(torch.float16).to(_dev())
    for group in (128, 64):
        m = patching.prepare_for_inference(model(group))
        mods = [m.self_attn.q_proj, m.self_attn.k_proj, m.self_attn.v_proj]
        assert all(isinstance(q, HIPQuantLinear) and not q.is_bf16 and q.meta.dtype == torch.float16 for q in mods)
        assert all("_group" in q.__dict__ for q in mods)            # grouped like any fp16 model
        assert mods[0](x).dtype == torch.float16
    # opt-in: bf16 kept at groups of 128, fp16 fallback at 64
    mk = patching.prepare_for_inference(model(128), keep_bf16=True)
    assert mk.self_attn.q_proj.is_bf16
    wb = mk.self_attn.q_proj.dequantize()
    assert wb.dtype == torch.bfloat16 and wb.shape == (256, 512)
    assert not patching.prepare_for_inference(model(64), keep_bf16=True).self_attn.q_proj.is_bf16
    # cache file: written from bf16 modules, read back into bf16 modules (not value-cast into fp16 meta)
    path = str(tmp_path / "bf16_cache.pt")
    m1 = patching.prepare_for_inference(model(128), keep_bf16=True, load_path=path)
    m2 = patching.prepare_for_inference(model(128), keep_bf16=True, load_path=path)
    assert m2.self_attn.q_proj.is_bf16
    xb = x.to(torch.bfloat16)
    assert torch.equal(m1.self_attn.q_proj(xb), m2.self_attn.q_proj(xb))
    assert torch.equal(m1.self_attn.q_proj.dequantize(), wb)
