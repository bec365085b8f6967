"""GPU: the drop-in module surface (HIPQuantLinear, prepare_for_inference) behaves like the reference pair."""
import copy
import glob
import os

import numpy as np
import pytest
import torch

from oracle import hqq_ref, linear_ref

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _dev():
    return torch.device("cuda:0")


class _Block(torch.nn.Module):
    def __init__(self, layers):
        super().__init__()
        for k, v in layers.items():
            setattr(self, k, v)


def _toy_model(bias=False):
    from amq_amd.hqq_format import random_hqq
    from amq_amd.patching import HQQWeightsModule
    specs = {"q_proj": (2, 256, 512), "k_proj": (3, 128, 512), "down_proj": (4, 512, 256)}
    hs = {k: random_hqq(n, kk, b, seed=i, bias=bias) for i, (k, (b, n, kk)) in enumerate(specs.items())}
    model = torch.nn.Module()
    model.layers = torch.nn.ModuleList([_Block({k: HQQWeightsModule(h.to(_dev())) for k, h in hs.items()})])
    model.norm = torch.nn.LayerNorm(8)            # a non-quantized child that must be left alone
    return model, hs


@pytest.mark.parametrize("bias", [False, True])
def test_prepare_for_inference_cache_roundtrip(tmp_path, bias):
    from amq_amd.patching import prepare_for_inference
    from amq_amd.quant_linear import HIPQuantLinear
    model, hs = _toy_model(bias)
    path = str(tmp_path / "toy_HIPLinear.pt")
    prepare_for_inference(model, backend="gptq", load_path=path)          # legacy backend name accepted
    assert os.path.exists(path)
    blk = model.layers[0]
    assert isinstance(blk.q_proj, HIPQuantLinear) and blk.q_proj.name == "q_proj" and isinstance(model.norm, torch.nn.LayerNorm)
    assert (blk.q_proj.bits, blk.q_proj.infeatures, blk.q_proj.outfeatures, blk.q_proj.group_size) == (2, 512, 256, 128)
    assert hasattr(blk.q_proj, "weight")                                   # dummy param like patch_add_weight_param
    x = torch.randn(2, 3, 512, generator=torch.Generator().manual_seed(0)).half()
    outs = {}
    for name in ("q_proj", "k_proj"):
        h = hs[name]
        y = getattr(blk, name)(x.to(_dev()))
        assert y.shape == (2, 3, h.shape[0]) and y.dtype == torch.float16
        w = hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), h.nbits, h.shape)
        ref = linear_ref.linear_f16(x.numpy(), w, None if h.bias is None else h.bias.numpy()).astype(np.float32)
        err = np.abs(y.float().cpu().numpy() - ref)
        assert np.all(err <= 1e-3 * np.abs(ref) + 1e-3 * np.sqrt(np.mean(ref ** 2)))
        outs[name] = y
    # second model: load the cache instead of re-packing (patching.py:185-189 behaviour)
    model2, _ = _toy_model(bias)
    prepare_for_inference(model2, backend="hip", load_path=path)
    for name in ("q_proj", "k_proj"):
        assert torch.equal(getattr(model2.layers[0], name)(x.to(_dev())), outs[name])
    # deepcopy / cpu round trip / setattr, as amq_speed_benchmark.py:231-256 does
    m3 = copy.deepcopy(model).to("cpu").to(_dev())
    assert torch.equal(m3.layers[0].q_proj(x.to(_dev())), outs["q_proj"])
    with pytest.raises(RuntimeError):
        prepare_for_inference(model2, backend="marlin")


def test_float32_input_is_cast_like_the_reference():
    from amq_amd.quant_linear import HIPQuantLinear
    from amq_amd.hqq_format import random_hqq
    h = random_hqq(64, 256, 4, seed=3)
    m = HIPQuantLinear.from_hqq(h, device=_dev())
    x = torch.randn(5, 256).to(_dev())
    y = m(x)
    assert y.dtype == torch.float32 and y.shape == (5, 64)
    assert torch.allclose(y, m(x.half()).float())
    with pytest.raises(NotImplementedError):
        HIPQuantLinear(8, 128, 256, 64)
    with pytest.raises(NotImplementedError):
        HIPQuantLinear(4, 96, 256, 64)
    assert HIPQuantLinear(4, 64, 256, 64).meta.numel() == 256 // 64 * 64 * 2       # groups of 64 / 32: 128 / group pairs per native tile row


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "hqq_b*_128x512.npz"))), ids=os.path.basename)
def test_pack_and_reference_buffer_imports(path):
    """pack(W, scales, zeros) (GPTQLinear.pack signature) and the GPTQ / AWQ buffer imports on reference captures"""
    from amq_amd.quant_linear import HIPQuantLinear
    from amq_amd.patching import load_reference_cache
    g = {k: v for k, v in np.load(path).items()}
    bits, (n, k) = int(g["nbits"]), tuple(int(v) for v in g["shape"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(_dev())
    m = HIPQuantLinear(bits, 128, k, n)
    m = m.to(_dev())
    m.pack(t(g["W_deq"]), t(g["scale"].reshape(n, -1)), t(g["zero"].reshape(n, -1)))
    assert np.array_equal(m.dequantize().cpu().numpy().view(np.uint16), g["W_deq"].view(np.uint16))
    sd = {"model.layers.0.mlp.up_proj.qweight": torch.from_numpy(g["gptq_qweight"]),
          "model.layers.0.mlp.up_proj.scales": torch.from_numpy(g["gptq_scales"]),
          "model.layers.0.mlp.up_proj.zeros": torch.from_numpy(g["gptq_zeros"])}
    if bits == 4:
        sd.update({"model.layers.0.self_attn.q_proj.qweight": torch.from_numpy(g["awq_qweight"]),
                   "model.layers.0.self_attn.q_proj.scales": torch.from_numpy(g["awq_scales"]),
                   "model.layers.0.self_attn.q_proj.scaled_zeros": torch.from_numpy(g["awq_scaled_zeros"])})
    mods = load_reference_cache(sd)
    up = mods["model.layers.0.mlp.up_proj"]
    assert (up.bits, up.outfeatures, up.infeatures, up.name) == (bits, n, k, "up_proj")
    y = up(t(g["gptq_x"])).float().cpu().numpy()
    yf = g["gptq_y"].astype(np.float32)
    assert np.max(np.abs(y - yf)) <= 3e-3 * np.sqrt(np.mean(yf ** 2)) + 1e-3 * np.max(np.abs(yf))
    if bits == 4:
        q = mods["model.layers.0.self_attn.q_proj"]
        assert torch.equal(q.qweight, up.qweight)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "hqq_b*_128x512.npz"))), ids=os.path.basename)
def test_reference_ffi_shaped_entry_points(path):
    """amq_vecquantmatmul_faster_old / amq_gemv_4bit / amq_gemm_4bit take the reference's own Format B / C buffers
    (captured from GPTQLinear / FT_QuantLinear) and behave like the pybind functions they replace."""
    import ctypes
    from amq_amd import _lib
    from oracle import gptq_ref, awq_ref
    lib = _lib.load()
    g = {k: v for k, v in np.load(path).items()}
    bits, (n, k) = int(g["nbits"]), tuple(int(v) for v in g["shape"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(_dev())
    vp = lambda x: ctypes.c_void_p(x.data_ptr())
    st = _lib.current_stream()
    for m in (1, 5, 128):
        x = t(g["gptq_x"][:m])
        ws = torch.empty(lib.amq_compat_workspace_bytes(bits, m, n, k), dtype=torch.uint8, device=_dev())
        mul = torch.full((m, n), 0.5, dtype=torch.float32, device=_dev())          # accumulated in place
        qw, sc, zr = t(g["gptq_qweight"]), t(g["gptq_scales"]), t(g["gptq_zeros"])
        for valid in (0, 1):                                                         # second call reuses the native copy
            _lib.check(lib.amq_vecquantmatmul_faster_old(bits, vp(x), vp(qw), vp(mul), vp(sc), vp(zr), 128, k // 2, m,
                                                         qw.shape[0], n, vp(ws), ws.numel(), valid, st))
        w = gptq_ref.dequant_kernel(g["gptq_qweight"], g["gptq_scales"], g["gptq_zeros"], bits)
        ref = linear_ref.linear_f16(g["gptq_x"][:m], w).astype(np.float32)
        got = mul.cpu().numpy()
        assert np.all(np.abs(got - (0.5 + 2 * ref)) <= 2e-3 * np.abs(ref) + 2e-3 * np.sqrt(np.mean(ref ** 2)))
    assert lib.amq_vecquantmatmul_faster_old(bits, vp(x), vp(qw), vp(mul), vp(sc), vp(zr), 128, k, 1, qw.shape[0], n,
                                             vp(ws), ws.numel(), 0, st) == -2        # wrong vec_height
    if bits == 4:
        kq, sc, sz = t(g["awq_qweight"]), t(g["awq_scales"]), t(g["awq_scaled_zeros"])
        w = awq_ref.dequant_kernel(g["awq_qweight"], g["awq_scales"], g["awq_scaled_zeros"])
        ws = torch.empty(lib.amq_compat_workspace_bytes(4, 1, n, k), dtype=torch.uint8, device=_dev())
        for fn, m in ((lib.amq_gemv_4bit, 3), (lib.amq_gemv_4bit, 9), (lib.amq_gemm_4bit, 64)):
            x = t(g["gptq_x"][:m])
            y = torch.empty(m, n, dtype=torch.float16, device=_dev())
            _lib.check(fn(vp(x), vp(kq), vp(sc), vp(sz), vp(y), m, n, k, 128, vp(ws), ws.numel(), 0, st))
            ref = linear_ref.linear_f16(g["gptq_x"][:m], w).astype(np.float32)
            assert np.all(np.abs(y.float().cpu().numpy() - ref) <= 1e-3 * np.abs(ref) + 1e-3 * np.sqrt(np.mean(ref ** 2)))


FINE_CASES = sorted(glob.glob(os.path.join(GOLDEN, "hqq_g64_b*.npz")) + glob.glob(os.path.join(GOLDEN, "hqq_g32_b*.npz")))


@pytest.mark.parametrize("path", FINE_CASES, ids=[os.path.basename(c) for c in FINE_CASES])
def test_reference_ffi_shaped_gptq_call_with_finer_groups(path):
    """vecquant{2,3,4}matmul_faster_old takes any groupsize (auto_gptq_kernel.cu:203: g = k / groupsize): the reference's own GPTQLinear
    buffers of its group-64 / group-32 layers through amq_vecquantmatmul_faster_old: the GEMV kernel up to 8 rows, the pair-aware few-row
    GEMM up to 256, the pair-aware tiled GEMM beyond (this call carries no workspace, so never the dequantize-once route)"""
    import ctypes
    from amq_amd import _lib
    from oracle import gptq_ref
    lib = _lib.load()
    g = {k: v for k, v in np.load(path).items()}
    bits, (n, k), G = int(g["nbits"]), tuple(int(v) for v in g["shape"]), int(g["group_size"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(_dev())
    vp = lambda x: ctypes.c_void_p(x.data_ptr())
    st = _lib.current_stream()
    qw, sc, zr = t(g["gptq_qweight"]), t(g["gptq_scales"]), t(g["gptq_zeros"])
    w = gptq_ref.dequant_kernel(g["gptq_qweight"], g["gptq_scales"], g["gptq_zeros"], bits, G)
    for m in (1, 5, 16, 40, 128):
        x = t(g["gptq_x"][:m])
        ws = torch.empty(lib.amq_compat_workspace_bytes(bits, m, n, k), dtype=torch.uint8, device=_dev())
        mul = torch.full((m, n), 0.5, dtype=torch.float32, device=_dev())
        for valid in (0, 1):
            _lib.check(lib.amq_vecquantmatmul_faster_old(bits, vp(x), vp(qw), vp(mul), vp(sc), vp(zr), G, k // 2, m,
                                                         qw.shape[0], n, vp(ws), ws.numel(), valid, st))
        ref = linear_ref.linear_f16(g["gptq_x"][:m], w).astype(np.float32)
        assert np.all(np.abs(mul.cpu().numpy() - (0.5 + 2 * ref)) <= 2e-3 * np.abs(ref) + 2e-3 * np.sqrt(np.mean(ref ** 2)))
    xs = np.concatenate([g["gptq_x"], g["gptq_x"], g["gptq_x"][:44]], axis=0)       # 300 rows
    x = t(xs)
    ws = torch.empty(lib.amq_compat_workspace_bytes(bits, 300, n, k), dtype=torch.uint8, device=_dev())
    mul = torch.zeros(300, n, dtype=torch.float32, device=_dev())
    _lib.check(lib.amq_vecquantmatmul_faster_old(bits, vp(x), vp(qw), vp(mul), vp(sc), vp(zr), G, k // 2, 300, qw.shape[0], n,
                                                 vp(ws), ws.numel(), 0, st))
    ref = linear_ref.linear_f16(xs, w).astype(np.float32)
    assert np.all(np.abs(mul.cpu().numpy() - ref) <= 2e-3 * np.abs(ref) + 2e-3 * np.sqrt(np.mean(ref ** 2)))


def test_module_walk_decode_matches_runner():
    """The drop-in shape of the decode loop (HF-style walk over HIPQuantLinear.forward calls, one launch per module) and
    the fused runner produce the same tokens on the same weights / caches; eager and hipGraph replays of the walk are
    bit-identical; the per-module argument cache survives deepcopy and notices moved buffers."""
    import copy
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    from amq_amd.module_walk import ModuleWalkLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2, 1, vocab=1024))
    al = {name: [b, c] for name, b, c in zip(cfg["linear"], [2, 3, 4, 3, 2, 4, 3], [4, 2, 3, 3, 4, 2, 2])}
    m = QuantLlama(cfg, al, device="cuda:0", max_seq=64, seed=8)
    mw = ModuleWalkLlama(m)
    ids = torch.randint(0, 1024, (9,), generator=torch.Generator().manual_seed(2)).to("cuda:0")
    m.prefill(ids)
    ref_tokens, ref_logits = [], []
    for _ in range(5):
        m.decode_step()
        ref_tokens.append(int(m.token.item())); ref_logits.append(m.logits.float().clone())
    for use_graph in (False, True):
        m.prefill(ids)
        same = True
        for i in range(5):
            mw.decode_step(use_graph=use_graph)
            scale = ref_logits[i].abs().max()
            if same:                                              # (a flipped argmax would send the two runs down different sequences)
                assert (m.logits.float() - ref_logits[i]).abs().max() <= 2e-2 * scale
                # round 3: grouped siblings, fused MLP, deferred norms and residual epilogues make the walk issue the runner's
                # own five launches per block -- the same bits, not just the same tokens
                assert torch.equal(m.logits.float(), ref_logits[i])
            same = same and int(m.token.item()) == ref_tokens[i]
            if i == 0:
                first = m.logits.clone() if not use_graph else first
                if use_graph:
                    assert torch.equal(m.logits, first)          # graph replay == eager launches of the same walk
        m.check()
    lin = mw.layers[0].self_attn.q_proj
    x = torch.randn(1, 512, device="cuda:0").half()
    y = lin(x)
    lin2 = copy.deepcopy(lin)                                    # amq_speed_benchmark.py:231 deep-copies the patched model
    assert lin2.qweight.data_ptr() != lin.qweight.data_ptr() and torch.equal(lin2(x), y)
    assert torch.equal(lin(x.reshape(1, 1, 512)).reshape(1, -1), y) and lin(x.float()).dtype == torch.float32


@pytest.mark.parametrize("rows", [1, 5, 40])
def test_module_accepts_bf16_and_fp32_inputs(rows):
    """HIPQuantLinear.forward on non-fp16 inputs: cast to fp16, computed by the fp16 kernels, returned in the caller's dtype -- what
    the reference modules do (autogptq.py:166-169 casts with a warning); there is no bf16 arithmetic path"""
    import numpy as np
    from amq_amd.hqq_format import random_hqq
    from amq_amd.quant_linear import HIPQuantLinear
    dev = torch.device("cuda:0")
    h = random_hqq(256, 512, 3, seed=2, bias=True)
    mod = HIPQuantLinear.from_hqq(h, device=dev)
    x = torch.randn(rows, 512, generator=torch.Generator().manual_seed(rows))
    y16 = mod(x.half().to(dev))
    for dt in (torch.bfloat16, torch.float32):
        xin = x.to(dt).to(dev)
        y = mod(xin)
        assert y.dtype == dt and y.shape == (rows, 256)
        want = mod(xin.to(torch.float16)).to(dt)            # the documented semantics
        assert torch.equal(y, want)
    # bf16 rounding of the INPUT is the only difference from the fp16 call
    yb = mod(x.to(torch.bfloat16).to(dev)).float()
    assert (yb - y16.float()).abs().max() <= 2e-2 * y16.float().abs().max() + 1e-2


def test_module_edge_shapes():
    """empty and ragged inputs through HIPQuantLinear.forward: zero rows, leading dims of any rank, a non-contiguous view, and the
    8 / 9-row seam between the weight-streaming GEMV and the MFMA GEMM -- all against the CPU oracle linear"""
    from amq_amd.hqq_format import random_hqq
    from amq_amd.quant_linear import HIPQuantLinear
    dev = torch.device("cuda:0")
    n, k = 272, 384                                             # N a multiple of 16 only, K = 3 groups
    h = random_hqq(n, k, 3, seed=5, bias=True)
    mod = HIPQuantLinear.from_hqq(h, device=dev)
    w = hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), 3, (n, k))
    y0 = mod(torch.empty(0, k, dtype=torch.float16, device=dev))
    assert y0.shape == (0, n) and y0.dtype == torch.float16
    assert mod(torch.empty(2, 0, k, dtype=torch.float16, device=dev)).shape == (2, 0, n)
    gen = torch.Generator().manual_seed(3)
    for shape in [(1, k), (8, k), (9, k), (2, 3, k), (2, 1, 5, k), (17, k)]:
        x = torch.randn(*shape, generator=gen).to(torch.float16)
        y = mod(x.to(dev))
        assert y.shape == shape[:-1] + (n,)
        ref = linear_ref.linear_f16(x.reshape(-1, k).numpy(), w, h.bias.numpy()).astype(np.float32)
        got = y.reshape(-1, n).float().cpu().numpy()
        assert np.all(np.abs(got - ref) <= 2e-3 * np.abs(ref) + 2e-3 * np.sqrt((ref ** 2).mean())), shape
    xt = torch.randn(k, 6, generator=gen).to(torch.float16).to(dev).t()      # [6, k], strides (1, 6)
    assert not xt.is_contiguous() and torch.equal(mod(xt), mod(xt.contiguous()))
    with pytest.raises(ValueError):
        mod(torch.zeros(2, k + 128, dtype=torch.float16, device=dev))


def test_second_gpu_while_the_first_is_current():
    """weights and activations on cuda:1 while cuda:0 stays the current device (HF device_map without set_device): every launch, its kernel
    attributes and its CU counts belong to cuda:1 -- through the ctypes path (ops) and through the C++ extension (module forward).  Needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from amq_amd import ops
    from amq_amd.hqq_format import random_hqq
    from amq_amd.quant_linear import HIPQuantLinear
    torch.cuda.set_device(0)
    d1 = torch.device("cuda:1")
    n, k, bits = 512, 1024, 3
    h = random_hqq(n, k, bits, seed=3)
    h0, h1 = h.to("cuda:0"), h.to(d1)
    q0, m0 = ops.repack_from_hqq(h0.W_q, h0.scale.reshape(-1), h0.zero.reshape(-1), bits, n, k)
    q1, m1 = ops.repack_from_hqq(h1.W_q, h1.scale.reshape(-1), h1.zero.reshape(-1), bits, n, k)
    assert q1.device == d1 and torch.equal(q1.cpu(), q0.cpu()) and torch.equal(m1.cpu(), m0.cpu())
    for rows in (1, 5, 200):
        x = torch.randn(rows, k, generator=torch.Generator().manual_seed(rows)).half()
        y0 = ops.linear(x.to("cuda:0"), q0, m0, bits, ops.MODE_HQQ, n, k)
        y1 = ops.linear(x.to(d1), q1, m1, bits, ops.MODE_HQQ, n, k)
        assert torch.cuda.current_device() == 0 and y1.device == d1 and torch.equal(y1.cpu(), y0.cpu())
    lin = HIPQuantLinear.from_hqq(h, device=d1)
    x = torch.randn(3, k, generator=torch.Generator().manual_seed(9)).half()
    assert torch.equal(lin(x.to(d1)).cpu(), ops.linear(x.to("cuda:0"), q0, m0, bits, ops.MODE_HQQ, n, k).cpu())
    assert torch.cuda.current_device() == 0
