"""GPU: BASELINE.json's full layer shapes (7B / 13B / 70B incl. GQA k/v 1024x8192 and 28672-wide MLP), checked
through size-independent properties, because the CPU oracle cannot finish these sizes in seconds:
  * GEMV (M=1, weight-streaming path) == row 0 of the tiled MFMA GEMM on the same weights (two independent kernels);
  * both == torch matmul on the weights returned by the bit-exact dequantize kernel (fp32 accumulate);
  * homogeneity in x: f(2x) == 2 f(x) exactly (power-of-two scaling commutes with every rounding in the path;
    fp16-subnormal outputs excepted);
  * determinism across repeated launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (N, K) from amq/configs/llama.json linear_shape
    (4096, 4096), (11008, 4096), (4096, 11008),            # 7B
    (5120, 5120), (13824, 5120), (5120, 13824),            # 13B
    (8192, 8192), (1024, 8192), (28672, 8192), (8192, 28672),   # 70B
    # the reference's other families (amq/configs/llama.json:82+, mistral.json, qwen2.json)
    (14336, 4096), (4096, 14336), (1024, 4096),            # Llama-3.x-8B / Mistral-7B (GQA 4: k/v 1024 x 4096)
    (3584, 3584), (512, 3584), (18944, 3584), (3584, 18944),    # Qwen2.5-7B (GQA 7, hidden 28 x 128)
    (27648, 5120), (5120, 27648),                          # Qwen2.5-32B
    (29568, 8192), (8192, 29568),                          # Qwen2.5-72B
]


def _native(bits, n, k, seed):
    from amq_amd.llama import _synthetic_linear
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    return _synthetic_linear(n, k, bits, g, torch.device("cuda:0"))


@pytest.mark.parametrize("n,k", SHAPES)
@pytest.mark.parametrize("bits", [2, 3, 4])
def test_fullsize_properties(bits, n, k):
    from amq_amd import ops
    dev = torch.device("cuda:0")
    l = _native(bits, n, k, seed=n + k + bits)
    x = torch.randn(1, k, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).half()
    y = ops.gemv(x, l.qn, l.mn, bits, l.mode, n, k)
    # (1) two independent kernels agree (fp32 accumulation order differs -> at most an fp16 ulp)
    xm = torch.cat([x, torch.randn(16, k, device=dev).half()])
    ym = ops.gemm(xm, l.qn, l.mn, bits, l.mode, n, k)
    rms = y.float().pow(2).mean().sqrt()
    assert (y[0].float() - ym[0].float()).abs().max() <= 2.0 ** -9 * y.float().abs().max() + 1e-3 * rms
    # (2) against torch on the bit-exactly dequantized weights
    w = ops.dequantize(l.qn, l.mn, bits, l.mode, n, k)
    ref = (x.float() @ w.float().t())
    err = (y.float() - ref).abs()
    assert torch.all(err <= 1e-3 * ref.abs() + 1e-3 * rms)
    assert torch.all((ym.float() - xm.float() @ w.float().t()).abs() <= 1e-3 * (xm.float() @ w.float().t()).abs() + 1e-3 * rms)
    # (3) exact homogeneity, (4) determinism
    y2 = ops.gemv(x * 2, l.qn, l.mn, bits, l.mode, n, k)
    normal = y.float().abs() >= 2.0 ** -13          # fp16-subnormal outputs have fixed absolute spacing: not scale invariant
    assert torch.equal(y2[normal], (y * 2)[normal]) and (y2.float() - 2 * y.float()).abs().max() <= 2.0 ** -24
    assert torch.equal(ops.gemv(x, l.qn, l.mn, bits, l.mode, n, k), y)


def test_llama70b_block_shapes_grouped_decode_step():
    """one 70B-shaped block (GQA: 64 q heads, 8 kv heads) through the grouped launches of the token step"""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = dict(arch.MODEL_CONFIGS["Llama-2-70b-hf"])
    cfg["n_block"] = 1
    cfg["vocab_size"] = 2048
    al = {name: [b] for name, b in zip(cfg["linear"], [4, 2, 3, 3, 2, 4, 3])}
    m = QuantLlama(cfg, al, max_seq=48, seed=0)
    ids = torch.randint(0, 2048, (20,), device="cuda:0")
    m.prefill(ids)
    a = m.logits.clone()
    for _ in range(3):
        m.decode_step()
    assert torch.isfinite(m.logits.float()).all() and int(m.pos.item()) == 23
    # same prefix through the eager path gives the same logits as the graph path
    m2 = QuantLlama(cfg, al, max_seq=48, seed=0)
    m2.prefill(ids)
    assert torch.equal(m2.logits, a)
    for _ in range(3):
        m2.decode_step(use_graph=False)
    assert torch.equal(m2.logits, m.logits)


def test_llama70b_whole_replica_decodes():
    """BASELINE.json configs[4]'s unit of work: ONE whole Llama-2-70B replica (80 blocks, GQA 64 / 8 heads, 28672-wide MLP, the synthesized avg-3-bit
    arch; ~26 GB of synthetic weights) prefills 64 tokens and decodes: the hipGraph step and the eager step produce the same tokens and logits bit for
    bit, every logit finite, no step past the cache, and the replica's linears weigh what the arch says (the bench line's roofline figure uses it)"""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = arch.MODEL_CONFIGS["Llama-2-70b-hf"]
    a, usage = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=())
    assert abs(usage - 3.0) < 0.05 and len(a["linear"]["self_attn.q_proj"]) == 80
    m = QuantLlama(cfg, a["linear"], device=torch.device("cuda:0"), max_seq=64 + 16, seed=0)
    assert len(m.blocks) == 80
    # packed payload + 0.25 bit of (scale, zero) per weight: BASELINE.md section 3's per-layer bytes, summed (25.9 GB per token)
    want = sum(blk[name].N * blk[name].K * (4 * blk[name].bits + 1) // 32 for blk in m.blocks for name in m.cfg["linear"])
    assert m.linear_bytes_per_token() == want and 25e9 < want < 27e9
    ids = torch.randint(0, m.vocab - 1, (64,), generator=torch.Generator().manual_seed(0)).to("cuda:0")
    m.prefill(ids)
    first = int(m.token.item())
    toks_graph = []
    for _ in range(6):
        m.decode_step()
        toks_graph.append(int(m.token.item()))
    logits_graph = m.logits.clone()
    m.check()
    assert torch.isfinite(logits_graph.float()).all() and int(m.pos.item()) == 70
    m.prefill(ids)                                          # the same replica again, eager steps over the same weights
    assert int(m.token.item()) == first
    toks_eager = []
    for _ in range(6):
        m.decode_step(use_graph=False)
        toks_eager.append(int(m.token.item()))
    assert toks_eager == toks_graph and torch.equal(m.logits, logits_graph)


def test_llama13b_block_batched_prompt_pass_at_config4_size():
    """BASELINE.json configs[3] at size: one 13B-shaped block, 16 x 2048 prompt rows through prefill_batch (every linear at
    M = 32768).  Checked through properties: finite logits; the hand-written kernels and the dequantize + library route
    agree; the batch is a batch (sequence b of the batched pass == the same sequence passed alone)."""
    from amq_amd import arch, ops
    from amq_amd.llama import QuantLlama
    cfg = dict(arch.MODEL_CONFIGS["Llama-2-13b-hf"])
    cfg["n_block"] = 1
    cfg["vocab_size"] = 2048
    al = {name: [b] for name, b in zip(cfg["linear"], [3, 2, 4, 3, 2, 3, 4])}
    m = QuantLlama(cfg, al, max_seq=2048, seed=1)
    ids = torch.randint(0, 2048, (16, 2048), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(2))
    saved = ops.LIB_GEMM_ROWS
    try:
        ops.LIB_GEMM_ROWS = 0                     # hand-written MFMA kernels at every size
        own = m.prefill_batch(ids).clone()
        one = m.prefill_batch(ids[5:6]).clone()
        ops.LIB_GEMM_ROWS = 1024                  # bit-exact dequantize kernel + library GEMM
        lib = m.prefill_batch(ids).clone()
    finally:
        ops.LIB_GEMM_ROWS = saved
    assert torch.isfinite(own.float()).all() and own.shape == (16, 2048)
    scale = own.float().abs().max()
    assert (own.float() - lib.float()).abs().max() <= 1e-2 * scale
    assert (own[5].float() - one[0].float()).abs().max() <= 1e-2 * scale


# ---- BASELINE.json configs[1]: "Llama-2-7B uniform 4-bit (AWQ pack)" -- the reference formats' own arithmetic at size.
# Buffers imported from the reference's GPTQ / AWQ caches carry MODE_FMA (w = fma(q, s, c), the CUDA kernels' dequant:
# auto_gptq_kernel.cu:206, gemv_cuda.cu:151; module side hqq/backends/ft.py:103-145, autogptq.py:111-156).  The packed
# buffers are built by the ORACLE's packers (numpy restatements pinned to the reference's captures in
# tests/test_oracle_golden.py) from random integers, imported through amq_repack_from_awq / amq_repack_from_gptq, and the
# GEMV is then checked at the three 7B layer shapes: weights bit-identical to the oracle's kernel dequant, GEMV == row 0 of
# the MFMA GEMM == fp32 matmul on those weights, exact homogeneity, determinism.
FMA_CASES = [(fmt, bits, n, k) for (n, k) in SHAPES[:3] for fmt, bits in (("awq", 4), ("gptq", 4), ("gptq", 3), ("gptq", 2))]


@pytest.mark.parametrize("fmt,bits,n,k", FMA_CASES, ids=["%s%d-%dx%d" % c for c in FMA_CASES])
def test_fullsize_reference_formats_fma_arithmetic(fmt, bits, n, k):
    import numpy as np
    from amq_amd import ops
    from oracle import awq_ref, gptq_ref
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + 3 * k + bits)
    q = rng.integers(0, 2 ** bits, size=(n, k), dtype=np.uint8)
    s = (rng.uniform(0.75, 1.25, size=(k // 128, n)) * 0.5 / (np.sqrt(k) * np.sqrt((4.0 ** bits - 1) / 12))).astype(np.float16)
    z = rng.uniform((2 ** bits - 1) / 2 - 0.5, (2 ** bits - 1) / 2 + 0.5, size=(k // 128, n)).astype(np.float16)
    if fmt == "awq":
        qweight = awq_ref.pack_intweight(q)
        scaled_zeros = (-(z.astype(np.float32) * s.astype(np.float32))).astype(np.float16)
        qn, mn = ops.repack_from_awq(torch.from_numpy(qweight).to(dev), torch.from_numpy(s).to(dev),
                                     torch.from_numpy(scaled_zeros).to(dev), n, k)
        w_ref = awq_ref.dequant_kernel(qweight, s, scaled_zeros)
    else:
        qweight = gptq_ref.pack_qweight(q, bits)
        scales = s.astype(np.float32)
        zeros = (z.astype(np.float16) * s).astype(np.float32)                  # GPTQLinear.zeros = fp16(z * s) kept as fp32
        qn, mn = ops.repack_from_gptq(torch.from_numpy(qweight).to(dev), torch.from_numpy(scales).to(dev),
                                      torch.from_numpy(zeros).to(dev), bits, n, k)
        w_ref = gptq_ref.dequant_kernel(qweight, scales, zeros, bits)
    w = ops.dequantize(qn, mn, bits, ops.MODE_FMA, n, k)
    assert np.array_equal(w.cpu().numpy().view(np.uint16), w_ref.view(np.uint16)), "weights differ from the reference kernels' fma dequant"
    x = torch.randn(1, k, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).half()
    y = ops.gemv(x, qn, mn, bits, ops.MODE_FMA, n, k)
    xm = torch.cat([x, torch.randn(16, k, device=dev).half()])
    ym = ops.gemm(xm, qn, mn, bits, ops.MODE_FMA, n, k)
    rms = y.float().pow(2).mean().sqrt()
    assert (y[0].float() - ym[0].float()).abs().max() <= 2.0 ** -9 * y.float().abs().max() + 1e-3 * rms
    ref = x.float() @ w.float().t()
    assert torch.all((y.float() - ref).abs() <= 1e-3 * ref.abs() + 1e-3 * rms)
    y2 = ops.gemv(x * 2, qn, mn, bits, ops.MODE_FMA, n, k)
    normal = y.float().abs() >= 2.0 ** -13
    assert torch.equal(y2[normal], (y * 2)[normal])
    assert torch.equal(ops.gemv(x, qn, mn, bits, ops.MODE_FMA, n, k), y)
    # the module built from the same buffers dispatches to the same kernels
    if fmt == "awq":
        from amq_amd.quant_linear import HIPQuantLinear
        mod = HIPQuantLinear.from_ft_buffers(torch.from_numpy(qweight).to(dev), torch.from_numpy(s).to(dev),
                                             torch.from_numpy(scaled_zeros).to(dev))
        assert mod.mode in (ops.MODE_FMA, ops.MODE_FMA1) and torch.equal(mod(x), y)


@pytest.mark.parametrize("bits,n,k", [(3, 13824, 5120), (4, 5120, 13824), (2, 1024, 8192), (3, 8192, 28672),
                                      (3, 14336, 4096), (2, 4096, 14336), (4, 18944, 3584), (3, 3584, 18944), (2, 512, 3584), (3, 29568, 8192)])
def test_13b_70b_shapes_meet_the_oracle_directly(bits, n, k):
    """VERDICT r4 (parity soft spot): "no 13B / 70B shape ever meets the oracle directly" -- the property tests above trust the HIP dequantize
    kernel.  Here one layer per large shape class (13B gate_proj and down_proj, 70B GQA k_proj and down_proj: 28672 columns) goes through the
    ORACLE on the host (numpy restatement of Quantizer.dequantize, hqq/core/quantize.py:184-199; ~20 s for the 235 M weights of the last case):
    the repacked payload dequantizes to the oracle's weights bit for bit, and the GEMV (1 and 5 rows) and the few-row GEMM (40 rows) match
    nn.Linear on those weights within the bar.  Round 6: the shape classes of the reference's other families too -- Llama-3.x / Mistral
    (14336-wide MLP), Qwen2.5-7B (K = 3584 = 28 x 128, 18944-wide MLP, 512-row k/v), Qwen2.5-72B (29568 x 8192)."""
    import numpy as np
    from amq_amd import ops
    from amq_amd.hqq_format import random_hqq
    from oracle import hqq_ref, linear_ref
    dev = torch.device("cuda:0")
    h = random_hqq(n, k, bits, seed=bits * 7 + (n + k) % 97)
    w_ref = np.asarray(hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), bits, (n, k)), np.float16)
    hd = h.to(dev)
    qn, mn = ops.repack_from_hqq(hd.W_q, hd.scale.reshape(-1), hd.zero.reshape(-1), bits, n, k)
    w_gpu = ops.dequantize(qn, mn, bits, ops.MODE_HQQ, n, k).cpu().numpy()
    assert np.array_equal(w_gpu.view(np.uint16), w_ref.view(np.uint16))
    del w_gpu
    for rows in (1, 5, 40):
        x = torch.randn(rows, k, generator=torch.Generator().manual_seed(rows)).to(torch.float16)
        y_ref = linear_ref.linear_f16(x.numpy(), w_ref).astype(np.float32)
        xg = x.to(dev)
        few = rows <= min(8, ops.gemv_max_rows(k, plain=True, norm=False))         # (K = 28672: the GEMV stages at most 2 rows in LDS)
        y = (ops.gemv(xg, qn, mn, bits, ops.MODE_HQQ, n, k) if few else ops.gemm(xg, qn, mn, bits, ops.MODE_HQQ, n, k)).float().cpu().numpy()
        rms = float(np.sqrt(np.mean(y_ref.astype(np.float64) ** 2)))
        assert np.all(np.abs(y - y_ref) <= 1e-3 * np.abs(y_ref) + 1e-3 * rms), (rows, float(np.abs(y - y_ref).max()))


@pytest.mark.parametrize("name", ["Llama-3.1-8B", "Mistral-7B-v0.3", "Qwen2.5-7B"])
def test_family_block_at_real_width(name):
    """One decoder block of the reference's other families at the models' REAL widths and vocabularies (Llama-3.1-8B: 14336-wide MLP, GQA 4,
    llama3 rope scaling, 128,256 tokens; Mistral-7B-v0.3: theta 1e6, 32,768 tokens; Qwen2.5-7B: hidden 3584, GQA 7, q / k / v bias, eps 1e-6,
    152,064 tokens): the runner's prompt pass and captured token step against the same block computed with torch ops on the weights the
    bit-exact dequantize kernel returns (HF's formulas: fp32 norm statistics, fp16 rotation from the fp32 cos / sin of the rotary embedding's
    own frequencies, softmax in fp32)."""
    from amq_amd import arch, ops
    from amq_amd.llama import QuantLlama
    dev = torch.device("cuda:0")
    cfg = dict(arch.MODEL_CONFIGS[name])
    cfg["n_block"] = 1
    al = {n_: [b] for n_, b in zip(cfg["linear"], [4, 2, 3, 3, 2, 4, 3])}
    m = QuantLlama(cfg, al, max_seq=640, seed=3)
    assert (m.inv_freq is not None) == (name == "Llama-3.1-8B") and m.has_bias == (name == "Qwen2.5-7B")
    blk = m.blocks[0]
    W = {n_: ops.dequantize(blk[n_].qn, blk[n_].mn, blk[n_].bits, blk[n_].mode, blk[n_].N, blk[n_].K).float() for n_ in cfg["linear"]}
    H, nh, nkv, eps = m.H, m.nh, m.nkv, m.eps
    inv = m.inv_freq.to(dev) if m.inv_freq is not None else 1.0 / (m.theta ** (torch.arange(0, 128, 2, device=dev).float() / 128))

    def lin(x, n_):
        y = (x.float() @ W[n_].t()).half()
        return y if blk[n_].bias is None else y + blk[n_].bias

    def norm(x, g):
        xf = x.float()
        return g * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).half()

    def rope(t, pos):
        fr = pos.float()[:, None] * inv[None, :]
        emb = torch.cat([fr, fr], -1)
        cos, sin = emb.cos().half()[:, None, :], emb.sin().half()[:, None, :]
        return t * cos + torch.cat([-t[..., 64:], t[..., :64]], -1) * sin

    def reference(ids, pos0=0):
        S = ids.numel()
        x = m.embed[ids]
        pos = torch.arange(pos0, pos0 + S, device=dev)
        h = norm(x, blk["ln1"])
        q = rope(lin(h, "self_attn.q_proj").view(S, nh, 128), pos)
        k = rope(lin(h, "self_attn.k_proj").view(S, nkv, 128), pos)
        v = lin(h, "self_attn.v_proj").view(S, nkv, 128)
        kk, vv = k.repeat_interleave(nh // nkv, 1), v.repeat_interleave(nh // nkv, 1)
        sc = torch.einsum("shd,thd->hst", q.float(), kk.float()) / 128 ** 0.5
        sc = sc + torch.full((S, S), float("-inf"), device=dev).triu(1)
        p = torch.softmax(sc, -1).half()
        a = torch.einsum("hst,thd->shd", p.float(), vv.float()).half().reshape(S, H)
        x = x + lin(a, "self_attn.o_proj")
        h2 = norm(x, blk["ln2"])
        x = x + lin(torch.nn.functional.silu(lin(h2, "mlp.gate_proj")) * lin(h2, "mlp.up_proj"), "mlp.down_proj")
        return (norm(x, m.norm).float() @ m.lm_head.float().t())

    S = 600 if name == "Llama-3.1-8B" else 48              # (Llama-3.1: positions where the scaled and the plain frequencies have drifted apart)
    ids = torch.randint(0, m.vocab, (S + 3,), generator=torch.Generator().manual_seed(5)).to(dev)
    want = reference(ids)                                   # logits of every row of the (S + 3)-token sequence
    scale = want[S - 1:].abs().max()
    lg = m.prefill(ids[:S]).float()
    assert torch.isfinite(lg).all() and (lg - want[S - 1]).abs().max() <= 1e-2 * scale
    for i in range(3):                                      # three captured token steps, fed the sequence's own tokens
        m.set_token(int(ids[S + i]))
        m.decode_step()
        assert (m.logits.float() - want[S + i]).abs().max() <= 1e-2 * scale, i
    assert m.graph is not None
    if name == "Llama-3.1-8B":
        # the table the rotating kernels read IS the scaled one: against torch on the same frequencies (HF's expression: fp32 product, fp32 cos / sin,
        # one rounding to fp16) and against the plain rope_theta table, which differs wherever the scaled wavelengths apply
        tab = m.rope_tab.view(640, 64, 2).float()
        fr = torch.arange(640, device=dev).float()[:, None] * inv[None, :]
        assert (tab[..., 0] - fr.cos().half().float()).abs().max() <= 2.0 ** -10 and (tab[..., 1] - fr.sin().half().float()).abs().max() <= 2.0 ** -10
        assert ((tab[..., 0] != fr.cos().half().float()).float().mean()) < 2e-3
        plain = ops.rope_table(640, m.theta, dev).view(640, 64, 2).float()
        assert (tab - plain).abs().max() > 0.1
