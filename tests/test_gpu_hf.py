"""The reference's actual caller shape (VERDICT r2 missing #2): an HF ``LlamaForCausalLM`` whose seven linears per block are
swapped by ``prepare_for_inference`` and then driven by HF's own forward (amq/amq_speed_benchmark.py:137-139, 152, 231-256;
hqq/utils/patching.py:143-223).  A tiny random model (no network); the oracle side is the SAME HF model with nn.Linear layers
holding the oracle's dequantized fp16 weights."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")


def _tiny_llama(n_kv_heads):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=n_kv_heads, vocab_size=1000, max_position_embeddings=256, rms_norm_eps=1e-5,
                      attn_implementation="eager")
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).to(torch.float16).to("cuda:0").eval()


def _quantize_linears(model, bits_cycle=(4, 2, 3, 3, 2, 4, 3), group=128):
    """replace every decoder linear by an HQQ stand-in (random HQQ weights of the layer's shape); returns the model and a
    reference copy whose nn.Linear weights are the ORACLE's dequantized weights"""
    from amq_amd.hqq_format import random_hqq
    from amq_amd.patching import HQQWeightsModule
    from oracle import hqq_ref
    ref = copy.deepcopy(model)
    names = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
    i = 0
    for layer, rlayer in zip(model.model.layers, ref.model.layers):
        for parent, rparent in ((layer.self_attn, rlayer.self_attn), (layer.mlp, rlayer.mlp)):
            for name in names:
                lin = getattr(parent, name, None)
                if lin is None:
                    continue
                n, k = lin.weight.shape
                bits = bits_cycle[i % len(bits_cycle)]
                h = random_hqq(n, k, bits, seed=100 + i, group=group)
                i += 1
                w = hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), bits, (n, k), group_size=group)
                getattr(rparent, name).weight.data = torch.from_numpy(w.astype(np.float16)).to("cuda:0")
                setattr(parent, name, HQQWeightsModule(h.to(torch.device("cuda:0"))))
    return model, ref


@pytest.mark.parametrize("n_kv_heads", [2, 1])
@pytest.mark.parametrize("seq", [1, 5, 24])
def test_hf_llama_with_swapped_linears_matches_oracle_weights(n_kv_heads, seq):
    from amq_amd.patching import prepare_for_inference
    from amq_amd.quant_linear import HIPQuantLinear
    model, ref = _quantize_linears(_tiny_llama(n_kv_heads))
    prepare_for_inference(model, backend="hip")
    from amq_amd.quant_linear import HIPLlamaMLP
    assert sum(isinstance(m, HIPLlamaMLP) for m in model.modules()) == 2          # both decoder MLPs fused (SiLU-gated, bias-free)
    mods = [m for m in model.modules() if isinstance(m, HIPQuantLinear)]
    assert len(mods) == 14 and all("_group" in m.__dict__ for m in mods if m.name in ("q_proj", "k_proj", "v_proj", "gate_proj", "up_proj"))
    ids = torch.randint(0, 1000, (1, seq), generator=torch.Generator().manual_seed(seq)).to("cuda:0")
    with torch.inference_mode():
        y = model(ids).logits.float()
        y_ref = ref(ids).logits.float()
    assert torch.isfinite(y).all()
    # HF's own fp16 matmuls on the oracle weights accumulate in a different order: logits agree to fp16 rounding of the stack
    # (measured on MI355X: <= 8e-4 of the largest logit over both head layouts and 1 / 5 / 24 rows; the bar is 2.5 x that)
    assert (y - y_ref).abs().max() <= 2e-3 * y_ref.abs().max()
    from amq_amd.quant_linear import HIPRMSNorm
    assert sum(isinstance(m, HIPRMSNorm) for m in model.modules()) == 4           # both norms of both layers fused into their consumers
    # grouped and ungrouped swaps are the same function, bit for bit (one launch vs three: same kernel per segment)
    model2, _ = _quantize_linears(_tiny_llama(n_kv_heads))
    prepare_for_inference(model2, backend="hip", group_siblings=False, fuse_mlp=False, fuse_layers=False)      # the plain module swap
    assert not any("forward" in l.__dict__ for l in model2.model.layers)
    model3, _ = _quantize_linears(_tiny_llama(n_kv_heads))
    prepare_for_inference(model3, backend="hip", fuse_norms=False)                # grouped, fused MLP, residual adds in the epilogues
    assert not any(isinstance(m, HIPRMSNorm) for m in model3.modules())
    assert all("forward" in l.__dict__ for l in model3.model.layers) and all("forward" in l.__dict__ for l in model.model.layers)
    with torch.inference_mode():
        y2 = model2(ids).logits.float()
        y3 = model3(ids).logits.float()
    assert torch.equal(y3, y2)          # (the residual adds formed in the o_proj / down_proj epilogues round like the separate adds)
    # the fused norms replace HF's torch-op RMSNorm by the GEMV prologue for <= 8 rows (same formula, fp32 statistics, another
    # summation order); more rows run HF's module itself
    if seq > 8:
        assert torch.equal(y, y2)
    else:
        assert (y - y2).abs().max() <= 4e-3 * y2.abs().max()


@pytest.mark.parametrize("group", [64, 32])
@pytest.mark.parametrize("seq", [1, 5, 24])
def test_hf_llama_with_finer_group_layers(group, seq):
    """layers quantized with groups of 64 / 32 (HQQ's default group_size is 64) through the same drop-in flow: the sibling groups, the fused
    MLP / norms / residual epilogues are the same launches (amq_gemv_grouped_f16 takes the group size); 24 rows: the GEMV kernel reaches 16,
    the rest is dequantize-once + the fp16 GEMM per linear"""
    from amq_amd.patching import prepare_for_inference
    from amq_amd.quant_linear import HIPQuantLinear, HIPLlamaMLP, HIPRMSNorm
    model, ref = _quantize_linears(_tiny_llama(2), group=group)
    prepare_for_inference(model, backend="hip")
    mods = [m for m in model.modules() if isinstance(m, HIPQuantLinear)]
    assert len(mods) == 14 and all(m.group_size == group and m.native_group == group for m in mods)
    assert all("_group" in m.__dict__ for m in mods if m.name in ("q_proj", "k_proj", "v_proj", "gate_proj", "up_proj"))
    assert sum(isinstance(m, HIPLlamaMLP) for m in model.modules()) == 2 and sum(isinstance(m, HIPRMSNorm) for m in model.modules()) == 4
    ids = torch.randint(0, 1000, (1, seq), generator=torch.Generator().manual_seed(seq)).to("cuda:0")
    with torch.inference_mode():
        y = model(ids).logits.float()
        y_ref = ref(ids).logits.float()
    assert torch.isfinite(y).all() and (y - y_ref).abs().max() <= 2e-3 * y_ref.abs().max()
    model2, _ = _quantize_linears(_tiny_llama(2), group=group)
    prepare_for_inference(model2, backend="hip", group_siblings=False, fuse_mlp=False, fuse_layers=False)      # the plain module swap
    model3, _ = _quantize_linears(_tiny_llama(2), group=group)
    prepare_for_inference(model3, backend="hip", fuse_norms=False)
    with torch.inference_mode():
        y2 = model2(ids).logits.float()
        y3 = model3(ids).logits.float()
    assert torch.equal(y3, y2)
    if seq > 8:
        assert torch.equal(y, y2)
    else:
        assert (y - y2).abs().max() <= 4e-3 * y2.abs().max()
    # greedy decode through HF's generate on the prepared model == on the plain swap
    with torch.inference_mode():
        t1 = model3.generate(ids, min_new_tokens=6, max_new_tokens=6, do_sample=False, num_beams=1)
        t2 = model2.generate(ids, min_new_tokens=6, max_new_tokens=6, do_sample=False, num_beams=1)
    assert torch.equal(t1, t2)


def test_kernel_arithmetic_option_matches_the_reference_kernels_weights():
    """prepare_for_inference(kernel_arithmetic=True): every swapped linear decodes to the weights the reference's GPTQ kernels decode to for the same
    HQQ layer (patch_hqq_to_gptq's buffers through the oracle's restatement of the kernel dequant), bit for bit; the HF model on those weights gives the
    same logits to fp16 rounding; the default (HQQ arithmetic) model is within one weight ulp of it; greedy tokens of the fused and plain swaps agree"""
    from amq_amd import ops
    from amq_amd.hqq_format import random_hqq
    from amq_amd.patching import prepare_for_inference
    from amq_amd.quant_linear import HIPQuantLinear
    from oracle import hqq_ref, gptq_ref
    model, ref = _quantize_linears(_tiny_llama(2))
    # the oracle side: GPTQLinear.pack(W_deq, scales, zeros) -> kernel dequant, per layer (same seeds as _quantize_linears)
    names = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
    bits_cycle, i = (4, 2, 3, 3, 2, 4, 3), 0
    kernel_w = {}
    for li, rlayer in enumerate(ref.model.layers):
        for parent in (rlayer.self_attn, rlayer.mlp):
            for name in names:
                lin = getattr(parent, name, None)
                if lin is None:
                    continue
                n, k = lin.weight.shape
                bits = bits_cycle[i % 7]
                h = random_hqq(n, k, bits, seed=100 + i)
                i += 1
                w_deq = np.asarray(hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), bits, (n, k)), np.float16)
                qw, sc, zr = gptq_ref.pack(w_deq, h.scale.numpy().reshape(n, -1), h.zero.numpy().reshape(n, -1), bits)
                wk = np.asarray(gptq_ref.dequant_kernel(qw, sc, zr, bits), np.float16)
                kernel_w[(li, name)] = wk
                lin.weight.data = torch.from_numpy(wk).to("cuda:0")
    prepare_for_inference(model, backend="hip", kernel_arithmetic=True)
    mods = {(li, m.name): m for li, layer in enumerate(model.model.layers) for m in layer.modules() if isinstance(m, HIPQuantLinear)}
    assert len(mods) == 14 and all(m.mode in (ops.MODE_FMA, ops.MODE_FMA1) for m in mods.values())
    for key, m in mods.items():
        assert np.array_equal(m.dequantize().cpu().numpy().view(np.uint16), kernel_w[key].view(np.uint16)), key
    ids = torch.randint(0, 1000, (1, 5), generator=torch.Generator().manual_seed(5)).to("cuda:0")
    with torch.inference_mode():
        y = model(ids).logits.float()
        y_ref = ref(ids).logits.float()
    assert (y - y_ref).abs().max() <= 2e-3 * y_ref.abs().max()
    model2, _ = _quantize_linears(_tiny_llama(2))
    prepare_for_inference(model2, backend="hip", group_siblings=False, fuse_mlp=False, fuse_layers=False, kernel_arithmetic=True)
    model3, _ = _quantize_linears(_tiny_llama(2))
    prepare_for_inference(model3, backend="hip")                              # HQQ arithmetic: a weight ulp away
    with torch.inference_mode():
        t1 = model.generate(ids, min_new_tokens=6, max_new_tokens=6, do_sample=False, num_beams=1)
        t2 = model2.generate(ids, min_new_tokens=6, max_new_tokens=6, do_sample=False, num_beams=1)
        y3 = model3(ids).logits.float()
    assert torch.equal(t1, t2)
    assert (y3 - y).abs().max() <= 1e-2 * y.abs().max() and not torch.equal(y3, y)
    # idempotent, and a converted module keeps working after a state_dict round trip
    m = next(iter(mods.values()))
    before = m.meta.clone()
    assert m.to_kernel_arithmetic() is m and torch.equal(m.meta, before)
    m2 = HIPQuantLinear(m.bits, m.group_size, m.infeatures, m.outfeatures).to("cuda:0")
    m2.load_state_dict({k: v for k, v in m.state_dict().items() if k != "weight"})      # (the dummy .weight HF code queries is not a buffer of the module)
    x = torch.randn(2, m.infeatures, device="cuda:0").half()
    assert m2.mode == m.mode and torch.equal(m2(x), m(x))


def test_deferred_norm_that_is_not_consumed_raises():
    """HIPRMSNorm hands its raw input on; if the grouped launch it was fused into never runs, the next forward fails loudly"""
    from amq_amd.patching import prepare_for_inference
    from amq_amd.quant_linear import HIPRMSNorm
    model, _ = _quantize_linears(_tiny_llama(2))
    prepare_for_inference(model, backend="hip")
    layer = model.model.layers[0]
    assert isinstance(layer.input_layernorm, HIPRMSNorm) and isinstance(layer.post_attention_layernorm, HIPRMSNorm)
    x = torch.randn(1, 1, 256, device="cuda:0").half()
    with torch.inference_mode():
        assert layer.input_layernorm(x) is x                      # deferred: the q/k/v launch will normalise
        with pytest.raises(RuntimeError, match="never consumed"):
            layer.input_layernorm(x)
        layer.self_attn.q_proj.__dict__["_group"][0].__dict__["_norm"] = None
        layer.input_layernorm(x)
        with pytest.raises(RuntimeError, match="different tensor"):
            layer.self_attn.q_proj(torch.randn(1, 1, 256, device="cuda:0").half())
        # many rows are not deferred: HF's own module runs
        xm = torch.randn(1, 24, 256, device="cuda:0").half()
        assert layer.post_attention_layernorm(xm) is not xm


def test_fused_norm_equals_rmsnorm_kernel_then_grouped_launch():
    """the RMSNorm prologue of the grouped GEMV is the arithmetic of amq_rmsnorm_f16: deferred norm + q/k/v == rmsnorm kernel + q/k/v"""
    from amq_amd import ops
    from amq_amd.patching import prepare_for_inference
    model, _ = _quantize_linears(_tiny_llama(2))
    prepare_for_inference(model, backend="hip")
    layer = model.model.layers[1]
    for rows in (1, 3, 8):
        x = torch.randn(1, rows, 256, generator=torch.Generator().manual_seed(rows)).half().to("cuda:0")
        with torch.inference_mode():
            h = layer.input_layernorm(x)
            got = [p(h) for p in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj)]
            hn = ops.rmsnorm(x.view(rows, 256), layer.input_layernorm.weight.data, layer.input_layernorm.variance_epsilon).view(1, rows, 256)
            want = [p(hn) for p in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj)]
            for a, b in zip(got, want):
                assert torch.equal(a, b)
            y = layer.mlp(layer.post_attention_layernorm(x))
            yn = layer.mlp(ops.rmsnorm(x.view(rows, 256), layer.post_attention_layernorm.weight.data,
                                       layer.post_attention_layernorm.variance_epsilon).view(1, rows, 256))
            assert torch.equal(y, yn)


def test_hf_swapped_model_survives_deepcopy_and_state_dict_roundtrip(tmp_path):
    """amq_speed_benchmark.py:231 deep-copies the patched model; patching.py:178-208 caches its state_dict"""
    from amq_amd.patching import prepare_for_inference
    model, _ = _quantize_linears(_tiny_llama(2))
    cache = str(tmp_path / "tiny_HIPLinear.pt")
    prepare_for_inference(model, backend="hip", load_path=cache)          # writes the cache
    ids = torch.randint(0, 1000, (1, 3), generator=torch.Generator().manual_seed(9)).to("cuda:0")
    with torch.inference_mode():
        y = model(ids).logits.clone()
        y_copy = copy.deepcopy(model)(ids).logits
    assert torch.equal(y, y_copy)
    fresh, _ = _quantize_linears(_tiny_llama(2))
    prepare_for_inference(fresh, backend="hip", load_path=cache)          # loads it instead of re-packing
    with torch.inference_mode():
        assert torch.equal(fresh(ids).logits, y)
    assert any(k.endswith("q_proj.qweight") for k in model.state_dict())  # per-linear keys, as in the reference's caches


@pytest.mark.parametrize("n_kv_heads", [2, 1])
def test_hf_greedy_loop_with_past_key_values(n_kv_heads):
    """the reference's non-FT decode protocol (amq/utils/speed.py:93-125): ``model(ids, past_key_values=...)`` for the prompt, then one
    token at a time fed back with HF's own KV cache -- through the swapped + fused model and through the same HF model on the oracle's
    dequantized weights.  The single-row steps take every fused path (grouped q/k/v with the deferred norm, o_proj / down_proj with
    the residual in the epilogue, fused MLP); the fused model is bit-identical to the plain module swap at every step."""
    from amq_amd.patching import prepare_for_inference
    model, ref = _quantize_linears(_tiny_llama(n_kv_heads))
    prepare_for_inference(model, backend="hip")
    plain, _ = _quantize_linears(_tiny_llama(n_kv_heads))
    prepare_for_inference(plain, backend="hip", group_siblings=False, fuse_mlp=False, fuse_layers=False)
    ids = torch.randint(0, 1000, (1, 6), generator=torch.Generator().manual_seed(11)).to("cuda:0")

    def greedy(m, steps=5):
        outs, cur, past = [], ids, None
        with torch.inference_mode():
            for _ in range(steps):
                o = m(cur, past_key_values=past, use_cache=True)
                past = o.past_key_values
                outs.append(o.logits[:, -1].float().clone())
                cur = o.logits[:, -1].max(1)[1].unsqueeze(1)                       # speed.py:121-122
        return outs

    got, want, base = greedy(model), greedy(ref), greedy(plain)
    for step, (a, b, c) in enumerate(zip(got, want, base)):
        assert torch.isfinite(a).all()
        if step == 0:
            assert torch.equal(a, c)                     # 6 prompt rows: deferred norms do not apply; residual epilogues round like the adds
        else:
            assert (a - c).abs().max() <= 4e-3 * c.abs().max()      # single rows: the norm is formed in the GEMV prologue (another summation order)
        assert (a - b).abs().max() <= 1e-2 * b.abs().max(), step   # HF's fp16 matmuls on the oracle weights; error compounds over the cached steps


@pytest.mark.parametrize("n_kv_heads", [2, 1])
def test_runner_from_swapped_hf_model(n_kv_heads):
    """QuantLlama.from_hf: the hipGraph token-step runner built over a swapped HF model (sharing its buffers) generates what HF's own
    forward generates from the same modules, and what the HF model on the oracle's dequantized weights generates -- the counterpart
    of switching the reference's benchmark to ``use_ft`` (amq_speed_benchmark.py:152; kernel/monkeypatch/ftllama_modeling.py)"""
    from amq_amd.llama import QuantLlama
    from amq_amd.patching import prepare_for_inference
    model, ref = _quantize_linears(_tiny_llama(n_kv_heads))
    prepare_for_inference(model, backend="hip")
    r = QuantLlama.from_hf(model, max_seq=64)
    q0 = model.model.layers[0].self_attn.q_proj
    assert r.blocks[0]["self_attn.q_proj"].qn.data_ptr() == q0.qweight.data_ptr()         # shared, not copied
    assert r.embed.data_ptr() == model.model.embed_tokens.weight.data_ptr() and r.nkv == n_kv_heads
    ids = torch.randint(0, 1000, (1, 7), generator=torch.Generator().manual_seed(5)).to("cuda:0")
    steps = 6
    with torch.inference_mode():
        hf_logits, hf_tokens, cur, past = [], [], ids, None
        ref_logits, rcur, rpast = [], ids, None
        for _ in range(steps):
            o = model(cur, past_key_values=past, use_cache=True)
            past, cur = o.past_key_values, o.logits[:, -1].max(1)[1].unsqueeze(1)
            hf_logits.append(o.logits[0, -1].float().clone()); hf_tokens.append(int(cur.item()))
            o = ref(rcur, past_key_values=rpast, use_cache=True)
            rpast, rcur = o.past_key_values, o.logits[:, -1].max(1)[1].unsqueeze(1)
            ref_logits.append(o.logits[0, -1].float().clone())
    got = [r.prefill(ids[0]).float().clone()]
    tokens = [int(r.token.item())]
    for _ in range(steps - 1):
        r.decode_step()
        got.append(r.logits.float().clone()); tokens.append(int(r.token.item()))
    r.check()
    same = True
    for i in range(steps):
        if same:
            assert (got[i] - hf_logits[i]).abs().max() <= 1e-2 * hf_logits[i].abs().max(), i      # other attention kernels, same weights
            assert (got[i] - ref_logits[i]).abs().max() <= 1e-2 * ref_logits[i].abs().max(), i
        same = same and tokens[i] == hf_tokens[i]
    assert tokens[0] == hf_tokens[0]
    with pytest.raises(ValueError, match="HIPQuantLinear"):
        QuantLlama.from_hf(ref)                                         # nn.Linear layers: not a swapped model


def test_benchmark_speed_takes_the_swapped_hf_model():
    """amq_speed_benchmark.py:152 / 253: ``benchmark_speed(model, ...)`` on the assembled HF model itself; same result schema"""
    from amq_amd.patching import prepare_for_inference
    from amq_amd.speed import benchmark_speed
    model, _ = _quantize_linears(_tiny_llama(2))
    prepare_for_inference(model, backend="hip")
    for mode in ("TPS", "GeMV", "GeMM", "TTFT"):
        r = benchmark_speed(model, iteration=2, sizes=(1, 16, 8), mode=mode, get_peak_memory=(mode == "TPS"))
        assert r[mode.lower()]["1.16.8"] > 0 and (("peak_memory" in r) == (mode == "TPS"))
    # use_ft=False: the reference's non-FT loops (speed.py:22-46, 93-125) over HF's own forward / generate on the swapped model
    for mode in ("TPS", "GeMV", "GeMM", "TTFT"):
        r = benchmark_speed(model, use_ft=False, iteration=2, sizes=(1, 16, 8), mode=mode, get_peak_memory=False)
        assert r[mode.lower()]["1.16.8"] > 0
    r = benchmark_speed(model, use_ft=False, iteration=1, sizes=(2, 16, 4), mode="GeMV", get_peak_memory=False)     # HF's forward takes a batch
    assert r["gemv"]["2.16.4"] > 0


@pytest.mark.parametrize("n_kv_heads", [2, 1])
def test_hf_generate_on_the_swapped_model(n_kv_heads):
    """the reference's TPS call (amq/utils/speed.py:31-36): ``model.generate(ids, min_new_tokens=G, max_new_tokens=G, do_sample=False,
    num_beams=1, attention_mask=...)`` on the prepare_for_inference'd HF model -- HF's generation loop, cache and sampling code over
    the fused modules.  Same tokens as the plain module swap (every fusion is value-preserving up to the norm's summation order) and,
    while the arg-max margins allow, as the same HF model on the oracle's dequantized weights."""
    from amq_amd.patching import prepare_for_inference
    G = 8
    model, ref = _quantize_linears(_tiny_llama(n_kv_heads))
    prepare_for_inference(model, backend="hip")
    plain, _ = _quantize_linears(_tiny_llama(n_kv_heads))
    prepare_for_inference(plain, backend="hip", group_siblings=False, fuse_mlp=False, fuse_layers=False)
    ids = torch.randint(0, 1000, (1, 9), generator=torch.Generator().manual_seed(4)).to("cuda:0")
    kw = dict(min_new_tokens=G, max_new_tokens=G, do_sample=False, num_beams=1, attention_mask=torch.ones_like(ids), pad_token_id=0)
    with torch.inference_mode():
        out = model.generate(ids, **kw)
        out_plain = plain.generate(ids, **kw)
        out_ref = ref.generate(ids, **kw)
        again = model.generate(ids, **kw)
    assert out.shape == (1, 9 + G) and torch.equal(out[:, :9], ids)
    assert torch.equal(out, again)                                       # repeatable: no state leaks between calls (deferred norms all consumed)
    # token-level agreement: identical until a step whose top-2 logit margin is inside the numerical distance of the two stacks
    def first_diff(a, b):
        d = (a[0] != b[0]).nonzero()
        return int(d[0]) if len(d) else a.shape[1]
    assert first_diff(out, out_plain) >= 9 + 1 and first_diff(out, out_ref) >= 9 + 1
    for other, tol in ((out_plain, 4e-3), (out_ref, 3e-2)):
        k = first_diff(out, other)
        if k < 9 + G:                                                   # a divergence must be a near-tie, not an error
            with torch.inference_mode():
                lg = model(out[:, :k]).logits[0, -1].float()
            top2 = lg.topk(2).values
            assert float(top2[0] - top2[1]) <= 2 * tol * float(lg.abs().max()), (k, top2)


def test_prepare_warns_when_a_fusion_step_matches_nothing():
    """a model that holds HIPQuantLinear projections under the Llama names but whose containers the fusion steps do not recognise
    (here: an MLP whose activation is not SiLU, decoder layers of an unknown class) gets ONE warning per step instead of a silent
    fall-back to one launch per projection; a recognised model stays silent."""
    import warnings
    from amq_amd.patching import prepare_for_inference
    model, _ = _quantize_linears(_tiny_llama(2))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        prepare_for_inference(model, backend="hip")                                   # everything fuses: no warning
    odd, _ = _quantize_linears(_tiny_llama(2))
    for layer in odd.model.layers:
        layer.mlp.act_fn = torch.nn.GELU()                                             # not a SiLU-gated MLP any more
    with pytest.warns(RuntimeWarning) as rec:
        prepare_for_inference(odd, backend="hip")
    msgs = " | ".join(str(w.message) for w in rec)
    assert "no MLP was fused" in msgs and "grouped" not in msgs            # (input_layernorm still fuses into q/k/v: no norm warning)
    odd2, _ = _quantize_linears(_tiny_llama(2))
    for layer in odd2.model.layers:                                          # a decoder-layer class the forward patch does not know
        layer.__class__ = type("RefactoredDecoderLayer", (layer.__class__,), {})
    with pytest.warns(RuntimeWarning, match="none took the fused forward"):
        prepare_for_inference(odd2, backend="hip")
    ids = torch.randint(0, 1000, (1, 3), generator=torch.Generator().manual_seed(1)).to("cuda:0")
    with torch.inference_mode():
        assert torch.isfinite(odd(ids).logits.float()).all()                          # (and the model still runs, unfused)


def test_fused_norm_follows_its_consumer_when_siblings_are_replaced():
    """replacing a sibling after prepare_for_inference (or preparing twice) must not leave a fused norm pointing at a group
    that no longer runs: the norm checks its consumer on every forward (falls back to HF's module), a second prepare re-targets it,
    and a forward that dies between the norm and its consumer does not brick the model."""
    from amq_amd.patching import prepare_for_inference
    from amq_amd.quant_linear import HIPRMSNorm
    model, _ = _quantize_linears(_tiny_llama(2))
    prepare_for_inference(model, backend="hip")
    ids = torch.randint(0, 1000, (1, 1), generator=torch.Generator().manual_seed(3)).to("cuda:0")
    with torch.inference_mode():
        y0 = model(ids).logits.float().clone()
    # replace k_proj of layer 0 by a copy of itself (same weights; the copy carries a group of COPIES, not the layer's q/k/v group)
    attn = model.model.layers[0].self_attn
    old = attn.k_proj
    new = copy.deepcopy(old)
    assert new.__dict__["_group"][0] is not old.__dict__["_group"][0]
    attn.k_proj = new
    norm = model.model.layers[0].input_layernorm
    assert isinstance(norm, HIPRMSNorm)
    with torch.inference_mode():
        y1 = model(ids).logits.float().clone()               # stale group: the norm notices and runs HF's module; q/v still group-launch
        y1b = model(ids).logits.float()
    assert torch.equal(y1, y1b)
    assert (y1 - y0).abs().max() <= 4e-3 * y0.abs().max()    # same function (the norm's summation order differs)
    prepare_for_inference(model, backend="hip")              # regroups q/k/v around the new sibling and re-targets the norm
    g = attn.q_proj.__dict__["_group"][0]
    assert g.members[1] is new and norm.__dict__["_consumer"] is g
    with torch.inference_mode():
        y2 = model(ids).logits.float()
    assert torch.equal(y2, y0)                                # fused again: the original bits
    # a forward that dies after the norm deferred: the next forward reports it once, the one after works
    x = torch.randn(1, 1, 256, device="cuda:0").half()
    with torch.inference_mode():
        norm(x)
        with pytest.raises(RuntimeError, match="never consumed"):
            norm(x)
        assert torch.equal(model(ids).logits.float(), y0)


def test_reference_driver_assembly_of_a_mixed_model():
    """amq_speed_benchmark.py:129-256 on tiny models: three uniformly quantized models are prepared one by one and parked on the CPU;
    a fresh fp16 base model is deep-copied and every linear is ``setattr``-ed from the model of its bit-width; the assembled model goes
    to the GPU and into ``benchmark_speed``.  Checked: HF's forward over the assembled model against the same assembly of
    oracle-weight nn.Linear layers; a second ``prepare_for_inference`` regroups the transplanted siblings and fuses the base model's
    layers without changing a bit for many rows; ``QuantLlama.from_hf`` / ``benchmark_speed`` take the assembled model."""
    from amq_amd.llama import QuantLlama
    from amq_amd.patching import prepare_for_inference
    from amq_amd.quant_linear import HIPLlamaMLP, HIPRMSNorm
    from amq_amd.speed import benchmark_speed
    prepared, refs = {}, {}
    for bits in (2, 3, 4):
        m, r = _quantize_linears(_tiny_llama(2), bits_cycle=(bits,))
        prepare_for_inference(m, backend="gptq" if bits < 4 else "ft")           # (the reference's backend names)
        prepared[bits], refs[bits] = m.to("cpu"), r
    base = _tiny_llama(2)
    model, ref = copy.deepcopy(base), copy.deepcopy(base)
    rng = np.random.default_rng(3)
    names = [("self_attn", n) for n in ("q_proj", "k_proj", "v_proj", "o_proj")] + [("mlp", n) for n in ("gate_proj", "up_proj", "down_proj")]
    arch = {n: [int(b) for b in rng.choice([2, 3, 4], size=2)] for n in names}
    arch[("self_attn", "q_proj")][1] = arch[("self_attn", "k_proj")][1] = arch[("self_attn", "v_proj")][1] = 3    # one trio from ONE model
    for li in range(2):
        for (module, linear) in names:
            b = arch[(module, linear)][li]
            for dst, src in ((model, prepared[b]), (ref, refs[b])):
                parent = getattr(dst.model.layers[li], module)
                delattr(parent, linear)                                           # amq_speed_benchmark.py:246-249
                setattr(parent, linear, getattr(getattr(src.model.layers[li], module), linear))
    model = model.eval().to("cuda:0")
    ids = torch.randint(0, 1000, (1, 12), generator=torch.Generator().manual_seed(2)).to("cuda:0")
    with torch.inference_mode():
        y = model(ids).logits.float()
        y1 = model(ids[:, :1]).logits.float()
        y_ref = ref(ids).logits.float()
    assert (y - y_ref).abs().max() <= 2e-3 * y_ref.abs().max()
    assert not any(isinstance(m, (HIPLlamaMLP, HIPRMSNorm)) for m in model.modules())          # the base model's containers: nothing fused yet
    prepare_for_inference(model, backend="hip")                                   # no HQQ layer left to convert: regroup + fuse only
    l1 = model.model.layers[1]
    g = l1.self_attn.q_proj.__dict__["_group"][0]
    assert g.members[1] is l1.self_attn.k_proj and g.members[2] is l1.self_attn.v_proj
    assert isinstance(l1.mlp, HIPLlamaMLP) and isinstance(l1.input_layernorm, HIPRMSNorm) and "forward" in l1.__dict__
    with torch.inference_mode():
        assert torch.equal(model(ids).logits.float(), y)                          # 12 rows: same kernels, same bits
        z1 = model(ids[:, :1]).logits.float()
    assert (z1 - y1).abs().max() <= 4e-3 * y1.abs().max()                         # 1 row: norms now formed in the GEMV prologues
    r = QuantLlama.from_hf(model, max_seq=64)
    assert [r.blocks[1][f"{m}.{n}"].bits for (m, n) in names] == [arch[k][1] for k in names]
    lg = r.prefill(ids[0]).float()
    assert (lg - y[0, -1]).abs().max() <= 1e-2 * y[0, -1].abs().max() and int(lg.argmax()) == int(y[0, -1].argmax())
    out = benchmark_speed(model, iteration=1, sizes=(1, 12, 4), mode="GeMV", get_peak_memory=False)
    assert out["gemv"]["1.12.4"] > 0
