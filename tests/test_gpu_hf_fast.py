"""The reference's ``use_ft`` call surface on the prepared HF object itself and its other model families (item 3).

``convert_model_to_hip(model)`` is to this build what ``convert_model_to_ft(model)`` + ``replace_generate_functions()`` are to the reference
(kernel/monkeypatch/ftllama_modeling.py:569-580, ftllama_generate.py:613-622): the caller's own ``model.generate(...)`` and
``model(ids, start_pos=..., use_cache=False)`` (amq/utils/speed.py:31-36, 65, 82) run the fused static-cache step.  Checked against HF's own
eager path over the same swapped modules and against HF modules holding the oracle's dequantized weights."""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")

DEV = "cuda:0"
NAMES = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def _swap_linears(model, bits_cycle=(4, 2, 3, 3, 2, 4, 3), seed=100):
    """every decoder linear -> an HQQ stand-in of its shape (random HQQ weights; the linear's own bias kept); returns (model, reference model whose
    nn.Linear weights are the oracle's dequantized weights)"""
    from amq_amd.hqq_format import random_hqq
    from amq_amd.patching import HQQWeightsModule
    from oracle import hqq_ref
    ref = copy.deepcopy(model)
    i = 0
    for layer, rlayer in zip(model.model.layers, ref.model.layers):
        for parent, rparent in ((layer.self_attn, rlayer.self_attn), (layer.mlp, rlayer.mlp)):
            for name in NAMES:
                lin = getattr(parent, name, None)
                if lin is None:
                    continue
                n, k = lin.weight.shape
                bits = bits_cycle[i % len(bits_cycle)]
                h = random_hqq(n, k, bits, seed=seed + i)
                i += 1
                w = hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), bits, (n, k))
                getattr(rparent, name).weight.data = torch.from_numpy(w.astype(np.float16)).to(DEV)
                h.bias = None if lin.bias is None else lin.bias.data.detach().clone()
                setattr(parent, name, HQQWeightsModule(h.to(torch.device(DEV))))
    return model, ref


def _tiny(family, layers=2):
    """tiny random models of the three families (head_dim 128)"""
    torch.manual_seed(0)
    if family == "llama":
        from transformers import LlamaConfig, LlamaForCausalLM
        cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=2, num_key_value_heads=1,
                          vocab_size=1000, max_position_embeddings=256, rms_norm_eps=1e-5, attn_implementation="eager")
        m = LlamaForCausalLM(cfg)
    elif family == "llama3.1":
        from transformers import LlamaConfig, LlamaForCausalLM
        cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=2, num_key_value_heads=1,
                          vocab_size=1000, max_position_embeddings=512, rms_norm_eps=1e-5, attn_implementation="eager", rope_theta=500000.0,
                          rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                        "original_max_position_embeddings": 64})
        m = LlamaForCausalLM(cfg)
    elif family == "mistral":
        from transformers import MistralConfig, MistralForCausalLM
        cfg = MistralConfig(hidden_size=512, intermediate_size=768, num_hidden_layers=layers, num_attention_heads=4, num_key_value_heads=1,
                            vocab_size=1000, max_position_embeddings=256, rms_norm_eps=1e-5, attn_implementation="eager", sliding_window=None,
                            rope_theta=1000000.0, head_dim=128)
        m = MistralForCausalLM(cfg)
    else:
        from transformers import Qwen2Config, Qwen2ForCausalLM
        cfg = Qwen2Config(hidden_size=896, intermediate_size=640, num_hidden_layers=layers, num_attention_heads=7, num_key_value_heads=1,
                          vocab_size=1000, max_position_embeddings=256, rms_norm_eps=1e-6, attn_implementation="eager", rope_theta=1000000.0,
                          tie_word_embeddings=False)
        m = Qwen2ForCausalLM(cfg)
        with torch.no_grad():                                    # (HF initialises biases to zero: give them values)
            for layer in m.model.layers:
                for nm in ("q_proj", "k_proj", "v_proj"):
                    getattr(layer.self_attn, nm).bias.normal_(0.0, 0.1)
    return m.to(torch.float16).to(DEV).eval()


def _prepared(family, layers=2):
    from amq_amd.patching import prepare_for_inference
    model, ref = _swap_linears(_tiny(family, layers))
    prepare_for_inference(model, backend="hip")
    return model, ref


def _hf_generate(model, ids, n):
    return model.generate(ids, min_new_tokens=n, max_new_tokens=n, do_sample=False, num_beams=1, attention_mask=torch.ones_like(ids),
                          pad_token_id=0)


def _assert_same_tokens_or_a_tie(model, fast, slow, n_prompt, eos=None):
    """two fp16 implementations of one function decode greedily: equal tokens, or -- at the first step where they part -- a near-tie under HF's own
    logits for the common prefix (the tiny random models settle into flat, repetitive distributions where the two best logits are rounding apart);
    what follows a parted step is not comparable"""
    if torch.equal(fast, slow):
        return
    for b in range(fast.shape[0]):
        diff = (fast[b] != slow[b]).nonzero()
        if len(diff) == 0:
            continue
        t = int(diff[0])
        assert t >= n_prompt
        with torch.inference_mode():
            lg = model(slow[b:b + 1, :t]).logits[0, -1].float()          # (no start_pos: HF's own forward over the same modules)
            if eos is not None:
                lg[eos] = float("-inf")
        gap = float(lg.max() - lg[int(fast[b, t])])
        assert gap <= 4e-3 * float(lg[torch.isfinite(lg)].abs().max()), (b, t, gap, fast[b, n_prompt:].tolist(), slow[b, n_prompt:].tolist())


def _golden_hf(bits=3):
    """an HF LlamaForCausalLM over a tiny checkpoint the REFERENCE wrote (tests/golden/ckpt: quantize_model + save_quantized)"""
    from transformers import LlamaConfig, LlamaForCausalLM
    from amq_amd.checkpoint import load_hqq_dir
    from amq_amd.hqq_format import HQQWeights
    from amq_amd.patching import HQQWeightsModule, prepare_for_inference
    root = os.path.join(os.path.dirname(__file__), "golden", "ckpt", f"{bits}bit")
    hf, mods = load_hqq_dir(root)
    cfg = LlamaConfig(**{k: v for k, v in hf.items() if k not in ("architectures", "transformers_version", "model_type")})
    cfg._attn_implementation = "eager"
    model = LlamaForCausalLM(cfg).to(torch.float16).to(DEV).eval()
    for name, w in mods.items():
        parent = model
        parts = name.split(".")
        for p in parts[:-1]:
            parent = getattr(parent, p)
        if isinstance(w, HQQWeights):
            w.name = parts[-1]
            setattr(parent, parts[-1], HQQWeightsModule(w.to(torch.device(DEV))))
        elif "weight" in w:
            getattr(parent, parts[-1]).weight.data = w["weight"].to(torch.float16).to(DEV)
    prepare_for_inference(model, backend="hip")
    return model


def test_generate_on_the_converted_object_equals_hf_eager_generate():
    """the caller's own model.generate(...) -- the reference harness' TPS loop (speed.py:31-36) -- over reference-written checkpoints and a random
    GQA model: the fast path returns the tokens HF's eager generate returns over the same modules, for 1 and 2 sequences"""
    from amq_amd import hf_fast
    for model, vocab in ((_golden_hf(3), 256), (_prepared("llama")[0], 1000)):
        for B in (1, 2):
            ids = torch.randint(1, vocab, (B, 9), generator=torch.Generator().manual_seed(5 + B)).to(DEV)
            with torch.inference_mode():
                slow = _hf_generate(model, ids, 12)
            hf_fast.convert_model_to_hip(model)
            assert hf_fast.convert_model_to_hip(model) is model                  # idempotent
            with torch.inference_mode():
                fast = _hf_generate(model, ids, 12)
                fast2 = _hf_generate(model, ids, 12)
            assert fast.shape == (B, 21) and fast.dtype == ids.dtype and torch.equal(fast[:, :9], ids)
            assert torch.equal(fast, fast2)
            _assert_same_tokens_or_a_tie(model, fast, slow, 9, model.generation_config.eos_token_id)
            assert torch.equal(fast[:, :9 + 3], slow[:, :9 + 3])
            assert B in hf_fast._RUNNERS[model]                                   # (it really was the runner)
            hf_fast.revert_model_to_hf(model)
            assert "forward" not in model.__dict__ and "generate" not in model.__dict__ and model not in hf_fast._RUNNERS
            # min_new_tokens = max_new_tokens: HF never emits an EOS id (MinNewTokensLengthLogitsProcessor); declare a token the free run DID pick
            # as EOS -- both paths must now avoid it, in the same way
            eos = int(slow[0, 9 + 3])
            model.generation_config.eos_token_id = eos
            with torch.inference_mode():
                slow_e = _hf_generate(model, ids, 12)
            hf_fast.convert_model_to_hip(model)
            with torch.inference_mode():
                fast_e = _hf_generate(model, ids, 12)
            assert eos not in fast_e[:, 9:].tolist()[0] and eos not in slow_e[:, 9:].tolist()[0] and not torch.equal(fast_e, fast)
            _assert_same_tokens_or_a_tie(model, fast_e, slow_e, 9, eos)
            assert torch.equal(fast_e[:, :9 + 4], slow_e[:, :9 + 4])                # (incl. the step where the free run picked `eos`)
            model.generation_config.eos_token_id = None
            hf_fast.revert_model_to_hf(model)


def test_forward_with_start_pos_is_the_fast_step():
    """model(ids, start_pos=p, use_cache=False) -- the harness' GeMM / GeMV loops (speed.py:65, 82): logits of every input row, fp32, against HF's
    own forward over the same modules; one-token calls take the captured step; a prompt fed in two chunks gives the one-chunk logits"""
    from amq_amd import hf_fast
    model, ref = _prepared("llama")
    ids = torch.randint(1, 1000, (1, 24), generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.inference_mode():
        want = model(ids).logits.float()                                          # HF's forward over the fused modules
        want_ref = ref(ids).logits.float()                                        # HF modules on the oracle's weights
    hf_fast.convert_model_to_hip(model)
    with torch.inference_mode():
        out = model(ids, start_pos=0, use_cache=False)
    lg = out.logits
    assert lg.shape == (1, 24, 1000) and lg.dtype == torch.float32 and out.start_pos == 24 and out.past_key_values is None
    scale = want_ref.abs().max()
    assert (lg - want).abs().max() <= 6e-3 * scale and (lg - want_ref).abs().max() <= 6e-3 * scale
    assert torch.equal(lg.argmax(-1), want.argmax(-1)) or (lg.argmax(-1) != want.argmax(-1)).sum() <= 1
    # the decode loop of speed.py:76-90, against HF's past_key_values loop over the same modules
    hf_fast.revert_model_to_hf(model)
    with torch.inference_mode():
        o = model(ids, use_cache=True)
        past, tok = o.past_key_values, o.logits[:, -1].max(1)[1].unsqueeze(1)
        slow_first = int(tok)
        slow_toks, slow_lg = [], []
        for _ in range(6):
            o = model(tok, past_key_values=past, use_cache=True)
            past = o.past_key_values
            slow_lg.append(o.logits[:, -1].float())
            tok = o.logits[:, -1].max(1)[1].unsqueeze(1)
            slow_toks.append(int(tok))
    hf_fast.convert_model_to_hip(model)
    with torch.inference_mode():
        o = model(ids, start_pos=0, use_cache=False)
        start = o.logits.shape[1]
        assert int(o.logits[:, -1].max(1)[1]) == slow_first
        fast_toks = []
        for i in range(6):
            # (fed HF's own greedy token, so that both sides see the same prefix even where a near-tie flips a choice)
            tok = torch.as_tensor([[slow_first if i == 0 else slow_toks[i - 1]]], device=DEV)
            o = model(tok, start_pos=start, use_cache=False)
            assert o.logits.shape == (1, 1, 1000)
            assert (o.logits[:, -1] - slow_lg[i]).abs().max() <= 6e-3 * scale, i
            start += o.logits.shape[1]
            fast_toks.append(int(o.logits[:, -1].max(1)[1]))
    assert sum(a_ == b_ for a_, b_ in zip(fast_toks, slow_toks)) >= 5, (fast_toks, slow_toks)
    r = hf_fast._RUNNERS[model][1]
    assert r.graph is not None and r.host_pos == 30                                # the one-token calls replayed the captured step
    # chunked prompt: rows 0..15 then 16..23 behind them
    with torch.inference_mode():
        a = model(ids[:, :16], start_pos=0, use_cache=False).logits
        b = model(ids[:, 16:], start_pos=16, use_cache=False).logits
    assert (torch.cat([a, b], 1) - lg).abs().max() <= 6e-3 * scale
    # two sequences at once
    ids2 = torch.randint(1, 1000, (2, 10), generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.inference_mode():
        l2 = model(ids2, start_pos=0, use_cache=False).logits
        w2 = ref(ids2).logits.float()
    assert l2.shape == (2, 10, 1000) and (l2 - w2).abs().max() <= 6e-3 * w2.abs().max()


def test_what_the_fast_path_does_not_serve_falls_through_or_says_so():
    from amq_amd import hf_fast
    model, _ = _prepared("llama")
    ids = torch.randint(1, 1000, (1, 7), generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.inference_mode():
        before = model(ids).logits.clone()
        keys = list(model.state_dict().keys())
    hf_fast.convert_model_to_hip(model)
    with torch.inference_mode():
        assert torch.equal(model(ids).logits, before)                             # no start_pos: HF's own forward, the same bits
        assert torch.equal(model(input_ids=ids, use_cache=True).logits, before)
        torch.manual_seed(0)
        s1 = model.generate(ids, max_new_tokens=5, do_sample=True, top_k=5, attention_mask=torch.ones_like(ids), pad_token_id=0)   # sampling: HF
        assert s1.shape[1] <= 12 and model not in hf_fast._RUNNERS
        g = model.generate(ids, max_new_tokens=5, do_sample=False, attention_mask=torch.ones_like(ids), pad_token_id=0)           # no min_new_tokens: HF (EOS may stop it)
        assert model not in hf_fast._RUNNERS and g.shape[1] <= 12
        mask = torch.ones_like(ids)
        mask[0, 0] = 0
        model.generate(ids, min_new_tokens=3, max_new_tokens=3, do_sample=False, num_beams=1, attention_mask=mask, pad_token_id=0)   # padded prompt: HF
        assert model not in hf_fast._RUNNERS
        with pytest.raises(ValueError, match="start_pos"):
            model(ids, start_pos=0, labels=ids)
        with pytest.raises(ValueError, match="start_pos"):
            model(ids, start_pos=0, use_cache=True)
    assert list(model.state_dict().keys()) == keys                                # buffers / parameters untouched
    # deepcopy: the copy is converted too, with its own runner over its own buffers
    cp = copy.deepcopy(model)
    with torch.inference_mode():
        a = _hf_generate(model, ids, 6)
        b = _hf_generate(cp, ids, 6)
    assert torch.equal(a, b) and hf_fast._RUNNERS[cp][1] is not hf_fast._RUNNERS[model][1]
    assert hf_fast._RUNNERS[cp][1].blocks[0]["self_attn.q_proj"].qn.data_ptr() == cp.model.layers[0].self_attn.q_proj.qweight.data_ptr()
    # a sequence that outgrows the runner's cache: rebuilt larger with the cache carried over
    with torch.inference_mode():
        o = model(ids, start_pos=0, use_cache=False)
        r0 = hf_fast._RUNNERS[model][1]
        tok = torch.as_tensor([[int(o.logits[:, -1].max(1)[1])]], device=DEV)
        o2 = model(tok, start_pos=7, use_cache=False)
        assert hf_fast._RUNNERS[model][1] is r0
    # a decoder the runner's block does not describe (per-head q / k norms: Qwen3, Gemma) is refused at convert time, with the reason
    hf_fast.revert_model_to_hf(model)
    model.model.layers[0].self_attn.q_norm = torch.nn.Identity()
    with pytest.raises(ValueError, match="q / k norms"):
        hf_fast.convert_model_to_hip(model)
    assert "forward" not in model.__dict__


@pytest.mark.parametrize("family", ["llama3.1", "mistral", "qwen2"])
def test_model_families_through_the_swap_and_the_runner(family):
    """Llama-3.1 (llama3 rope scaling: positions beyond original_max_position_embeddings / factor matter), Mistral (GQA 4), Qwen2 (q / k / v bias,
    GQA 7, eps 1e-6, 7 x 128 hidden): prepare_for_inference + HF's forward against HF modules on the oracle's weights; the runner (converted
    object) against both; greedy tokens equal HF's eager generate"""
    from amq_amd import hf_fast
    from amq_amd.quant_linear import HIPQuantLinear
    model, ref = _prepared(family)
    assert sum(isinstance(m, HIPQuantLinear) for m in model.modules()) == 14
    if family == "qwen2":
        assert all(getattr(l.self_attn, n).bias is not None for l in model.model.layers for n in ("q_proj", "k_proj", "v_proj"))
    S = 150 if family == "llama3.1" else 40                                        # (llama3.1: past the scaled wavelengths of the tiny config)
    ids = torch.randint(1, 1000, (1, S), generator=torch.Generator().manual_seed(7)).to(DEV)
    with torch.inference_mode():
        y = model(ids).logits.float()
        y_ref = ref(ids).logits.float()
        slow = _hf_generate(model, ids[:, :20], 10)
    scale = y_ref.abs().max()
    assert torch.isfinite(y).all() and (y - y_ref).abs().max() <= 3e-3 * scale
    hf_fast.convert_model_to_hip(model)
    with torch.inference_mode():
        lg = model(ids, start_pos=0, use_cache=False).logits
        fast = _hf_generate(model, ids[:, :20], 10)
    assert (lg - y_ref).abs().max() <= 6e-3 * scale
    _assert_same_tokens_or_a_tie(model, fast, slow, 20)
    assert torch.equal(fast[:, :20 + 3], slow[:, :20 + 3])
    r = hf_fast._RUNNERS[model][1]
    if family == "llama3.1":
        assert r.inv_freq is not None and not torch.allclose(r.inv_freq.cpu(), 1.0 / (500000.0 ** (torch.arange(0, 128, 2).float() / 128)))
    if family == "qwen2":
        assert r.has_bias and (r.nh, r.nkv, r.H) == (7, 1, 896)


@pytest.mark.parametrize("family", ["mistral", "qwen2", "llama"])
def test_grouped_query_families_decode_over_a_long_cache(family):
    """a cache of 1024 rows puts the decode attention of the grouped-query families (4 / 7 / 2 query heads per kv head) on the grouped kernel
    (attn_decode_gqa_kernel: one workgroup per kv head and chunk, matrix cores): a 600-token prompt, then decode steps fed HF's own greedy tokens --
    every step's logits against HF's past_key_values loop over the same modules"""
    from amq_amd import ops
    from amq_amd.llama import QuantLlama
    model, _ = _prepared(family)
    model.config.max_position_embeddings = 2048                    # (the tiny configs say 256; the default rotary embedding does not depend on it)
    ids = torch.randint(1, 1000, (1, 600), generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.inference_mode():
        o = model(ids, use_cache=True)
        past, tok = o.past_key_values, o.logits[:, -1].max(1)[1].unsqueeze(1)
        first_lg = o.logits[0, -1].float()
        toks, slow_lg = [int(tok)], []
        for _ in range(5):
            o = model(tok, past_key_values=past, use_cache=True)
            past = o.past_key_values
            slow_lg.append(o.logits[0, -1].float())
            tok = o.logits[:, -1].max(1)[1].unsqueeze(1)
            toks.append(int(tok))
    r = QuantLlama.from_hf(model, max_seq=1024)
    assert r.nh // r.nkv in (2, 4, 7) and ops.attn_decode_splits(1024, r.nh, 1, r.nkv) > 1
    scale = first_lg.abs().max()
    with torch.inference_mode():
        r.prefill(ids[0])
        assert (r.logits.view(-1).float() - first_lg).abs().max() <= 6e-3 * scale
        for i in range(5):
            r.set_token(torch.as_tensor([toks[i]], device=DEV))
            r.decode_step()
            assert (r.logits.view(-1).float() - slow_lg[i]).abs().max() <= 6e-3 * scale, i
    assert r.host_pos == 605
