"""convert_model_to_hip -- the reference's ``use_ft`` call surface on the prepared HF object itself.

With ``--use_ft`` the reference patches the model it hands to its harness: ``convert_model_to_ft(model)`` replaces the Llama forwards so that
``model(input_ids, start_pos=..., use_cache=False)`` runs the static-cache fast step (kernel/monkeypatch/ftllama_modeling.py:427-492, 569-580) and
``replace_generate_functions()`` patches ``GenerationMixin._sample`` so that the caller's own ``model.generate(...)`` drives it
(ftllama_generate.py:613-622); the harness then calls exactly those two things (amq/utils/speed.py:31-36, 65, 82).

Here the fast step is the hipGraph runner (llama.QuantLlama).  ``convert_model_to_hip(model)`` binds one lazily over the swapped model's own
buffers (QuantLlama.from_hf: no weight copies) and gives THIS model instance

  * ``model(input_ids, start_pos=p, use_cache=False)`` -- batch 1..8, any prompt length that fits the cache: the runner's prompt pass (captured per
    (length, start_pos)) or, for one new token at the runner's current position, the captured token step.  Returns a ``CausalLMOutputWithPast``
    whose ``logits`` are fp32 ``[B, S, vocab]`` for every input row, as HF's forward does (``.start_pos`` = the next position, as the
    reference's output carries);
  * ``model.generate(ids, min_new_tokens=n, max_new_tokens=n, do_sample=False, num_beams=1, attention_mask=all ones)`` -- greedy, fixed length,
    batch 1..8: prefill + n - 1 graph replays without a host sync in between; returns ``[B, S + n]`` ids like HF.

Anything else -- sampling, beams, an attention mask with holes, ``past_key_values``, ``labels``, ``inputs_embeds``, hidden-state / attention outputs,
stopping criteria, streamers, more than 8 sequences, a call without ``start_pos`` -- falls through to the model's original ``forward`` / ``generate``
(HF's own, over the fused modules).  ``state_dict`` / ``deepcopy`` / ``.to()`` are untouched: the runners live outside the module, keyed weakly by it.
"""
import types
import weakref

import torch

try:
    from transformers.modeling_outputs import CausalLMOutputWithPast
except Exception:                                   # (transformers is needed only once a model is converted)
    CausalLMOutputWithPast = None

_RUNNERS = weakref.WeakKeyDictionary()      # model -> {batch: QuantLlama}
_BUCKETS = (256, 512, 1024, 2048, 4096, 8192, 16384, 32768)
# generation_config fields that make HF add a logits processor / warper / constraint to a greedy run: any of them set -> HF's own generate
_GC_PROCESSORS = ("repetition_penalty", "encoder_repetition_penalty", "no_repeat_ngram_size", "encoder_no_repeat_ngram_size", "bad_words_ids",
                  "force_words_ids", "constraints", "forced_bos_token_id", "forced_eos_token_id", "exponential_decay_length_penalty", "suppress_tokens",
                  "begin_suppress_tokens", "sequence_bias", "guidance_scale", "watermarking_config", "renormalize_logits", "remove_invalid_values",
                  "penalty_alpha", "dola_layers", "prompt_lookup_num_tokens", "num_beam_groups", "diversity_penalty")
MAX_BATCH = 8


def _bucket(n, limit):
    for b in _BUCKETS:
        if n <= b:
            return min(b, max(limit, n))
    return n


def _runner(model, batch, need):
    """the runner for ``batch`` sequences with room for ``need`` positions (built on first use; rebuilt larger -- cache contents carried over -- when a
    sequence outgrows it: the attention launch is chosen by the cache's size, so the cache is not made larger than asked for)"""
    from .llama import QuantLlama
    per = _RUNNERS.setdefault(model, {})
    r = per.get(batch)
    limit = int(getattr(model.config, "max_position_embeddings", 1 << 30) or (1 << 30))
    if need > limit:
        raise ValueError(f"{need} positions exceed the model's max_position_embeddings ({limit})")
    if r is not None and r.max_seq >= need and _same_weights(r, model):
        return r
    new = QuantLlama.from_hf(model, max_seq=_bucket(need, limit), batch=batch)
    new.all_logits = True
    if r is not None and _same_weights(r, model) and r.host_pos > 0:
        for nb, ob in zip(new.blocks, r.blocks):
            nb["kc"][:, :, :r.host_pos].copy_(ob["kc"][:, :, :r.host_pos])
            nb["vc"][:, :, :r.host_pos].copy_(ob["vc"][:, :, :r.host_pos])
        new.set_pos(r.host_pos)
        new.set_token(r.token)
    per[batch] = new
    return new


def _same_weights(r, model):
    """the runner still reads the buffers the modules own (a linear replaced or moved since -- the reference's driver setattr's linears between
    models, amq_speed_benchmark.py:231-251 -- means a new runner)"""
    layer = model.model.layers[0]
    q = layer.self_attn.q_proj
    return r.blocks[0]["self_attn.q_proj"].qn.data_ptr() == q.qweight.data_ptr() and r.nb == len(model.model.layers)


def _plain_ids(input_ids):
    return (isinstance(input_ids, torch.Tensor) and input_ids.dim() == 2 and input_ids.dtype in (torch.int64, torch.int32)
            and 1 <= input_ids.shape[0] <= MAX_BATCH and input_ids.shape[1] >= 1)


def _mask_is_full(mask, ids):
    if mask is None:
        return True
    return isinstance(mask, torch.Tensor) and mask.shape == ids.shape and bool(mask.ne(0).all())


def _fast_forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, start_pos=None, inputs_embeds=None,
                  labels=None, use_cache=None, **kwargs):
    """``LlamaForCausalLM.forward`` with the reference's extra ``start_pos`` argument (ftllama_modeling.py:428-441)"""
    orig = self.__dict__["_amq_orig_forward"]
    extras = {k: v for k, v in kwargs.items() if v is not None and v is not False and not (k == "return_dict" and v is True)
              and not (k == "logits_to_keep" and v == 0)}
    if (start_pos is None or not _plain_ids(input_ids) or past_key_values is not None or inputs_embeds is not None or labels is not None
            or position_ids is not None or use_cache or extras or not input_ids.is_cuda or not _mask_is_full(attention_mask, input_ids)):
        if start_pos is not None:
            raise ValueError("model(..., start_pos=) serves input_ids [1..8, S] on the GPU with use_cache=False and nothing else "
                             "(no mask with holes, past_key_values, labels, inputs_embeds or extra outputs); drop start_pos for HF's own forward")
        return orig(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids, past_key_values=past_key_values,
                    inputs_embeds=inputs_embeds, labels=labels, use_cache=use_cache, **kwargs)
    B, S = input_ids.shape
    start_pos = int(start_pos)
    # room for the tokens that usually follow (a rebuilt runner re-captures its step): twice the context, at most 1024 more, never past the model's limit
    limit = int(getattr(self.config, "max_position_embeddings", 1 << 30) or (1 << 30))
    end = start_pos + S
    r = _runner(self, B, max(end, min(limit, max(end + 1, 2 * end if end <= 1024 else end + 1024))))
    ids = input_ids if B > 1 else input_ids[0]
    if S == 1 and start_pos == r.host_pos and start_pos > 0:
        r.set_token(input_ids.reshape(-1))
        r.decode_step()
        logits = r.logits.view(B, 1, r.vocab).float()
    else:
        r.prefill(ids, start_pos=start_pos)
        logits = r.logits_rows.float()
    out = CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=None, hidden_states=None, attentions=None)
    out["start_pos"] = start_pos + S          # (the reference's output carries it: ftllama_modeling.py:486)
    return out


def _fast_generate(self, inputs=None, generation_config=None, logits_processor=None, stopping_criteria=None, prefix_allowed_tokens_fn=None,
                   synced_gpus=None, assistant_model=None, streamer=None, negative_prompt_ids=None, negative_prompt_attention_mask=None,
                   custom_generate=None, **kwargs):
    orig = self.__dict__["_amq_orig_generate"]

    def fall():
        return orig(inputs, generation_config=generation_config, logits_processor=logits_processor, stopping_criteria=stopping_criteria,
                    prefix_allowed_tokens_fn=prefix_allowed_tokens_fn, synced_gpus=synced_gpus, assistant_model=assistant_model, streamer=streamer,
                    negative_prompt_ids=negative_prompt_ids, negative_prompt_attention_mask=negative_prompt_attention_mask,
                    custom_generate=custom_generate, **kwargs)

    kw = dict(kwargs)
    ids = inputs if inputs is not None else kw.pop("input_ids", None)
    if inputs is not None and "input_ids" in kw:
        return fall()
    gc = getattr(self, "generation_config", None)
    n = kw.pop("max_new_tokens", None)
    nmin = kw.pop("min_new_tokens", None)
    greedy = kw.pop("do_sample", getattr(gc, "do_sample", False)) in (False, None) and kw.pop("num_beams", getattr(gc, "num_beams", 1)) in (1, None)
    mask = kw.pop("attention_mask", None)
    eos = kw.pop("eos_token_id", getattr(gc, "eos_token_id", None))
    kw.pop("pad_token_id", None)                             # (fixed-length greedy decoding never pads)
    for k in ("return_dict_in_generate", "output_scores", "output_logits", "output_attentions", "output_hidden_states", "use_cache"):
        if kw.get(k) in (None, False) or (k == "use_cache" and kw.get(k) is True):
            kw.pop(k, None)
    others = [generation_config, logits_processor, stopping_criteria, prefix_allowed_tokens_fn, assistant_model, streamer, negative_prompt_ids,
              negative_prompt_attention_mask, custom_generate]
    eos = [] if eos is None else ([int(e) for e in eos] if isinstance(eos, (list, tuple)) else [int(eos)])
    if (kw or any(o is not None and (not hasattr(o, "__len__") or len(o)) for o in others) or synced_gpus or not greedy or n is None or nmin != n
            or not _plain_ids(ids) or not ids.is_cuda or int(n) < 1 or len(eos) > 8 or not _mask_is_full(mask, ids)):
        return fall()
    # the model's own generation defaults must ask for plain greedy decoding too (any logits processor HF would add changes the tokens)
    if gc is not None and any(getattr(gc, k, None) not in (None, False, 0, 1, 1.0, [], ()) for k in _GC_PROCESSORS):
        return fall()
    B, S = ids.shape
    n = int(n)
    r = _runner(self, B, S + n)
    # min_new_tokens = max_new_tokens: HF never lets an EOS id through (MinNewTokensLengthLogitsProcessor sets their logits to -inf on every step)
    if tuple(eos) != getattr(r, "_suppressed", ()):
        r.set_suppressed(eos)
    new = r.generate(ids if B > 1 else ids[0], n)
    return torch.cat([ids, new.view(B, n).to(ids.dtype)], dim=1)


def convert_model_to_hip(model):
    """convert_model_to_ft(model) + replace_generate_functions() (ftllama_modeling.py:569-580, ftllama_generate.py:613-622) for the HIP backend:
    call it on the model ``prepare_for_inference(model, backend='hip')`` returned (a Llama-family ``*ForCausalLM`` whose decoder linears are
    HIPQuantLinear modules on one GPU).  Patches THIS instance's ``forward`` and ``generate`` (see the module docstring); idempotent; returns the
    model.  ``revert_model_to_hf(model)`` undoes it."""
    if not (hasattr(model, "lm_head") and hasattr(getattr(model, "model", None), "layers")):
        raise TypeError("convert_model_to_hip expects a Llama-family causal LM (model.model.layers, model.lm_head)")
    if "_amq_orig_forward" in model.__dict__:
        return model
    from .llama import QuantLlama
    QuantLlama.check_hf(model)                               # refuse now, with the reason, what the runner cannot serve
    model.__dict__["_amq_orig_forward"] = model.forward
    model.__dict__["_amq_orig_generate"] = model.generate
    model.forward = types.MethodType(_fast_forward, model)
    model.generate = types.MethodType(_fast_generate, model)
    return model


def revert_model_to_hf(model):
    for name in ("forward", "generate"):
        if "_amq_orig_" + name in model.__dict__:
            model.__dict__.pop(name, None)
            model.__dict__.pop("_amq_orig_" + name)
    _RUNNERS.pop(model, None)
    return model


def replace_generate_functions():
    """The reference patches ``GenerationMixin`` globally (ftllama_generate.py:613-622) and its driver calls this right after
    ``convert_model_to_ft`` (amq_speed_benchmark.py:79-80).  Here ``convert_model_to_hip`` patches the one instance, so this is a no-op kept for
    scripts that make both calls."""
    return None
