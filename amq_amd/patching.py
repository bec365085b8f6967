"""prepare_for_inference / patch_linearlayers -- the reference's plug-in API
(hqq/utils/patching.py:39-49, 143-223) for the HIP backend.

    prepare_for_inference(model, backend="hip", load_path=None)

walks ``named_children`` and replaces every HQQLinear by a HIPQuantLinear, with
the reference's cache-file behaviour (patching.py:178-208): if ``load_path`` is
given and missing, the patched ``model.state_dict()`` is saved there; if it
exists, empty modules are created and the state dict is loaded instead of
re-packing.  ``backend="gptq"`` and ``backend="ft"`` are accepted as aliases so
that ``amq_speed_benchmark.py:137-139`` runs unchanged: both produce
HIPQuantLinear modules (one class serves 2/3/4 bit).

Reference caches written by the CUDA build (``*_GPTQLinear.pt`` / ``*_FTLinear.pt``,
keys ``<module>.qweight|scales|zeros`` and ``<module>.qweight|scales|scaled_zeros``)
are imported with ``load_reference_cache``.
"""
import os

import torch

from . import ops
from .hqq_format import HQQWeights, from_hqq_layer
from .quant_linear import HIPLlamaMLP, HIPQuantLinear, HIPRMSNorm, LinearGroup

# sibling linears of one parent module that read the same input (HF LlamaAttention / LlamaMLP attribute names)
SIBLING_GROUPS = (("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj"))

HIP_BACKENDS = ("hip", "gptq", "ft")


def is_hqq_layer(layer):
    """HQQLinear / HQQLinearLoRA of the reference (quantize.py:387, peft.py) by
    duck typing -- this package never imports the reference."""
    if isinstance(layer, HQQWeightsModule):
        return True
    cls = type(layer).__name__
    if cls == "HQQLinear":
        return hasattr(layer, "W_q") and hasattr(layer, "meta")
    if cls == "HQQLinearLoRA":
        return hasattr(layer, "linear_layer")
    return False


class HQQWeightsModule(torch.nn.Module):
    """Minimal stand-in for HQQLinear holding an HQQWeights (used by the
    synthetic-model builders and tests; real HQQLinear objects work as well)."""

    def __init__(self, weights: HQQWeights):
        super().__init__()
        self.weights = weights
        self.W_q = weights.W_q
        self.meta = weights.meta
        self.bias = weights.bias
        self.name = weights.name
        self.device = weights.W_q.device


def patch_linearlayers(model, fct, patch_params=None, verbose=False):
    """The reference's walk (patching.py:39-49): depth first over ``named_children``; every HQQ layer gets its attribute name as
    ``layer.name`` and is replaced by ``fct(layer, patch_param)``; anything else is descended into."""
    stack = [model]
    while stack:
        parent = stack.pop()
        for attr, child in list(parent.named_children()):
            if not is_hqq_layer(child):
                stack.append(child)
                continue
            child.name = attr
            setattr(parent, attr, fct(child, patch_params))


def _inner(layer):
    return layer.linear_layer if type(layer).__name__ == "HQQLinearLoRA" else layer


def patch_hqq_to_hip(layer, patch_params=None, load=False):
    """Counterpart of patch_hqq_to_gptq / patch_hqq_to_ft (autogptq.py:291-341,
    ft.py:148-197): HQQLinear -> HIPQuantLinear on the layer's device."""
    if not is_hqq_layer(layer):
        return layer
    hqq_layer = _inner(layer)
    h = hqq_layer.weights if isinstance(hqq_layer, HQQWeightsModule) else from_hqq_layer(hqq_layer)
    h.name = getattr(hqq_layer, "name", None) or getattr(layer, "name", None)
    n, k = h.shape
    device = (patch_params or {}).get("device", None) or h.W_q.device
    # bfloat16 layers (compute_dtype = bfloat16) become fp16 modules like every module the reference's patchers build (ft.py:62) -- grouped, fused,
    # served by the decode runner -- unless the caller asks to keep HQQ's bf16 arithmetic (keep_bf16) and the bf16 kernels serve the layer's groups
    bf16 = bool((patch_params or {}).get("keep_bf16", False)) and h.scale.dtype == torch.bfloat16 and HIPQuantLinear.bf16_serves(h.group_size)
    if load:
        new = HIPQuantLinear(h.nbits, h.group_size, k, n, bias=h.bias, name=h.name,
                             weight_dtype=torch.bfloat16 if bf16 else torch.float16).to(device)
    else:
        new = HIPQuantLinear.from_hqq(h, device=device, keep_bf16=bf16)
    if type(layer).__name__ == "HQQLinearLoRA":
        layer.linear_layer = new
        return layer
    return new


def patch_hqq_to_hip_load(layer, patch_params=None):
    return patch_hqq_to_hip(layer, patch_params, load=True)


def patch_add_weight_param(layer, patch_param):
    """patching.py:76-91: dummy ``.weight`` so HF code can query dtype/device."""
    if isinstance(layer, HIPQuantLinear) and not hasattr(layer, "weight"):
        layer.weight = torch.nn.Parameter(torch.zeros((1,), device=layer.qweight.device, dtype=torch.float16),
                                          requires_grad=False)
    return layer


def _walk_hip(model, fct):
    for name, layer in model.named_children():
        if isinstance(layer, HIPQuantLinear):
            setattr(model, name, fct(layer, None))
        else:
            _walk_hip(layer, fct)


def _fusable(m):
    """a HIPQuantLinear the grouped / fused launches take"""
    return isinstance(m, HIPQuantLinear) and not m.is_bf16        # (bfloat16 modules run unfused: ops.linear_bf16)


def _same_group(mods):
    """siblings of ONE launch share the granularity of their native meta (128, or 64 / 32: amq_gemv_grouped_f16 takes one `group`)"""
    return len({getattr(m, "native_group", ops.GROUP) for m in mods}) == 1


def group_sibling_linears(model):
    """Install a :class:`LinearGroup` wherever a module holds HIPQuantLinear children named like one of SIBLING_GROUPS with equal
    input sizes: their few-row forwards then run as ONE grouped launch.  Returns the number of groups made.  The modules
    themselves (buffers, state_dict keys) are not changed.  Siblings that already form exactly this group are left alone; siblings
    that carry groups from ELSEWHERE -- the reference's driver assembles a mixed-precision model by ``setattr``-ing linears taken
    from three separately prepared models (amq_speed_benchmark.py:231-251) -- are regrouped with their new neighbours."""
    made = 0
    for parent in model.modules():
        for names in SIBLING_GROUPS:
            mods = [getattr(parent, n, None) for n in names]
            if not all(_fusable(m) for m in mods) or len({m.infeatures for m in mods}) != 1 or not _same_group(mods):
                continue
            grps = [m.__dict__.get("_group") for m in mods]
            if all(g is not None and g[0] is grps[0][0] and g[1] == i for i, g in enumerate(grps)) and len(grps[0][0].members) == len(mods) \
                    and all(a is b for a, b in zip(grps[0][0].members, mods)):
                continue
            LinearGroup(mods)
            made += 1
    return made


def fuse_llama_mlps(model):
    """Replace every SiLU-gated MLP whose gate / up / down projections are (bias-free) HIPQuantLinear modules by
    :class:`HIPLlamaMLP` (two launches per few-row forward).  Recognised by structure: children ``gate_proj``, ``up_proj``,
    ``down_proj`` and an ``act_fn`` that is SiLU.  Returns the number of modules replaced."""
    n = 0
    for parent in list(model.modules()):
        for name, mod in list(parent.named_children()):
            if isinstance(mod, HIPLlamaMLP):
                continue
            kids = [getattr(mod, k, None) for k in ("gate_proj", "up_proj", "down_proj")]
            act = getattr(mod, "act_fn", None)
            is_silu = isinstance(act, torch.nn.SiLU) or type(act).__name__ in ("SiLUActivation", "SiLU")
            if all(_fusable(k) and k.bias is None for k in kids) and is_silu and _same_group(kids[:2]):      # (gate / up share a launch)
                setattr(parent, name, HIPLlamaMLP(*kids))
                n += 1
    return n


# RMSNorm classes whose forward is weight * x / sqrt(mean(x^2) + eps) with fp32 statistics (HF LlamaRMSNorm; module_walk's own)
# (transformers' Mistral / Qwen2 classes are generated from the Llama ones: the same forward, token for token)
RMSNORM_CLASSES = ("LlamaRMSNorm", "MistralRMSNorm", "Qwen2RMSNorm", "_RMSNorm")
# decoder layers whose ``forward`` is LlamaDecoderLayer's (checked by signature as well: fuse_llama_layers)
HF_LAYER_CLASSES = ("LlamaDecoderLayer", "MistralDecoderLayer", "Qwen2DecoderLayer")


def fuse_llama_norms(model):
    """In every Llama decoder layer -- a module with children ``input_layernorm``, ``self_attn``, ``post_attention_layernorm``,
    ``mlp`` -- whose q/k/v projections are one LinearGroup and whose MLP is a HIPLlamaMLP, wrap the two RMSNorms in
    :class:`HIPRMSNorm`: few-row forwards then form the norm in the prologue of the grouped launch that follows it (9 -> 7
    launches per block for a module swap).  Only the norm classes of RMSNORM_CLASSES are touched.  Returns the number wrapped."""
    n = 0
    for layer in list(model.modules()):
        attn, mlp = getattr(layer, "self_attn", None), getattr(layer, "mlp", None)
        if attn is None or mlp is None:
            continue
        qkv = [getattr(attn, k, None) for k in SIBLING_GROUPS[0]]
        grp = qkv[0].__dict__.get("_group") if all(_fusable(m) and m.bias is None for m in qkv) else None
        if grp is not None and not (len(grp[0].members) == 3 and all(a is b for a, b in zip(grp[0].members, qkv))):
            grp = None
        for name, consumer in (("input_layernorm", grp[0] if grp is not None else None),
                               ("post_attention_layernorm", mlp if isinstance(mlp, HIPLlamaMLP) else None)):
            norm = getattr(layer, name, None)
            slot = "qkv" if name == "input_layernorm" else "mlp"
            if isinstance(norm, HIPRMSNorm):
                # a second prepare_for_inference (or a sibling transplanted since): the wrapper exists but may point at a group that no
                # longer owns q/k/v -- re-target it, or un-fuse it when its launch is gone
                if consumer is None:
                    setattr(layer, name, norm.__dict__["_inner"])
                elif norm.__dict__.get("_consumer") is not consumer or norm.__dict__.get("_owner") is not layer:
                    norm.retarget(consumer, layer, slot)
                    n += 1
                continue
            if consumer is None or norm is None or type(norm).__name__ not in RMSNORM_CLASSES or getattr(norm, "weight", None) is None:
                continue
            setattr(layer, name, HIPRMSNorm(norm, consumer, layer, slot))
            n += 1
    return n


def _fused_attention(layer, h, residual, call):
    """``residual + self_attn(h)`` with the add formed in o_proj's epilogue when that call can take it (HIPQuantLinear._forward_residual):
    the residual is offered to o_proj through a one-shot holder for the duration of the attention call only."""
    o_proj = getattr(layer.self_attn, "o_proj", None)
    if not _fusable(o_proj) or not residual.is_cuda or residual.dtype is not torch.float16:
        return residual + call(h)
    hold = [residual]
    o_proj.__dict__["_residual"] = hold
    try:
        out = call(h)
    finally:
        o_proj.__dict__.pop("_residual", None)
    return out if hold[0] is None else residual + out


def _hf_llama_layer_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                            position_embeddings=None, **kwargs):
    """transformers 5.x ``LlamaDecoderLayer.forward`` (modeling_llama.py) with the two ``residual + hidden_states`` adds folded into
    the o_proj / down_proj launches: same modules, same order, same values (the epilogue adds the residual to the fp16-rounded
    projection, i.e. the two roundings of the separate add)."""
    attn = lambda h: self.self_attn(hidden_states=h, attention_mask=attention_mask, position_ids=position_ids,
                                    past_key_values=past_key_values, use_cache=use_cache, position_embeddings=position_embeddings,
                                    **kwargs)[0]
    hidden_states = _fused_attention(self, self.input_layernorm(hidden_states), hidden_states, attn)
    h = self.post_attention_layernorm(hidden_states)
    if isinstance(self.mlp, HIPLlamaMLP):
        return self.mlp(h, residual=hidden_states)
    return hidden_states + self.mlp(h)


def _walk_block_forward(self, x):
    """module_walk._Block.forward (the HF layer's shape without HF's argument plumbing), fused the same way"""
    x = _fused_attention(self, self.input_layernorm(x), x, self.self_attn)
    h = self.post_attention_layernorm(x)
    if isinstance(self.mlp, HIPLlamaMLP):
        return self.mlp(h, residual=x)
    return x + self.mlp(h)


_HF_LAYER_PARAMS = ["self", "hidden_states", "attention_mask", "position_ids", "past_key_values", "use_cache", "position_embeddings", "kwargs"]


def fuse_llama_layers(model):
    """Fold the decoder layers' residual adds into the o_proj / down_proj launches: the layer instance's ``forward`` is replaced
    by a function that runs the SAME sub-modules in the same order and offers each residual to the projection that precedes its add
    (7 -> 5 launches per block for few rows; more rows, biases or other dtypes fall back to the plain add).  Only layers whose
    forward is known are touched: transformers' ``LlamaDecoderLayer`` with exactly the 5.x signature (its body is what
    ``_hf_llama_layer_forward`` restates) and module_walk's ``_Block``; the reference's FT path replaces Llama forwards the same
    way (kernel/monkeypatch/ftllama_modeling.py:39-46, 127-155).  Returns the number of layers patched."""
    import inspect
    import types
    n = 0
    for layer in model.modules():
        cls = type(layer)
        if "forward" in layer.__dict__ or not all(hasattr(layer, k) for k in ("input_layernorm", "self_attn", "post_attention_layernorm", "mlp")):
            continue
        if cls.__name__ == "_Block" and cls.__module__.endswith("module_walk"):
            layer.forward = types.MethodType(_walk_block_forward, layer)
        elif cls.__name__ in HF_LAYER_CLASSES and list(inspect.signature(cls.forward).parameters) == _HF_LAYER_PARAMS:
            layer.forward = types.MethodType(_hf_llama_layer_forward, layer)
        else:
            continue
        n += 1
    return n


def _merge_zeros_with_lora(model):
    """``allow_merge=True`` (patching.py:210-216, patch_merge_zeros_with_lora :239-275): the reference folds the zero-point term
    of an HQQLinearLoRA into its LoRA factors -- only for layers quantized WITHOUT groups along axis 1 (":243-245: axis == 0 or a
    group_size -> 'Skipping zeros lora merging'").  Every AMQ layer has group_size 128, so on AMQ's path the reference skips every
    layer; this does the same walk, prints the same line for what the reference skips and refuses the one case it would merge
    (group-less symmetric kernels are not part of this package)."""
    for name, layer in model.named_modules():
        if type(layer).__name__ != "HQQLinearLoRA":
            continue
        meta = getattr(getattr(layer, "linear_layer", None), "meta", None) or {}
        if meta.get("axis") == 0 or meta.get("group_size") is not None:
            print("Skipping zeros lora merging for", getattr(layer, "name", name))
        else:
            raise NotImplementedError(f"allow_merge: {name} is quantized without groups; zero/LoRA merging for group-less layers "
                                      "is outside the AMQ path (group_size 128)")


def prepare_for_inference(model, allow_merge=False, backend="hip", verbose=False, load_path=None, group_siblings=True, fuse_mlp=True,
                          fuse_norms=True, fuse_layers=True, kernel_arithmetic=False, keep_bf16=False):
    """patching.py:143-223 for the HIP backend.  ``kernel_arithmetic`` (default off): the swapped linears dequantize like the reference's GPTQ /
    FT CUDA kernels -- w = fma(q, s, -fp16(z s)), one rounding, what its own ``backend='gptq'`` / ``'ft'`` paths compute -- instead of like HQQ's
    ``dequantize()`` (two roundings, the default here: bit-identical to ``W_deq``).  The two differ by the rounding of c = fp16(z s): |dw| <= ~ulp(z s),
    i.e. relative to |z s|, not to |w| (several ulps of a weight with |q - z| << z) -- exactly the reference's ``backend='gptq'`` numerics; 5-8 % faster decode
    (HIPQuantLinear.to_kernel_arithmetic).  The cache file always holds the HQQ form.  ``group_siblings`` (default on; not in the reference): q/k/v and gate/up
    siblings are additionally tied into grouped launches (group_sibling_linears); ``fuse_mlp`` / ``fuse_norms``: SiLU-gated MLPs
    and the decoder layers' RMSNorms are fused into those launches (fuse_llama_mlps, fuse_llama_norms); ``fuse_layers``: the decoder
    layers' residual adds move into the o_proj / down_proj epilogues (fuse_llama_layers).  ``keep_bf16`` (default off): layers quantized with
    compute_dtype = bfloat16 keep HQQ's bf16 arithmetic (bfloat16 modules: ungrouped, unfused, groups of 128 and multiples) instead of
    becoming fp16 modules as under the reference's patchers."""
    if backend not in HIP_BACKENDS:
        raise RuntimeError(f"backend '{backend}' is not available in amq_amd (use one of {HIP_BACKENDS})")
    if allow_merge:
        _merge_zeros_with_lora(model)
    # the cache-file contract of patching.py:178-208: no path -> convert in place; a path that does not exist yet -> convert and
    # write the patched state_dict there; an existing file -> build empty modules and load it instead of re-packing
    cached = load_path is not None and os.path.exists(load_path)
    patch_linearlayers(model, patch_hqq_to_hip_load if cached else patch_hqq_to_hip, {"keep_bf16": keep_bf16}, verbose=verbose)
    if cached:
        print("Loading the model from", load_path)
        model.load_state_dict(torch.load(load_path, weights_only=True))
    elif load_path is not None:
        print("Saving the model to", load_path)
        torch.save(model.state_dict(), load_path)
    else:
        print("No load_path provided, using the model as is")
    _walk_hip(model, patch_add_weight_param)
    if kernel_arithmetic:
        for mod in model.modules():
            if isinstance(mod, HIPQuantLinear):
                mod.to_kernel_arithmetic()
    if group_siblings:
        group_sibling_linears(model)
    if fuse_mlp:                      # (not in the reference's generic patcher; its FT path swaps whole Llama sub-modules too)
        fuse_llama_mlps(model)
    if fuse_norms and group_siblings:
        fuse_llama_norms(model)
    if fuse_layers:                   # the decoder layers' residual adds into the o_proj / down_proj epilogues
        fuse_llama_layers(model)
    _warn_unfused(model, group_siblings, fuse_mlp, fuse_norms and group_siblings, fuse_layers)
    return model


def _warn_unfused(model, group_siblings, fuse_mlp, fuse_norms, fuse_layers):
    """One warning per fusion step that was asked for and found NOTHING to fuse on a model that does hold HIPQuantLinear projections
    with the Llama names: the steps recognise HF's module structure by shape (child names, SiLU, LlamaRMSNorm, the 5.x decoder-layer
    signature), so a transformers refactor would otherwise fall back to 13 launches per block without a word.  Correctness is not
    affected either way."""
    import warnings
    lins = [(n.rsplit(".", 1)[-1], m) for n, m in model.named_modules() if isinstance(m, HIPQuantLinear)]
    names = {n for n, _ in lins}
    if not lins:
        return
    sib = {n for g in SIBLING_GROUPS for n in g}
    if group_siblings and names & sib and not any("_group" in m.__dict__ for n, m in lins if n in sib):
        warnings.warn("prepare_for_inference: q/k/v / gate/up siblings found but none could be grouped (group_sibling_linears matched "
                      "nothing): every projection runs as its own launch", RuntimeWarning, stacklevel=3)
    if fuse_mlp and {"gate_proj", "up_proj", "down_proj"} <= names and not any(isinstance(m, HIPLlamaMLP) for m in model.modules()):
        warnings.warn("prepare_for_inference: gate/up/down projections found but no MLP was fused (fuse_llama_mlps matched nothing: "
                      "no SiLU-gated, bias-free MLP of three HIPQuantLinear children)", RuntimeWarning, stacklevel=3)
    layers = [m for m in model.modules() if all(hasattr(m, k) for k in ("input_layernorm", "self_attn", "post_attention_layernorm", "mlp"))
              and any(isinstance(c, HIPQuantLinear) for c in m.modules())]
    if fuse_norms and layers and not any(isinstance(m, HIPRMSNorm) for m in model.modules()):
        warnings.warn("prepare_for_inference: decoder layers found but no RMSNorm was fused into its consumer (fuse_llama_norms matched "
                      "nothing: needs a grouped q/k/v, a fused MLP and a LlamaRMSNorm-class norm)", RuntimeWarning, stacklevel=3)
    if fuse_layers and layers and not any("forward" in m.__dict__ for m in layers):
        warnings.warn("prepare_for_inference: decoder layers found but none took the fused forward (fuse_llama_layers matched nothing: "
                      "the layer class is not transformers' LlamaDecoderLayer with the known forward signature); residual adds stay "
                      "separate launches", RuntimeWarning, stacklevel=3)


def load_reference_cache(state_dict, device="cuda"):
    """Import a backend cache written by the reference's CUDA build
    (patching.py:182-189 / 198-205).  Returns {module_prefix: HIPQuantLinear}.

    GPTQLinear entries: ``<p>.qweight`` int32, ``<p>.scales``/``<p>.zeros`` fp32;
    FT_QuantLinear entries: ``<p>.qweight`` int16, ``<p>.scales``/``<p>.scaled_zeros`` fp16."""
    out = {}
    prefixes = sorted({k[: -len(".qweight")] for k in state_dict if k.endswith(".qweight")})
    for p in prefixes:
        qw = state_dict[p + ".qweight"].to(device)
        bias = state_dict.get(p + ".bias")
        bias = None if bias is None else bias.to(device)
        if qw.dtype == torch.int16:
            out[p] = HIPQuantLinear.from_ft_buffers(qw, state_dict[p + ".scales"].to(device),
                                                    state_dict[p + ".scaled_zeros"].to(device), bias=bias, name=p.split(".")[-1])
        elif qw.dtype == torch.int32:
            scales = state_dict[p + ".scales"].to(device)
            k_over_g, n = scales.shape
            bits = qw.shape[0] * 32 // (k_over_g * 128)
            out[p] = HIPQuantLinear.from_gptq_buffers(qw, scales, state_dict[p + ".zeros"].to(device), bits, bias=bias,
                                                      name=p.split(".")[-1])
        else:
            raise ValueError(f"{p}.qweight: unexpected dtype {qw.dtype}")
    return out
