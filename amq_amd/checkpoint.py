"""On-disk formats of the reference.

``{dir}/config.json`` + ``{dir}/qmodel.pt`` as written by ``AutoHQQHFModel.save_quantized``
(hqq/models/base.py:244-258, 405-435): ``qmodel.pt`` is ``{module_name: state_dict}`` where an HQQLinear's
state dict is the *un-encoded* form of HQQLinear.state_dict (quantize.py:643-680): ``W_q`` plus the meta
entries (``nbits, group_size, shape, scale, zero, axis, packing, view_as_float, ...``) as plain Python values,
optional ``bias``; every other module contributes its ordinary ``{'weight': ...}``.

AMQ keeps one such directory per bit-width, ``{save_path}/{model}_{n}bit_128gs_1axis``
(amq_quantization_proxy.py:40-42, amq_speed_benchmark.py:129-131); ``load_mixed`` picks every linear from the
directory its arch entry names, exactly what amq_speed_benchmark.py:231-251 does with module objects.
"""
import json
import os

import torch

from .hqq_format import GROUP, HQQWeights

LINEARS = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
           "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]


def load_hqq_dir(path):
    """-> (hf_config dict, {module_name: HQQWeights | {'weight': tensor, ...}})"""
    with open(os.path.join(path, "config.json")) as f:
        hf = json.load(f)
    # tensors + plain Python values + torch.Size / torch.dtype: all on torch's weights-only allow-list, so no arbitrary
    # pickle code runs (the reference loads the same file the same way, hqq/models/base.py:254-257)
    raw = torch.load(os.path.join(path, "qmodel.pt"), map_location="cpu", weights_only=True)
    out = {}
    for name, sd in raw.items():
        if "W_q" in sd:
            meta = sd["meta"] if "meta" in sd else sd          # old files nest the meta dict (quantize.py:751-752)
            if meta.get("axis", 1) != 1 or meta.get("view_as_float", False):
                raise ValueError(f"{name}: only axis=1, non-float-view HQQ layers are supported")
            if meta.get("quant_scale", False) or meta.get("quant_zero", False):
                raise ValueError(f"{name}: quantized scale/zero are not supported (deprecated in the reference as well)")
            if int(meta["group_size"]) not in (64, 32) and (int(meta["group_size"]) < GROUP or int(meta["group_size"]) % GROUP):
                raise ValueError(f"{name}: group size {meta['group_size']} (32, 64 or multiples of 128 only)")
            out[name] = HQQWeights(sd["W_q"], meta["scale"].to(torch.float16).reshape(-1, 1),
                                   meta["zero"].to(torch.float16).reshape(-1, 1), int(meta["nbits"]),
                                   tuple(int(v) for v in meta["shape"]), int(meta["group_size"]), sd.get("bias"), name.split(".")[-1])
        else:
            out[name] = sd
    return hf, out


def runner_config(hf):
    """HF LlamaConfig json -> the dict arch.MODEL_CONFIGS uses"""
    from .arch import _cfg
    hd = hf.get("head_dim") or hf["hidden_size"] // hf["num_attention_heads"]
    if hd != 128:
        raise ValueError("head_dim must be 128")
    kv = hf.get("num_key_value_heads") or hf["num_attention_heads"]
    numel = hf["num_hidden_layers"] * (2 * hf["hidden_size"] * hf["hidden_size"] + 2 * hf["hidden_size"] * kv * 128
                                       + 3 * hf["hidden_size"] * hf["intermediate_size"])
    c = dict(_cfg(hf["num_hidden_layers"], hf["hidden_size"], hf["intermediate_size"], hf["num_attention_heads"], kv,
                  numel, vocab=hf["vocab_size"]))
    c["rms_norm_eps"] = float(hf.get("rms_norm_eps", 1e-5))
    c["rope_theta"] = float(hf.get("rope_theta", 10000.0) or 10000.0)
    return c


def load_mixed(dirs, arch_linear, device="cuda:0", max_seq=256, batch=1):
    """dirs: {bits: checkpoint dir}; arch_linear: {'self_attn.q_proj': [bits]*n_block, ...}.
    Returns a QuantLlama holding real weights."""
    from .llama import QuantLlama
    loaded = {b: load_hqq_dir(d) for b, d in dirs.items()}
    hf, any_w = next(iter(loaded.values()))
    cfg = runner_config(hf)
    layers = {}
    for i in range(cfg["n_block"]):
        for name in LINEARS:
            b = int(arch_linear[name][i])
            w = loaded[b][1][f"model.layers.{i}.{name}"]
            if not isinstance(w, HQQWeights) or w.nbits != b:
                raise ValueError(f"model.layers.{i}.{name}: expected a {b}-bit HQQ layer in {dirs[b]}")
            layers[(i, name)] = w
    dense = {"embed": any_w["model.embed_tokens"]["weight"].to(torch.float16),
             "lm_head": any_w["lm_head"]["weight"].to(torch.float16),
             "norm": any_w["model.norm"]["weight"].to(torch.float16),
             "ln1": [any_w[f"model.layers.{i}.input_layernorm"]["weight"].to(torch.float16) for i in range(cfg["n_block"])],
             "ln2": [any_w[f"model.layers.{i}.post_attention_layernorm"]["weight"].to(torch.float16) for i in range(cfg["n_block"])]}
    return QuantLlama(cfg, arch_linear, device=device, max_seq=max_seq, hqq_layers=layers, dense=dense, batch=batch)
