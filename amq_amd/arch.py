"""Per-layer bit-config ("arch") handling: the JSON schema AMQ's search writes
and the selection rule amq_speed_benchmark.py applies to it.

  * schema  (amq/search/optimizer.py:166-171, amq/search/space.py:106-118):
        {"archive": [[arch, metric, bits_usage], ...], "candidates": [...], ...}
        arch = {"linear": {"self_attn.q_proj": [bits]*n_block, ... 7 keys}}
  * bits_usage (amq/utils/func.py:101-114): sum(out*in*(bits + 32/group)) / model_numel
    -- "avg 3 bits" includes the 0.25 bit of fp16 scale + zero per weight.
  * selection (amq/amq_speed_benchmark.py:209-229): keep |bits_usage - target| < 0.05,
    take the candidate with the most 4-bit layers; without a file: uniform 2/3/4.
No searched ``.stats`` file ships with the reference, so ``synthesize_arch``
draws one with the SearchSpace.sample recipe (amq/search/space.py:34-84).
"""
import json
import math

import numpy as np

LINEARS = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
           "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]


def _cfg(n_block, hidden, inter, heads, kv_heads, numel, vocab=32000, **extra):
    """``extra``: rms_norm_eps, rope_theta, rope_scaling (HF's dict, "llama3"), qkv_bias (Qwen2: q / k / v projections carry a bias)"""
    kv = hidden * kv_heads // heads
    return {
        **extra,
        "n_block": n_block, "hidden_size": hidden, "intermediate_size": inter, "num_heads": heads,
        "num_kv_heads": kv_heads, "head_dim": hidden // heads, "vocab_size": vocab, "model_numel": numel,
        "linear": list(LINEARS),
        "linear_shape": {
            "self_attn.q_proj": [hidden, hidden], "self_attn.k_proj": [kv, hidden],
            "self_attn.v_proj": [kv, hidden], "self_attn.o_proj": [hidden, hidden],
            "mlp.gate_proj": [inter, hidden], "mlp.up_proj": [inter, hidden], "mlp.down_proj": [hidden, inter],
        },
    }


_LLAMA3_SCALING = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}

# shapes / model_numel as in amq/configs/llama.json:2-79 (Llama 2), :82+ (Llama 3.x), mistral.json, qwen2.json -- the families the reference lists
# (README.md:90-92); rope / norm / bias settings as in the models' published config.json (they do not change any linear's shape)
MODEL_CONFIGS = {
    "Llama-2-7b-hf": _cfg(32, 4096, 11008, 32, 32, 6476005376),
    "Llama-2-13b-hf": _cfg(40, 5120, 13824, 40, 40, 12687769600),
    "Llama-2-70b-hf": _cfg(80, 8192, 28672, 64, 8, 68451041280),
    "Meta-Llama-3-8B": _cfg(32, 4096, 14336, 32, 8, 6979321856, vocab=128256, rope_theta=500000.0),
    "Llama-3.1-8B": _cfg(32, 4096, 14336, 32, 8, 6979321856, vocab=128256, rope_theta=500000.0, rope_scaling=_LLAMA3_SCALING),
    "Llama-3.1-8B-Instruct": _cfg(32, 4096, 14336, 32, 8, 6979321856, vocab=128256, rope_theta=500000.0, rope_scaling=_LLAMA3_SCALING),
    "Llama-3.1-70B": _cfg(80, 8192, 28672, 64, 8, 68451041280, vocab=128256, rope_theta=500000.0, rope_scaling=_LLAMA3_SCALING),
    "Mistral-7B-v0.3": _cfg(32, 4096, 14336, 32, 8, 6979321856, vocab=32768, rope_theta=1000000.0),
    "Qwen2.5-7B": _cfg(28, 3584, 18944, 28, 4, 6525288448, vocab=152064, rope_theta=1000000.0, rms_norm_eps=1e-6, qkv_bias=True),
    "Qwen2.5-14B": _cfg(48, 5120, 13824, 40, 8, 13212057600, vocab=152064, rope_theta=1000000.0, rms_norm_eps=1e-6, qkv_bias=True),
    "Qwen2.5-32B": _cfg(64, 5120, 27648, 40, 8, 31205621760, vocab=152064, rope_theta=1000000.0, rms_norm_eps=1e-6, qkv_bias=True),
    "Qwen2.5-72B": _cfg(80, 8192, 29568, 64, 8, 70212648960, vocab=152064, rope_theta=1000000.0, rms_norm_eps=1e-6, qkv_bias=True),
    # small shapes for tests / smoke (not in the reference)
    "tiny-llama-test": _cfg(2, 256, 512, 4, 4, 2 * (4 * 256 * 256 + 3 * 256 * 512), vocab=1000),
}

# Llama-2-7B layers whose measured sensitivity exceeds 2x the median are pinned to 4 bit by the
# search (amq/search/optimizer.py:53-55; list derived in SURVEY.md 3.4 from amq/sensitivity/*.json)
PINNED_7B = ["0.self_attn.v_proj", "1.self_attn.v_proj", "1.mlp.down_proj", "31.mlp.down_proj"]


def rope_inv_freq(config):
    """fp32 inverse frequencies [64] and attention scaling of a config's rotary embedding, as transformers resolves them
    (modeling_rope_utils: ``_compute_default_rope_parameters`` / ``_compute_llama3_parameters``); None for the plain rope_theta form."""
    import torch
    rs = config.get("rope_scaling") or None
    if not rs or rs.get("rope_type", rs.get("type", "default")) == "default":
        return None, 1.0
    kind = rs.get("rope_type", rs.get("type"))
    theta = float(config.get("rope_theta", 10000.0))
    dim = int(config.get("head_dim", 128))
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float32) / dim))
    if kind == "linear":
        return inv / float(rs["factor"]), 1.0
    if kind != "llama3":
        raise ValueError(f"rope_scaling type '{kind}' is not served (default, linear, llama3)")
    factor, lo, hi, old = float(rs["factor"]), float(rs["low_freq_factor"]), float(rs["high_freq_factor"]), float(rs["original_max_position_embeddings"])
    low_wavelen, high_wavelen = old / lo, old / hi
    wavelen = 2 * math.pi / inv
    out = torch.where(wavelen > low_wavelen, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * out / factor + smooth * out
    medium = ~(wavelen < high_wavelen) & ~(wavelen > low_wavelen)
    return torch.where(medium, smoothed, out), 1.0


def get_bits_usage(arch, config, group_size=128):
    """amq/utils/func.py:101-114"""
    mem = 0.0
    for linear, bits in arch["linear"].items():
        out_dim, in_dim = config["linear_shape"][linear]
        g = in_dim if group_size == -1 else group_size
        for b in bits:
            mem += int(out_dim) * int(in_dim) * (b + (32 / g if b < 16 else 0))
    return mem / config["model_numel"]


def uniform_arch(config, bits):
    return {"linear": {name: [bits] * config["n_block"] for name in config["linear"]}}


def select_arch(stats, target_bits, config=None):
    """amq_speed_benchmark.py:209-229.  ``stats``: path or the loaded dict."""
    if isinstance(stats, str):
        with open(stats) as f:
            stats = json.load(f)
    archs = stats["archive"] + stats["candidates"]
    cands = [a for a in archs if abs(a[-1] - target_bits) < 0.05]
    if not cands:
        raise ValueError(f"no arch within 0.05 of target_bits={target_bits}")
    bits = [np.concatenate([np.asarray(b) for b in a[0]["linear"].values()]) for a in cands]
    count4 = [(b == 4.0).sum() for b in bits]
    return cands[int(np.argmax(count4))][0]["linear"]


def synthesize_arch(config, target_bits=3.0, seed=0, pinned=(), bits_range=(2, 3, 4), group_size=128, tol=0.05):
    """SearchSpace.sample (amq/search/space.py:34-84) with numpy's default_rng(seed):
    per draw a random probability vector over bits_range, 7 independent
    per-linear lists, pinned layers forced to max(bits_range); accept the first
    draw with |bits_usage - target| < tol.  Returns (arch, bits_usage)."""
    rng = np.random.default_rng(seed)
    nb = config["n_block"]
    for _ in range(100000):
        prob = rng.random(len(bits_range))
        p = prob / prob.sum()
        lists = {name: rng.choice(bits_range, size=nb, p=p, replace=True).tolist() for name in config["linear"]}
        for pin in pinned:
            blk, linear = pin.split(".", maxsplit=1)
            lists[linear][int(blk)] = max(bits_range)
        arch = {"linear": {k: [int(v) for v in vals] for k, vals in lists.items()}}
        usage = get_bits_usage(arch, config, group_size)
        if abs(usage - target_bits) < tol:
            return arch, usage
    raise RuntimeError("could not synthesize an arch")


def write_stats(path, arch, bits_usage, metric=0.0):
    """A ``.stats`` file in the search's schema holding one arch (so select_arch reads it back)."""
    with open(path, "w") as f:
        json.dump({"archive": [[arch, metric, bits_usage]], "candidates": [], "iteration": 0}, f)


def arch_bits(arch_linear, name, block):
    b = arch_linear[name][block]
    for v in (2, 3, 4):
        if math.isclose(b, v):
            return v
    raise ValueError(f"bit should be 2, 3, 4, but got {b}")      # amq_speed_benchmark.py:243
