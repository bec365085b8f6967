"""CPU: bit-config (arch) schema, bits_usage and selection rule (amq_speed_benchmark.py:209-229, func.py:101-114)."""
import json

import numpy as np
import pytest

from amq_amd import arch


def test_bits_usage_matches_reference_definition():
    cfg = arch.MODEL_CONFIGS["Llama-2-7b-hf"]
    for b in (2, 3, 4):
        assert arch.get_bits_usage(arch.uniform_arch(cfg, b), cfg) == pytest.approx(b + 0.25)
    # shapes as amq/configs/llama.json: layer numel sums to model_numel
    tot = sum(n * k for n, k in cfg["linear_shape"].values()) * cfg["n_block"]
    assert tot == cfg["model_numel"] == 6476005376
    assert arch.MODEL_CONFIGS["Llama-2-70b-hf"]["linear_shape"]["self_attn.k_proj"] == [1024, 8192]
    assert sum(n * k for n, k in arch.MODEL_CONFIGS["Llama-2-13b-hf"]["linear_shape"].values()) * 40 == 12687769600


def test_synthesize_and_select_roundtrip(tmp_path):
    cfg = arch.MODEL_CONFIGS["Llama-2-7b-hf"]
    a, usage = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=arch.PINNED_7B)
    assert abs(usage - 3.0) < 0.05
    assert set(a["linear"]) == set(arch.LINEARS) and all(len(v) == 32 for v in a["linear"].values())
    assert a["linear"]["self_attn.v_proj"][0] == 4 and a["linear"]["mlp.down_proj"][31] == 4
    a2, _ = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=arch.PINNED_7B)
    assert a == a2                                            # deterministic
    p = tmp_path / "iter_0.stats"
    arch.write_stats(str(p), a, usage)
    assert arch.select_arch(str(p), 3.0) == a["linear"]
    with pytest.raises(ValueError):
        arch.select_arch(str(p), 2.0)


def test_select_prefers_most_4bit():
    cfg = arch.MODEL_CONFIGS["tiny-llama-test"]
    lo = {"linear": {k: [2, 4] for k in arch.LINEARS}}
    hi = {"linear": {k: [4, 2] if i else [4, 4] for i, k in enumerate(arch.LINEARS)}}
    stats = {"archive": [[lo, 0.1, 3.25]], "candidates": [[hi, 0.2, 3.27], [lo, 0.3, 9.0]]}
    assert arch.select_arch(stats, 3.25) == hi["linear"]
    assert arch.arch_bits(hi["linear"], "mlp.up_proj", 1) == 2
    with pytest.raises(ValueError):
        arch.arch_bits({"x": [5]}, "x", 0)
