"""CPU: host-side format plumbing of the product (amq_amd/hqq_format.py, checkpoint loader) against the oracle and
the golden captures -- no GPU, no compute through the library."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from amq_amd import hqq_format
from amq_amd.checkpoint import load_hqq_dir, runner_config
from oracle import hqq_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("rows", [8, 20, 52, 410])
def test_pack_rows_equals_reference_bitpack(bits, rows):
    if (bits == 4 and rows % 2) or (bits == 2 and rows % 4):
        pytest.skip("HQQ packs 4/2-bit row chunks of equal size")
    q = torch.randint(0, 2 ** bits, (rows, 128), generator=torch.Generator().manual_seed(rows + bits), dtype=torch.int32)
    mine = hqq_format.pack_rows(q, bits).numpy()
    ref = {4: hqq_ref.pack_4bit_u8, 3: hqq_ref.pack_3bit_32, 2: hqq_ref.pack_2bit_u8}[bits](q.numpy())
    assert mine.dtype == ref.dtype and np.array_equal(mine, ref)


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_quantize_rtn_is_a_valid_hqq_layer(bits):
    w = torch.randn(64, 256, generator=torch.Generator().manual_seed(bits)) * 0.02
    h = hqq_format.quantize_rtn(w.half(), bits)
    assert h.scale.dtype == torch.float16 and h.scale.shape == (64 * 256 // 128, 1) and h.meta["packing"] == hqq_format.PACKING[bits]
    deq = hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), bits, (64, 256)).astype(np.float32)
    step = h.scale.float().numpy().repeat(128, axis=1).reshape(64, 256)
    assert np.all(np.abs(deq - w.half().float().numpy()) <= 0.51 * step + 1e-4)      # round-to-nearest within half a step


def test_from_hqq_layer_duck_typing_and_rejections():
    class HQQLinear:                                    # same attribute surface as the reference class
        pass
    h = hqq_format.random_hqq(32, 256, 3, seed=1, bias=True)
    layer = HQQLinear()
    layer.W_q, layer.meta, layer.bias, layer.name = torch.nn.Parameter(h.W_q, requires_grad=False), dict(h.meta), h.bias, "q_proj"
    got = hqq_format.from_hqq_layer(layer)
    assert got.nbits == 3 and tuple(got.shape) == (32, 256) and got.name == "q_proj" and torch.equal(got.W_q, h.W_q)
    for key, val, exc in (("axis", 0, ValueError), ("group_size", 96, ValueError), ("nbits", 8, NotImplementedError),
                          ("view_as_float", True, ValueError)):
        bad = dict(h.meta); bad[key] = val
        layer.meta = bad
        with pytest.raises(exc):
            hqq_format.from_hqq_layer(layer)


def test_from_hqq_layer_keeps_a_bf16_layer_in_bf16():
    """compute_dtype = bfloat16 layers (scale / zero in bf16, quantize.py:516) stay bf16 -- the bf16 entry points reproduce their arithmetic; any other
    meta dtype goes to fp16, the dtype of the reference's kernels"""
    class HQQLinear:
        pass
    h = hqq_format.random_hqq(32, 256, 4, seed=2)
    for src, want in ((torch.bfloat16, torch.bfloat16), (torch.float32, torch.float16), (torch.float16, torch.float16)):
        layer = HQQLinear()
        meta = dict(h.meta)
        meta["scale"], meta["zero"] = h.scale.to(src), h.zero.to(src)
        layer.W_q, layer.meta, layer.bias, layer.name = h.W_q, meta, None, "v_proj"
        got = hqq_format.from_hqq_layer(layer)
        assert got.scale.dtype == want and got.zero.dtype == want and got.scale.shape == (32 * 256 // 128, 1)
    # one bf16, one fp16: not a bf16 layer
    meta = dict(h.meta)
    meta["scale"] = h.scale.to(torch.bfloat16)
    layer.meta = meta
    assert hqq_format.from_hqq_layer(layer).scale.dtype == torch.float16


def test_checkpoint_loader_reads_reference_files():
    root = os.path.join(GOLDEN, "ckpt")
    for bits in (2, 3, 4):
        hf, w = load_hqq_dir(os.path.join(root, f"{bits}bit"))
        cfg = runner_config(hf)
        assert (cfg["hidden_size"], cfg["intermediate_size"], cfg["n_block"], cfg["vocab_size"], cfg["head_dim"]) == (256, 512, 2, 256, 128)
        q = w["model.layers.1.mlp.down_proj"]
        assert isinstance(q, hqq_format.HQQWeights) and q.nbits == bits and tuple(q.shape) == (256, 512)
        assert w["lm_head"]["weight"].shape == (256, 256) and w["model.norm"]["weight"].dtype == torch.float16
        # the payload is a valid HQQ layer: the oracle dequantizes it to finite values in the weight range
        d = hqq_ref.dequantize(q.W_q.numpy(), q.scale.numpy(), q.zero.numpy(), bits, (256, 512))
        assert np.isfinite(d.astype(np.float32)).all() and np.abs(d.astype(np.float32)).max() < 1.0
    exp = np.load(os.path.join(root, "expected.npz"))
    assert set(json.loads(str(exp["arch"]))) == set(hqq_format.PACKING and ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj",
                                                                            "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"])


def test_fuse_llama_norms_structure_state_dict_and_deepcopy():
    """patching.fuse_llama_norms (host logic only, no launch): a decoder-layer-shaped module gets its two RMSNorms wrapped, the
    state_dict keys do not change, a deep copy (amq_speed_benchmark.py:231) ties the COPIED norms to the COPIED consumers, and
    other norm classes / layers with a biased projection are left alone"""
    import copy
    import torch.nn as nn
    from amq_amd import patching
    from amq_amd.quant_linear import HIPLlamaMLP, HIPQuantLinear, HIPRMSNorm, LinearGroup

    class LlamaRMSNorm(nn.Module):
        def __init__(self, n):
            super().__init__()
            self.weight, self.variance_epsilon = nn.Parameter(torch.ones(n, dtype=torch.float16)), 1e-5

        def forward(self, x):
            return x * self.weight

    class OtherNorm(LlamaRMSNorm):
        pass

    class Attn(nn.Module):
        def __init__(self, bias=False):
            super().__init__()
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                setattr(self, n, HIPQuantLinear(4, 128, 256, 256, bias=torch.zeros(256) if bias and n == "k_proj" else None))

    class MLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj, self.down_proj = HIPQuantLinear(3, 128, 256, 512), HIPQuantLinear(2, 128, 256, 512), HIPQuantLinear(4, 128, 512, 256)
            self.act_fn = nn.SiLU()

    class Layer(nn.Module):
        def __init__(self, norm=LlamaRMSNorm, bias=False):
            super().__init__()
            self.input_layernorm, self.post_attention_layernorm = norm(256), norm(256)
            self.self_attn, self.mlp = Attn(bias), MLP()

    model = nn.ModuleList([Layer(), Layer(OtherNorm), Layer(bias=True)])
    keys = sorted(model.state_dict())
    assert patching.group_sibling_linears(model) == 6 and patching.fuse_llama_mlps(model) == 3
    assert patching.fuse_llama_norms(model) == 2 + 0 + 1          # layer 1: unknown norm class; layer 2: biased k_proj keeps input_layernorm
    assert sorted(model.state_dict()) == keys
    l0 = model[0]
    assert isinstance(l0.input_layernorm, HIPRMSNorm) and isinstance(l0.post_attention_layernorm, HIPRMSNorm)
    assert isinstance(l0.input_layernorm.__dict__["_consumer"], LinearGroup) and l0.post_attention_layernorm.__dict__["_consumer"] is l0.mlp
    assert not isinstance(model[1].input_layernorm, HIPRMSNorm) and not isinstance(model[2].input_layernorm, HIPRMSNorm)
    assert isinstance(model[2].post_attention_layernorm, HIPRMSNorm) and isinstance(l0.mlp, HIPLlamaMLP)
    c0 = copy.deepcopy(model)[0]
    assert c0.input_layernorm.__dict__["_consumer"] is c0.self_attn.q_proj.__dict__["_group"][0]
    assert c0.post_attention_layernorm.__dict__["_consumer"] is c0.mlp and c0.mlp is not l0.mlp
    assert c0.input_layernorm.weight is c0.input_layernorm.__dict__["_inner"].weight
    # CPU tensors are never deferred: the wrapped module runs
    x = torch.ones(1, 256, dtype=torch.float16)
    assert l0.input_layernorm(x) is not x
    assert patching.fuse_llama_norms(model) == 0                  # idempotent


def test_fuse_llama_layers_only_touches_known_forwards():
    """patching.fuse_llama_layers: a ``LlamaDecoderLayer`` whose forward has exactly the transformers 5.x parameter list gets the fused
    forward (instance attribute: state_dict unchanged, deepcopy keeps it bound to the COPY); another signature, or another class, is
    left alone; on CPU tensors the fused forward is the plain composition"""
    import copy
    import torch.nn as nn
    from amq_amd import patching

    class Norm(nn.Module):
        def forward(self, x):
            return x * 2

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.o_proj = nn.Linear(4, 4, bias=False)

        def forward(self, hidden_states=None, **kw):
            return self.o_proj(hidden_states), None

    class MLP(nn.Module):
        def forward(self, x):
            return x + 1

    def make(name, good_signature):
        if good_signature:
            def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                        position_embeddings=None, **kwargs):
                raise AssertionError("the original forward must not run once fused")
        else:
            def forward(self, hidden_states, attention_mask=None):
                return hidden_states
        cls = type(name, (nn.Module,), {"forward": forward})
        layer = cls()
        layer.input_layernorm, layer.post_attention_layernorm, layer.self_attn, layer.mlp = Norm(), Norm(), Attn(), MLP()
        return layer

    model = nn.ModuleList([make("LlamaDecoderLayer", True), make("LlamaDecoderLayer", False), make("OtherLayer", True)])
    keys = sorted(model.state_dict())
    assert patching.fuse_llama_layers(model) == 1 and patching.fuse_llama_layers(model) == 0
    assert "forward" in model[0].__dict__ and "forward" not in model[1].__dict__ and "forward" not in model[2].__dict__
    assert sorted(model.state_dict()) == keys
    x = torch.randn(2, 4)
    a = x + model[0].self_attn.o_proj(x * 2)                      # residual + attn(norm(x))
    want = a + (a * 2 + 1)                                        # residual + mlp(norm(.))
    assert torch.allclose(model[0](x), want)
    c = copy.deepcopy(model)[0]
    assert c.forward.__self__ is c and torch.allclose(c(x), want)
    assert "_residual" not in model[0].self_attn.o_proj.__dict__


def test_allow_merge_walk_matches_the_reference_skip_rule(capsys):
    """prepare_for_inference(allow_merge=True): grouped layers are skipped with the reference's message (patching.py:243-245),
    a group-less HQQLinearLoRA is refused"""
    import torch.nn as nn
    from amq_amd import patching

    class HQQLinearLoRA(nn.Module):
        def __init__(self, group_size):
            super().__init__()
            self.linear_layer = nn.Identity()
            self.linear_layer.meta = {"axis": 1, "group_size": group_size}
            self.name = "q_proj"

    patching._merge_zeros_with_lora(nn.ModuleList([HQQLinearLoRA(128)]))
    assert "Skipping zeros lora merging for q_proj" in capsys.readouterr().out
    with pytest.raises(NotImplementedError, match="without groups"):
        patching._merge_zeros_with_lora(nn.ModuleList([HQQLinearLoRA(None)]))
