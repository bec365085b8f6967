"""q / k / v GEMV + decode attention as ONE launch (amq_gemv_qkv_attn_f16, amq_gemv.hip: gemv_qkv_attn_kernel) against the two
separate launches it replaces (amq_gemv_grouped_f16 with the RMSNorm prologue + amq_attn_decode_cur_f16): bit-identical q / k /
v vectors, attention output, cache rows, and -- through the runner -- logits and tokens over 32 steps.  Reference counterpart:
the q/k/v projections and single-query attention of the patched decoder layer, amq/kernel/monkeypatch/ftllama_modeling.py:127-164."""
import numpy as np
import pytest
import torch

def _ab_built():
    from amq_amd import _lib
    import os
    return os.path.exists(_lib.AB_LIB_PATH)


# the A/B routes (include/amq_hip_ab.h: measured negatives) live in libamq_hip_ab.so, which the default build no longer makes (`make -C amq_amd/csrc ab`)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _ab_built(), reason="libamq_hip_ab.so not built (make -C amq_amd/csrc ab): A/B routes only")]


def _dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _exact_math():
    """the A/B kernels of this module carry the EXACT two-rounding bodies only: the step they are compared with bit for bit runs the same arithmetic"""
    from amq_amd import ops
    old = ops.DEFAULT_GEMV_OPTS
    ops.DEFAULT_GEMV_OPTS = ops.GemvOpts(math=ops.MATH_EXACT)
    yield
    ops.DEFAULT_GEMV_OPTS = old


def _pair(cfg, al, max_seq, seed):
    from amq_amd.llama import QuantLlama
    mf = QuantLlama(cfg, al, device="cuda:0", max_seq=max_seq, seed=seed, engine=False)
    ms = QuantLlama(cfg, al, device="cuda:0", max_seq=max_seq, seed=seed, engine=False)
    assert mf.can_fuse_qkv_attn
    mf.fuse_qkv_attn, ms.fuse_qkv_attn = True, False
    return mf, ms


def _run(cfg, al, max_seq, prompt, steps, seed, graph_from=2):
    mf, ms = _pair(cfg, al, max_seq, seed)
    ids = torch.randint(0, cfg["vocab_size"] - 1, (prompt,), generator=torch.Generator().manual_seed(seed)).to(_dev())
    assert torch.equal(mf.prefill(ids), ms.prefill(ids))
    for step in range(steps):
        g = step >= graph_from
        mf.decode_step(use_graph=g)
        ms.decode_step(use_graph=g)
        assert torch.equal(mf.q, ms.q) and torch.equal(mf.k, ms.k) and torch.equal(mf.v, ms.v), f"q/k/v differ at step {step}"
        assert torch.equal(mf.att, ms.att), f"attention output differs at step {step}"
        assert torch.equal(mf.logits, ms.logits) and torch.equal(mf.token, ms.token), f"logits differ at step {step}"
    mf.check()
    for bf, bs in zip(mf.blocks, ms.blocks):
        assert torch.equal(bf["kc"], bs["kc"]) and torch.equal(bf["vc"], bs["vc"])
    assert int(mf._tickets.abs().sum().item()) == 0            # every launch leaves its tickets zero
    return mf


@pytest.mark.parametrize("gqa", [False, True])
def test_fused_qkv_attention_equals_two_launches_over_32_steps(gqa):
    from amq_amd import arch
    cfg = dict(arch._cfg(3, 512, 1024, 4, 2 if gqa else 4, 1, vocab=1024))
    rng = np.random.default_rng(3)
    al = {name: [int(b) for b in rng.choice([2, 3, 4], size=3)] for name in cfg["linear"]}
    _run(cfg, al, max_seq=96, prompt=20, steps=32, seed=5)


def test_fused_qkv_attention_context_beyond_register_rows():
    """more cached keys than the 128 the attention role holds in registers: the rest continue from memory, same bits"""
    from amq_amd import arch
    cfg = dict(arch._cfg(2, 512, 1024, 4, 4, 1, vocab=1024))
    _run(cfg, None, max_seq=400, prompt=300, steps=10, seed=1)


@pytest.mark.parametrize("model", ["Llama-2-7b-hf", "Llama-2-13b-hf", "Llama-2-70b-hf"])
def test_fused_qkv_attention_real_shapes_one_block(model):
    """one block at the real shapes: 7B (rpt 1, 768 workgroups), 13B (5120: two x chunks per thread, rpt 2), 70B (GQA 64 / 8)"""
    from amq_amd import arch
    cfg = dict(arch.MODEL_CONFIGS[model])
    cfg["n_block"] = 1
    cfg["vocab_size"] = 2048
    al = {name: [b] for name, b in zip(cfg["linear"], [3, 2, 4, 3, 2, 3, 4])}
    _run(cfg, al, max_seq=80, prompt=17, steps=6, seed=2)


def test_fused_qkv_attention_position_guard():
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2, 1, vocab=1024))
    m = QuantLlama(cfg, None, device="cuda:0", max_seq=16, seed=4, engine=False)
    assert m.can_fuse_qkv_attn
    m.fuse_qkv_attn = True
    m.prefill(torch.randint(0, 1024, (13,), generator=torch.Generator().manual_seed(1)).to(_dev()))
    for _ in range(3):
        m.decode_step()
    m.check()
    kc = [b["kc"].clone() for b in m.blocks]
    for _ in range(2):
        m.graph.replay()                      # past the cache: attention skipped, sticky error word, tickets still cleaned
    torch.cuda.synchronize()
    assert int(m.step_err.item()) == 1 and int(m.pos.item()) == 16
    assert int(m._tickets.abs().sum().item()) == 0
    for b, k0 in zip(m.blocks, kc):
        assert torch.equal(b["kc"], k0)
