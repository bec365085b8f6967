"""The one-launch-per-token decode engine (amq_engine.hip, amq_decode_engine_f16) against the five-launch step.

Reference counterpart: the per-token loop of amq/kernel/monkeypatch/ftllama_modeling.py:167-230.  The engine must be
BIT-IDENTICAL to the same step issued as separate launches -- q/k/v, o_proj, gate/up, down_proj through
amq_gemv_grouped_f16 with amq_gemv_opts.waves = 16 (the engine's workgroup is the 16-wave form of that kernel), attention
through amq_attn_decode_cur_f16 -- which tests/test_gpu_decode.py / test_gpu_kernels.py compare with the oracle.  Every
step's logits, the greedy tokens and the KV caches are compared bit for bit, MHA and GQA, 2/3/4-bit layers mixed, graph
replay and eager launches, full grid and small grids (several row-tiles and segment boundaries inside one workgroup).
"""
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


def _pair(cfg, al, max_seq, seed, grid=0):
    """(engine runner, five-launch runner) over the same synthetic weights"""
    from amq_amd.llama import QuantLlama
    me = QuantLlama(cfg, al, device="cuda:0", max_seq=max_seq, seed=seed, engine=True)
    if grid:
        me.engine.grid = grid
    mr = QuantLlama(cfg, al, device="cuda:0", max_seq=max_seq, seed=seed, engine=False)
    mr.fuse_qkv_attn = False                                    # five separate launches (the fused q/k/v + attention launch is 8-wave only)
    for be, br in zip(me.blocks, mr.blocks):                    # same generator, same draws
        for name in cfg["linear"]:
            assert torch.equal(be[name].qn, br[name].qn) and torch.equal(be[name].mn, br[name].mn)
    return me, mr


def _step_ref(mr, use_graph):
    """the five-launch step with 16-wave GEMV workgroups (what the engine's stages are)"""
    from amq_amd import ops
    old = ops.DEFAULT_GEMV_OPTS
    ops.DEFAULT_GEMV_OPTS = ops.GemvOpts(waves=16, math=ops.MATH_EXACT)
    try:
        mr.decode_step(use_graph=use_graph)
    finally:
        ops.DEFAULT_GEMV_OPTS = old


def _compare_run(cfg, al, max_seq, prompt, steps, seed, grid=0, graph_from=2):
    me, mr = _pair(cfg, al, max_seq, seed, grid)
    ids = torch.randint(0, cfg["vocab_size"] - 1, (prompt,), generator=torch.Generator().manual_seed(seed)).to(_dev())
    le = me.prefill(ids).clone()
    lr = mr.prefill(ids).clone()
    assert torch.equal(le, lr)
    for step in range(steps):
        g = step >= graph_from
        me.decode_step(use_graph=g)
        _step_ref(mr, g)
        assert torch.equal(me.logits, mr.logits), f"logits differ at step {step}"
        assert torch.equal(me.token, mr.token), f"token differs at step {step}"
        assert torch.equal(me.x, mr.x), f"next embedding differs at step {step}"
    me.check()
    mr.check()
    for be, br in zip(me.blocks, mr.blocks):
        assert torch.equal(be["kc"], br["kc"]) and torch.equal(be["vc"], br["vc"])
    assert torch.isfinite(me.logits.float()).all()
    return me, mr


@pytest.mark.parametrize("gqa", [False, True])
def test_engine_equals_five_launch_step_over_32_steps(gqa):
    """VERDICT r2 item 1's test: engine == 5-launch graph over 32 steps, MHA and GQA (mixed 2/3/4-bit layers; the first two
    steps eager, the rest replayed from a hipGraph on both sides)"""
    from amq_amd import arch
    cfg = dict(arch._cfg(3, 512, 1408, 4, 2 if gqa else 4, 1, vocab=1024))      # 1408 = 11 k-tiles: waves 11 .. 15 idle in down_proj
    rng = np.random.default_rng(7)
    al = {name: [int(b) for b in rng.choice([2, 3, 4], size=3)] for name in cfg["linear"]}
    _compare_run(cfg, al, max_seq=96, prompt=20, steps=32, seed=11)


@pytest.mark.parametrize("grid", [1, 3, 7, 32, 100])
def test_engine_small_grids_cross_segment_and_rowtile_boundaries(grid):
    """with few workgroups one workgroup owns many row-tiles, its range crosses the q|k|v and gate|up segment boundaries
    (different bit-widths inside one stage of one workgroup) and more than 16 row-tiles reach the generic epilogue path"""
    from amq_amd import arch
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2, 1, vocab=1024))
    al = {"self_attn.q_proj": [4, 2], "self_attn.k_proj": [3, 4], "self_attn.v_proj": [2, 3], "self_attn.o_proj": [3, 2],
          "mlp.gate_proj": [2, 4], "mlp.up_proj": [4, 3], "mlp.down_proj": [3, 4]}
    if grid < 4:
        # fewer workgroups than heads: refused loudly on the host (the attention stage needs one workgroup per head)
        from amq_amd import _lib
        me, _ = _pair(cfg, al, 64, 2, grid)
        me.prefill(torch.arange(5, device=_dev()))
        with pytest.raises(_lib.AmqError, match="one workgroup per head"):
            me.decode_step(use_graph=False)
        return
    _compare_run(cfg, al, max_seq=64, prompt=9, steps=6, seed=2, grid=grid, graph_from=3)


def test_engine_context_beyond_register_prefetch():
    """contexts longer than the 192 keys the attention stage holds in registers continue from memory: same bits"""
    from amq_amd import arch
    cfg = dict(arch._cfg(2, 512, 1024, 4, 4, 1, vocab=1024))
    _compare_run(cfg, None, max_seq=300, prompt=250, steps=12, seed=5)


def test_engine_7b_shapes_one_block_at_size():
    """one decoder block at Llama-2-7B shapes (4096 / 11008, 32 heads) on the full grid: every workgroup of the real launch
    geometry takes part (3 / 1 / 5-6 / 1 row-tiles per workgroup)"""
    from amq_amd import arch
    cfg = dict(arch._cfg(1, 4096, 11008, 32, 32, 1, vocab=2048))
    al = {"self_attn.q_proj": [3], "self_attn.k_proj": [2], "self_attn.v_proj": [4], "self_attn.o_proj": [3],
          "mlp.gate_proj": [2], "mlp.up_proj": [3], "mlp.down_proj": [4]}
    _compare_run(cfg, al, max_seq=80, prompt=17, steps=8, seed=3)


def test_engine_position_guard_and_error_words():
    """a graph replayed past the cache: the engine skips the attention stage (no cache write, no LDS score write), raises
    the sticky word of the step state, and no barrier times out"""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2, 1, vocab=1024))
    m = QuantLlama(cfg, None, device="cuda:0", max_seq=16, seed=4, engine=True)
    m.prefill(torch.randint(0, 1024, (13,), generator=torch.Generator().manual_seed(1)).to(_dev()))
    for _ in range(3):
        m.decode_step()
    m.check()
    kc = [b["kc"].clone() for b in m.blocks]
    for _ in range(2):
        m.graph.replay()
    torch.cuda.synchronize()
    assert int(m.step_err.item()) == 1 and int(m.pos.item()) == 16
    m.engine.check()                                      # barriers fine
    for b, k0 in zip(m.blocks, kc):
        assert torch.equal(b["kc"], k0)


def test_engine_refuses_what_it_does_not_run():
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 4, 1, vocab=1024))
    with pytest.raises(ValueError, match="decode engine"):
        QuantLlama(cfg, None, device="cuda:0", max_seq=4096, seed=0, engine=True)        # long caches: the split attention path
    with pytest.raises(ValueError, match="decode engine"):
        QuantLlama(cfg, None, device="cuda:0", max_seq=64, seed=0, batch=2, engine=True)
    assert QuantLlama(cfg, None, device="cuda:0", max_seq=4096, seed=0).engine is None   # auto: five launches
