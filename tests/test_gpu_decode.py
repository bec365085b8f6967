"""GPU: decode-step surroundings and the token-step runner vs plain torch references."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rms_ref(x, g, eps):
    xf = x.float()
    return g * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(torch.float16)


def test_rmsnorm_matches_llama_rmsnorm():
    from amq_amd import ops
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(5, 4096, generator=g) * 3).half().to(_dev())
    gamma = (1 + 0.1 * torch.randn(4096, generator=g)).half().to(_dev())
    y = ops.rmsnorm(x, gamma, 1e-5)
    ref = _rms_ref(x, gamma, 1e-5)
    assert torch.allclose(y.float(), ref.float(), rtol=2e-3, atol=2e-3)
    assert (y != ref).float().mean() < 0.02          # same two-rounding arithmetic; rsqrt/sum order may flip an ulp


@pytest.mark.parametrize("norm", [False, True])
def test_gemv_f16w(norm):
    from amq_amd import ops
    g = torch.Generator().manual_seed(1)
    n, k = 1000, 1024
    w = (torch.randn(n, k, generator=g) * 0.03).half().to(_dev())
    x = torch.randn(k, generator=g).half().to(_dev())
    gamma = (1 + 0.1 * torch.randn(k, generator=g)).half().to(_dev())
    y = ops.gemv_f16w(x, w, gamma=gamma if norm else None, eps=1e-5)
    xin = _rms_ref(x[None], gamma, 1e-5)[0] if norm else x
    ref = (w.double() @ xin.double())
    err = (y.double() - ref).abs()
    assert torch.all(err <= 1e-3 * ref.abs() + 1e-3 * ref.pow(2).mean().sqrt())


def test_decode_tail():
    """argmax (first maximum) + position increment + embedding gather in one launch"""
    from amq_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    for vocab, hidden in ((32000, 4096), (1003, 256)):
        embed = torch.randn(vocab, hidden, generator=g).half().to(dev)
        logits = torch.randn(vocab, generator=g).half().to(dev)
        logits[vocab // 3] = 9.0
        logits[vocab // 3 + 7] = 9.0                    # tie: the first one wins
        token = torch.zeros(1, dtype=torch.int64, device=dev)
        pos = torch.full((1,), 41, dtype=torch.int32, device=dev)
        x = torch.zeros(1, hidden, dtype=torch.float16, device=dev)
        ops.decode_tail(logits, embed, token, pos, x)
        assert int(token.item()) == vocab // 3 == int(torch.argmax(logits.float().cpu()).item())
        assert int(pos.item()) == 42
        assert torch.equal(x[0], embed[vocab // 3])
        # with the rope hand-off: cur = table row of the NEW position (clamped to the last row)
        tab = ops.rope_table(44, 10000.0, dev)
        cur = torch.zeros(128, dtype=torch.float16, device=dev)
        for expect_row in (43, 43):
            ops.decode_tail(logits, embed, token, pos, x, table=tab, cur=cur)
            assert torch.equal(cur, tab.view(44, 128)[expect_row])
        assert int(pos.item()) == 44


def _rope_ref(t, pos):
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32, device=t.device) / 128.0))
    fr = torch.tensor([float(pos)], device=t.device)[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().half(), emb.sin().half()
    rot = torch.cat([-t[..., 64:], t[..., :64]], -1)
    return t * cos + rot * sin


@pytest.mark.parametrize("batch,seq,nh,nkv,pos0", [(1, 5, 4, 4, 0), (3, 37, 8, 2, 0), (2, 64, 5, 1, 11), (4, 300, 32, 8, 0)])
def test_rope_rows_matches_hf_expression(batch, seq, nh, nkv, pos0):
    """amq_rope_rows_f16 (in place on the q / k projections of a batched prompt pass) == HF apply_rotary_pos_emb in fp16,
    bit for bit: t * cos + rotate_half(t) * sin with the table's fp16 cos / sin, position = pos0 + row % seq"""
    from amq_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(batch * 100 + seq)
    rows = batch * seq
    q = torch.randn(rows, nh * 128, generator=g).half().to(dev)
    k = torch.randn(rows, nkv * 128, generator=g).half().to(dev)
    tab = ops.rope_table(pos0 + seq + 3, 10000.0, dev)
    q2, k2 = q.clone(), k.clone()
    ops.rope_rows(q2, k2, tab, seq, nh, nkv, pos0=pos0)
    for r in sorted({0, 1, seq - 1, seq % rows, rows // 2, rows - 1}):
        pos = pos0 + r % seq
        assert torch.equal(q2[r].view(nh, 128), _rope_ref(q[r].view(nh, 128), pos)), (r, pos)
        assert torch.equal(k2[r].view(nkv, 128), _rope_ref(k[r].view(nkv, 128), pos)), (r, pos)
    # every row against the table-driven expression at once
    cs = tab.view(-1, 64, 2)[pos0 + (torch.arange(rows, device=dev) % seq)]                 # [rows, 64, 2]
    cos = torch.cat([cs[..., 0], cs[..., 0]], -1)[:, None, :]; sin = torch.cat([cs[..., 1], cs[..., 1]], -1)[:, None, :]
    for t, t2, h in ((q, q2, nh), (k, k2, nkv)):
        tv = t.view(rows, h, 128)
        rot = torch.cat([-tv[..., 64:], tv[..., :64]], -1)
        assert torch.equal(t2.view(rows, h, 128), tv * cos + rot * sin)


@pytest.mark.parametrize("nh,nkv", [(4, 4), (8, 2)])
def test_attn_decode_sequence(nh, nkv):
    """feed tokens one by one; compare each step with eager HF-style attention over the running cache"""
    from amq_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    max_seq = 48
    kc = torch.zeros(1, nkv, max_seq, 128, dtype=torch.float16, device=dev)
    vc = torch.zeros_like(kc)
    out = torch.zeros(1, nh * 128, dtype=torch.float16, device=dev)
    ks, vs = [], []
    posd = torch.zeros(1, dtype=torch.int32, device=dev)
    for pos in range(40):
        q = torch.randn(1, nh, 128, generator=g).half().to(dev)
        k = torch.randn(1, nkv, 128, generator=g).half().to(dev)
        v = torch.randn(1, nkv, 128, generator=g).half().to(dev)
        if pos % 2:
            posd.fill_(pos)
            ops.attn_decode(q.reshape(1, -1), k.reshape(1, -1), v.reshape(1, -1), kc, vc, out, posd, nh, nkv,
                            table=ops.rope_table(max_seq, 10000.0, dev) if pos % 4 == 1 else None)
        else:
            ops.attn_decode(q.reshape(1, -1), k.reshape(1, -1), v.reshape(1, -1), kc, vc, out, pos, nh, nkv)
        ks.append(_rope_ref(k[0], pos)); vs.append(v[0])
        K = torch.stack(ks, 1).repeat_interleave(nh // nkv, 0)     # [nh, T, 128]
        V = torch.stack(vs, 1).repeat_interleave(nh // nkv, 0)
        qr = _rope_ref(q[0], pos)                                   # [nh, 128]
        w = (torch.matmul(qr[:, None, :], K.transpose(1, 2)) * (128 ** -0.5))
        p = torch.softmax(w.float(), -1).half()
        ref = torch.matmul(p, V)[:, 0, :].reshape(-1)
        assert torch.allclose(out[0].float(), ref.float(), rtol=1e-2, atol=3e-3), (pos, (out[0].float() - ref.float()).abs().max())
        # the cache holds the rotated keys / raw values
        assert torch.equal(kc[0, :, pos], ks[-1]) and torch.equal(vc[0, :, pos], vs[-1])


@pytest.mark.parametrize("pos,max_seq,nh,nkv,batch,n_splits", [
    (100, 2048, 8, 2, 2, 6),          # one active chunk: bit-identical to the single-workgroup kernel
    (255, 2048, 8, 8, 1, 0),          # T = 256: still one chunk
    (256, 2048, 8, 2, 1, 0),          # T = 257: second chunk holds only the new token
    (511, 2048, 4, 4, 3, 0), (1500, 2048, 8, 2, 2, 0), (2047, 2048, 8, 4, 1, 0),
    (1500, 2048, 16, 8, 1, 0),        # 8 kv heads: the kv-group -> XCD head mapping
    (4000, 4096, 4, 1, 1, 0),         # 15 chunks: the last arriver combines them in two batches of O rows
    (4000, 4096, 32, 32, 1, 0),       # the auto policy's whole rounds of the chip: 16 x 32 workgroups
    (3000, 4096, 4, 2, 1, 4)])        # chunks of 768 keys: beyond the register prefetch, remainder loop per chunk
def test_attn_decode_split_context(pos, max_seq, nh, nkv, batch, n_splits):
    """amq_attn_decode_split_f16 (several workgroups per head, last-arriver combine) against the single-workgroup kernel and
    the eager fp32 formula; cache append, determinism across launches, tickets left zero; both position sources."""
    from amq_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(7 * pos + max_seq)
    kc = torch.zeros(batch, nkv, max_seq, 128, dtype=torch.float16, device=dev)
    vc = torch.zeros_like(kc)
    kc[:, :, :pos] = torch.randn(batch, nkv, pos, 128, generator=g).half().to(dev)
    vc[:, :, :pos] = torch.randn(batch, nkv, pos, 128, generator=g).half().to(dev)
    kc[:, :, pos + 1:] = float("nan"); vc[:, :, pos + 1:] = float("nan")        # rows that must never contribute
    q = torch.randn(batch, nh * 128, generator=g).half().to(dev)
    k = torch.randn(batch, nkv * 128, generator=g).half().to(dev)
    v = torch.randn(batch, nkv * 128, generator=g).half().to(dev)
    tab = ops.rope_table(max_seq, 10000.0, dev)
    posd = torch.full((1,), pos, dtype=torch.int32, device=dev)

    def run(ns, cur_mode):
        kc_, vc_ = kc.clone(), vc.clone()
        out = torch.zeros(batch, nh * 128, dtype=torch.float16, device=dev)
        if cur_mode:
            cur, pos_state, err = ops.new_step_state(dev)
            cur.copy_(tab.view(max_seq, 128)[pos]); pos_state.fill_(pos)
            ops.attn_decode(q, k, v, kc_, vc_, out, pos_state, nh, nkv, cur=cur, n_splits=ns)
            assert int(err.item()) == 0
        else:
            ops.attn_decode(q, k, v, kc_, vc_, out, posd, nh, nkv, table=tab, n_splits=ns)
        return out, kc_, vc_

    one, kc1, vc1 = run(1, False)
    got, kc2, vc2 = run(n_splits, False)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(kc2[:, :, :pos + 1], kc1[:, :, :pos + 1]) and torch.equal(vc2[:, :, :pos + 1], vc1[:, :, :pos + 1])   # appended once, same row
    ns = n_splits or ops.attn_decode_splits(max_seq, nh, batch, nkv)
    chunk = max(256, (((pos + 1 + ns - 1) // ns) + 31) // 32 * 32)
    grouped = 2 <= nh // nkv <= 16                                     # served by attn_decode_gqa_kernel (matrix cores, the prompt kernel's numerics)
    if pos + 1 <= chunk and not grouped:
        assert torch.equal(got, one)                       # one active chunk: the single-workgroup kernel's bits
    else:
        assert (got.float() - one.float()).abs().max() <= 2e-3 * one.float().abs().max() + 1e-3
    again, _, _ = run(n_splits, False)
    assert torch.equal(again, got)                         # chunk-order combine: deterministic
    junk = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
    junk.fill_(1)                                          # ... also with the cache rows and the workspace cold in HBM
    cold, _, _ = run(n_splits, False)
    assert torch.equal(cold, got)
    del junk
    cur_out, _, _ = run(n_splits, True)
    assert torch.equal(cur_out, got)                       # step-state position / rotation source: same bits
    assert all(int(t.abs().sum().item()) == 0 for t in ops._ATTN_TICKETS._cur.values())
    # eager fp32 formula over the cache after the append
    K = kc2[:, :, :pos + 1].repeat_interleave(nh // nkv, 1).float()                 # [B, nh, T, 128]
    V = vc2[:, :, :pos + 1].repeat_interleave(nh // nkv, 1).float()
    qr = torch.stack([_rope_ref(q[b].view(nh, 128), pos) for b in range(batch)]).float()   # [B, nh, 128]
    w = torch.einsum("bhd,bhtd->bht", qr, K) * (128 ** -0.5)
    ref = torch.einsum("bht,bhtd->bhd", torch.softmax(w, -1), V).reshape(batch, -1)
    assert (got.float() - ref).abs().max() <= 4e-3 * ref.abs().max() + 1e-3


@pytest.mark.parametrize("pos,max_seq,nh,nkv,batch,n_splits", [
    (1500, 2048, 8, 2, 2, 0),         # G = 4 (Llama-3.x / Mistral), two sequences
    (3000, 4096, 28, 4, 1, 0),        # G = 7 (Qwen2.5-7B)
    (4000, 4096, 64, 8, 3, 0),        # G = 8 (70B / Qwen2.5-72B), three sequences
    (2500, 4096, 10, 2, 1, 0),        # G = 5 (Qwen2.5-14B / 32B)
    (700, 4096, 6, 3, 1, 16),         # G = 2, chunks of 256: three active of sixteen
    (100, 2048, 8, 2, 1, 8),          # one active chunk: written normalised by its workgroup, no ticket
    (0, 2048, 8, 2, 1, 0),            # the first token of a sequence: nothing cached, every LDS row comes from the new key / value
    (63, 2048, 4, 1, 1, 32), (64, 2048, 4, 1, 1, 32), (65, 2048, 4, 1, 1, 32),        # one tile per workgroup: the new token last in its tile / first of the next
    (255, 1024, 8, 4, 1, 8), (256, 1024, 8, 4, 1, 8),                                 # two tiles per workgroup
    (1023, 1024, 32, 2, 1, 0),        # G = 16: a full MFMA row block; the last row of the cache
    (8000, 8192, 32, 8, 1, 0),        # 32 chunks of two stages
    (3000, 4096, 4, 1, 1, 4),         # chunks of 1024 keys: eight stages per workgroup, the running (m, l, O) across them
    (2999, 4096, 12, 4, 2, 3),        # chunks of 1408 keys (eleven stages), the last one ends mid-stage; two sequences
    (30000, 32768, 8, 2, 1, 0),       # 118 active chunks of 256 keys
    (30000, 32768, 8, 2, 1, 8)])      # eight chunks of 32 stages
def test_attn_decode_gqa_kernel(pos, max_seq, nh, nkv, batch, n_splits):
    """grouped-query heads over a long cache: ONE workgroup per (kv head, chunk) scores the chunk against all the group's query heads on the matrix cores
    (attn_decode_gqa_kernel) -- against the per-query-head split kernel (the same call with every query head given its own copy of the kv head's
    cache rows: n_kv_heads = n_heads) and the eager fp32 formula; cache append once, tickets left zero, deterministic, both position sources."""
    from amq_amd import ops
    dev = _dev()
    G = nh // nkv
    g = torch.Generator().manual_seed(11 * pos + nh)
    kc = torch.zeros(batch, nkv, max_seq, 128, dtype=torch.float16, device=dev)
    vc = torch.zeros_like(kc)
    kc[:, :, :pos] = torch.randn(batch, nkv, pos, 128, generator=g).half().to(dev)
    vc[:, :, :pos] = torch.randn(batch, nkv, pos, 128, generator=g).half().to(dev)
    kc[:, :, pos:] = float("nan"); vc[:, :, pos:] = float("nan")                # rows that must never contribute (the new token's row is written by the call)
    q = torch.randn(batch, nh * 128, generator=g).half().to(dev)
    k = torch.randn(batch, nkv * 128, generator=g).half().to(dev)
    v = torch.randn(batch, nkv * 128, generator=g).half().to(dev)
    tab = ops.rope_table(max_seq, 10000.0, dev)
    posd = torch.full((1,), pos, dtype=torch.int32, device=dev)
    ns = n_splits or ops.attn_decode_splits(max_seq, nh, batch, nkv)
    assert ns > 1

    def run(cur_mode, table=tab):
        kc_, vc_ = kc.clone(), vc.clone()
        out = torch.zeros(batch, nh * 128, dtype=torch.float16, device=dev)
        if cur_mode:
            cur, pos_state, err = ops.new_step_state(dev)
            cur.copy_(tab.view(max_seq, 128)[pos]); pos_state.fill_(pos)
            ops.attn_decode(q, k, v, kc_, vc_, out, pos_state, nh, nkv, cur=cur, n_splits=ns)
            assert int(err.item()) == 0
        else:
            ops.attn_decode(q, k, v, kc_, vc_, out, posd, nh, nkv, table=table, n_splits=ns)
        return out, kc_, vc_

    got, kc_g, vc_g = run(False)
    assert torch.isfinite(got.float()).all()
    # the appended row: HF's rotation of the new key, the raw value; nothing else written
    for b_ in range(batch):
        assert torch.equal(kc_g[b_, :, pos], _rope_ref(k[b_].view(nkv, 128), pos)) and torch.equal(vc_g[b_, :, pos], v[b_].view(nkv, 128))
    assert torch.equal(kc_g[:, :, :pos], kc[:, :, :pos]) and torch.equal(vc_g[:, :, :pos], vc[:, :, :pos])
    assert torch.isnan(kc_g[:, :, pos + 1:]).all()
    assert all(int(t.abs().sum().item()) == 0 for t in ops._ATTN_TICKETS._cur.values())
    # the per-query-head kernels over per-head copies of the cache and of the new key / value (n_kv_heads = n_heads: never the grouped form)
    kc_h, vc_h = kc.repeat_interleave(G, 1).contiguous(), vc.repeat_interleave(G, 1).contiguous()
    k_h = k.view(batch, nkv, 128).repeat_interleave(G, 1).reshape(batch, -1).contiguous()
    v_h = v.view(batch, nkv, 128).repeat_interleave(G, 1).reshape(batch, -1).contiguous()
    out_h = torch.zeros_like(got)
    ops.attn_decode(q, k_h, v_h, kc_h, vc_h, out_h, posd, nh, nh, table=tab, n_splits=max(2, -(-max_seq // 272)))
    assert (got.float() - out_h.float()).abs().max() <= 2e-3 * out_h.float().abs().max() + 1e-3
    again, _, _ = run(False)
    assert torch.equal(again, got)                         # fixed merge orders: deterministic
    junk = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
    junk.fill_(1)                                          # ... also with the cache rows and the workspace cold in HBM
    cold, _, _ = run(False)
    assert torch.equal(cold, got)
    del junk
    cur_out, kc_c, _ = run(True)
    assert torch.equal(cur_out, got) and torch.equal(kc_c[:, :, pos], kc_g[:, :, pos])    # step-state position / rotation source: same bits
    if pos < 4096:
        th_out, kc_t, _ = run(False, table=None)           # cos / sin computed in the kernel from rope_theta: the table's values
        assert torch.equal(th_out, got) and torch.equal(kc_t[:, :, pos], kc_g[:, :, pos])
    # eager fp32 formula over the cache after the append
    K = kc_g[:, :, :pos + 1].repeat_interleave(G, 1).float()
    V = vc_g[:, :, :pos + 1].repeat_interleave(G, 1).float()
    qr = torch.stack([_rope_ref(q[b_].view(nh, 128), pos) for b_ in range(batch)]).float()
    w = torch.einsum("bhd,bhtd->bht", qr, K) * (128 ** -0.5)
    ref = torch.einsum("bht,bhtd->bhd", torch.softmax(w, -1), V).reshape(batch, -1)
    assert (got.float() - ref).abs().max() <= 4e-3 * ref.abs().max() + 1e-3


def test_attn_decode_split_position_guard():
    """a device-side position outside the cache: no append, no output, sticky error word (as the single-workgroup kernel)"""
    from amq_amd import ops
    dev = _dev()
    nh, nkv, max_seq = 4, 2, 1024
    kc = torch.zeros(1, nkv, max_seq, 128, dtype=torch.float16, device=dev)
    vc = torch.zeros_like(kc)
    q = torch.randn(1, nh * 128).half().to(dev); k = torch.randn(1, nkv * 128).half().to(dev); v = torch.randn(1, nkv * 128).half().to(dev)
    out = torch.full((1, nh * 128), 7.0, dtype=torch.float16, device=dev)
    cur, pos_state, err = ops.new_step_state(dev)
    pos_state.fill_(max_seq)
    ops.attn_decode(q, k, v, kc, vc, out, pos_state, nh, nkv, cur=cur, n_splits=3)
    torch.cuda.synchronize()
    assert int(err.item()) == 1 and bool((out == 7.0).all()) and float(kc.abs().sum()) == 0.0
    assert all(int(t.abs().sum().item()) == 0 for t in ops._ATTN_TICKETS._cur.values())


@pytest.mark.parametrize("pos", [0, 1, 31, 127, 128, 383, 384, 385, 700, 1023])
def test_attn_decode_long_context(pos):
    """one step at a given position over a pre-filled cache: covers the speculative prefetch (first 128 keys), the
    register-prefetch limit (384 keys), the remainder loop beyond it, and the last row of the cache"""
    from amq_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(100 + pos)
    nh, nkv, max_seq = 8, 4, 1024
    kc = torch.zeros(1, nkv, max_seq, 128, dtype=torch.float16, device=dev)
    vc = torch.zeros_like(kc)
    if pos:
        kc[0, :, :pos] = torch.randn(nkv, pos, 128, generator=g).half().to(dev)
        vc[0, :, :pos] = torch.randn(nkv, pos, 128, generator=g).half().to(dev)
    # poison the rows that must never contribute
    kc[0, :, pos + 1:] = 1000.0
    vc[0, :, pos + 1:] = 1000.0
    q = torch.randn(1, nh, 128, generator=g).half().to(dev)
    k = torch.randn(1, nkv, 128, generator=g).half().to(dev)
    v = torch.randn(1, nkv, 128, generator=g).half().to(dev)
    out = torch.zeros(1, nh * 128, dtype=torch.float16, device=dev)
    posd = torch.full((1,), pos, dtype=torch.int32, device=dev)
    tab = ops.rope_table(max_seq, 10000.0, dev)
    ops.attn_decode(q.reshape(1, -1), k.reshape(1, -1), v.reshape(1, -1), kc, vc, out, posd, nh, nkv, table=tab)
    # the graph-replay form (cos/sin row of the current position handed over by the previous step) gives the same bits
    kc2, vc2, out2 = kc.clone(), vc.clone(), torch.zeros_like(out)
    kc2[0, :, pos] = 1000.0
    cur, pos_state, _err = ops.new_step_state(dev)
    cur.copy_(tab.view(max_seq, 128)[pos]); pos_state.fill_(pos)
    ops.attn_decode(q.reshape(1, -1), k.reshape(1, -1), v.reshape(1, -1), kc2, vc2, out2, pos_state, nh, nkv, cur=cur)
    assert torch.equal(out2, out) and torch.equal(kc2[0, :, pos], kc[0, :, pos])
    kr = _rope_ref(k[0], pos)
    assert torch.equal(kc[0, :, pos], kr) and torch.equal(vc[0, :, pos], v[0])
    K = kc[0, :, :pos + 1].repeat_interleave(nh // nkv, 0)
    V = vc[0, :, :pos + 1].repeat_interleave(nh // nkv, 0)
    qr = _rope_ref(q[0], pos)
    w = torch.matmul(qr[:, None, :], K.transpose(1, 2)) * (128 ** -0.5)
    p_ = torch.softmax(w.float(), -1).half()
    ref = torch.matmul(p_, V)[:, 0, :].reshape(-1)
    assert torch.allclose(out[0].float(), ref.float(), rtol=1e-2, atol=3e-3), (out[0].float() - ref.float()).abs().max()


def _ref_model(m):
    """dense fp16 torch mirror of a QuantLlama (weights dequantized by the bit-exact dequantize kernel)"""
    from amq_amd import ops
    blocks = []
    for blk in m.blocks:
        d = {}
        for name in m.cfg["linear"]:
            l = blk[name]
            d[name] = ops.dequantize(l.qn, l.mn, l.bits, l.mode, l.N, l.K)
        d["ln1"], d["ln2"] = blk["ln1"], blk["ln2"]
        blocks.append(d)
    return blocks


def _ref_forward(m, blocks, ids):
    """HF-Llama-style eager forward of the whole prefix (fp16 linears, fp32 softmax); returns last-token logits"""
    import torch.nn.functional as F
    S = ids.numel()
    x = m.embed.index_select(0, ids)
    pos = torch.arange(S, device=ids.device)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32, device=ids.device) / 128.0))
    emb = torch.cat([pos.float()[:, None] * inv[None], pos.float()[:, None] * inv[None]], -1)
    cos, sin = emb.cos().half()[:, None], emb.sin().half()[:, None]

    def rope(t):
        return t * cos + torch.cat([-t[..., 64:], t[..., :64]], -1) * sin

    for d in blocks:
        h = _rms_ref(x, d["ln1"], 1e-5)
        q = rope(F.linear(h, d["self_attn.q_proj"]).view(S, m.nh, 128)).transpose(0, 1)
        k = rope(F.linear(h, d["self_attn.k_proj"]).view(S, m.nkv, 128)).transpose(0, 1).repeat_interleave(m.nh // m.nkv, 0)
        v = F.linear(h, d["self_attn.v_proj"]).view(S, m.nkv, 128).transpose(0, 1).repeat_interleave(m.nh // m.nkv, 0)
        w = torch.matmul(q, k.transpose(1, 2)) * (128 ** -0.5)
        w = w + torch.full((S, S), float("-inf"), device=ids.device).triu(1).half()
        a = torch.matmul(torch.softmax(w.float(), -1).half(), v).transpose(0, 1).reshape(S, m.H)
        x = x + F.linear(a, d["self_attn.o_proj"])
        h2 = _rms_ref(x, d["ln2"], 1e-5)
        x = x + F.linear(F.silu(F.linear(h2, d["mlp.gate_proj"])) * F.linear(h2, d["mlp.up_proj"]), d["mlp.down_proj"])
    return F.linear(_rms_ref(x[-1:], m.norm, 1e-5), m.lm_head)[0]


@pytest.mark.parametrize("gqa", [False, True])
def test_token_step_runner_matches_dense_reference(gqa):
    """prefill (MFMA GEMM path) + graph-replayed decode steps (GEMV path) reproduce a dense fp16
    HF-style forward with the same (bit-exactly dequantized) weights."""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2 if gqa else 4, 1, vocab=1024))
    rng = np.random.default_rng(0)
    al = {name: [int(b) for b in rng.choice([2, 3, 4], size=2)] for name in cfg["linear"]}
    m = QuantLlama(cfg, al, device="cuda:0", max_seq=64, seed=3)
    blocks = _ref_model(m)
    ids = torch.randint(0, 1024, (12,), generator=torch.Generator().manual_seed(5)).to(_dev())
    logits = m.prefill(ids).clone()
    ref = _ref_forward(m, blocks, ids)
    scale = ref.float().abs().max()
    assert (logits.float() - ref.float()).abs().max() <= 2e-2 * scale
    seq = ids.clone()
    for step in range(6):
        tok = m.token.clone()
        seq = torch.cat([seq, tok])
        m.decode_step(use_graph=(step >= 2))        # eager launches first, then hipGraph replays
        ref = _ref_forward(m, blocks, seq)
        assert (m.logits.float() - ref.float()).abs().max() <= 2e-2 * scale, step
        assert int(m.pos.item()) == seq.numel()


@pytest.mark.parametrize("gqa,S", [(False, 12), (True, 40), (False, 5), (True, 60)])
def test_fused_prefill_matches_framework_glue(gqa, S):
    """The fused many-row pass (RoPE + cache write, SiLU*up, residual epilogues) against the same pass with framework
    ops for the glue: same next token, logits and KV cache within fp16 rounding of the differently-ordered roundings."""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2 if gqa else 4, 1, vocab=1024))
    m = QuantLlama(cfg, None, device="cuda:0", max_seq=64, seed=2)
    ids = torch.randint(0, 1024, (S,), generator=torch.Generator().manual_seed(S)).to(_dev())
    a = m._prefill_unfused(ids).clone()
    tok_a = int(m.token.item())
    kc_a = [b["kc"].clone() for b in m.blocks]
    vc_a = [b["vc"].clone() for b in m.blocks]
    for b in m.blocks:
        b["kc"].zero_()
        b["vc"].zero_()
    m.reset()
    f = m.prefill(ids, use_graph=False).clone()
    scale = a.float().abs().max()
    assert (f.float() - a.float()).abs().max() <= 1e-2 * scale
    assert int(m.token.item()) == tok_a and int(m.pos.item()) == S
    for blk, ka, va in zip(m.blocks, kc_a, vc_a):
        assert torch.equal(blk["kc"][0, :, S:], ka[0, :, S:])                       # rows past the prompt untouched
        assert (blk["kc"].float() - ka.float()).abs().max() <= 1e-2 * ka.float().abs().max()
        assert (blk["vc"].float() - va.float()).abs().max() <= 1e-2 * va.float().abs().max()
    g = m.prefill(ids, use_graph=True).clone()                                      # captured replay = eager launches
    assert torch.equal(g, f)


def test_rope_cache_matches_decode_append():
    """amq_rope_cache_f16 writes exactly the cache rows the decode kernel appends for the same k / v, and rotates q
    like HF apply_rotary_pos_emb on fp16 tensors."""
    from amq_amd import ops, _lib
    dev = _dev()
    nh, nkv, S, max_seq = 4, 2, 9, 32
    g = torch.Generator().manual_seed(11)
    q = torch.randn(S, nh * 128, generator=g).half().to(dev)
    k = torch.randn(S, nkv * 128, generator=g).half().to(dev)
    v = torch.randn(S, nkv * 128, generator=g).half().to(dev)
    tab = ops.rope_table(max_seq, 10000.0, dev)
    kc = torch.zeros(nkv, max_seq, 128, dtype=torch.float16, device=dev)
    vc = torch.zeros_like(kc)
    q_rot = q.clone()
    ops.rope_cache(q_rot, k, v, kc, vc, tab, 3, nh, nkv)                             # rows 3 .. 11
    kc2 = torch.zeros(1, nkv, max_seq, 128, dtype=torch.float16, device=dev)
    vc2 = torch.zeros_like(kc2)
    out = torch.empty(1, nh * 128, dtype=torch.float16, device=dev)
    for s in range(S):
        pos = torch.tensor([3 + s], dtype=torch.int32, device=dev)
        ops.attn_decode(q[s:s + 1], k[s:s + 1], v[s:s + 1], kc2, vc2, out, pos, nh, nkv, 10000.0, table=tab)
    assert torch.equal(kc, kc2[0]) and torch.equal(vc, vc2[0])
    assert torch.equal(kc[:, :3], torch.zeros_like(kc[:, :3])) and torch.equal(kc[:, 12:], torch.zeros_like(kc[:, 12:]))
    ref = torch.stack([_rope_ref(q[s].view(nh, 128), 3 + s) for s in range(S)]).reshape(S, nh * 128)
    assert torch.equal(q_rot, ref)                                                   # HF apply_rotary_pos_emb in fp16
    with pytest.raises(_lib.AmqError):
        ops.rope_cache(q_rot, k, v, kc, vc, tab, max_seq - 2, nh, nkv)               # rows would run past the cache


def test_silu_mul():
    from amq_amd import ops, _lib
    g = torch.randn(7, 1384, generator=torch.Generator().manual_seed(1)).mul(3).half().to(_dev())
    u = torch.randn(7, 1384, generator=torch.Generator().manual_seed(2)).half().to(_dev())
    y = ops.silu_mul(g, u)
    ref = torch.nn.functional.silu(g) * u
    assert torch.allclose(y.float(), ref.float(), rtol=2e-3, atol=1e-4)
    with pytest.raises(_lib.AmqError):
        ops.silu_mul(g[:, :1383].contiguous(), u[:, :1383].contiguous())


def test_generate_is_deterministic_and_graph_equals_eager():
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 4, 1, vocab=1024))
    m = QuantLlama(cfg, None, device="cuda:0", max_seq=64, seed=1)
    ids = torch.arange(10, device=_dev())
    a = m.generate(ids, 12, use_graph=True).clone()
    m.reset()
    b = m.generate(ids, 12, use_graph=False).clone()
    assert torch.equal(a, b)


def test_speed_harness_modes_and_schema(tmp_path, monkeypatch):
    """benchmark_speed / the CLI produce the reference's result schema ({row: {mode: {'B.S.G': v}}, 'args'})"""
    from amq_amd import arch, speed_benchmark
    arch.MODEL_CONFIGS["tiny-512"] = dict(arch._cfg(2, 512, 1024, 4, 4, 2 * (4 * 512 * 512 + 3 * 512 * 1024), vocab=1024))
    try:
        a, usage = arch.synthesize_arch(arch.MODEL_CONFIGS["tiny-512"], 3.0, seed=1, tol=0.3)
        stats = tmp_path / "iter_0.stats"
        arch.write_stats(str(stats), a, usage)
        monkeypatch.chdir(tmp_path)
        res = speed_benchmark.main(["--model_name", "tiny-512", "--tps", "--gemv", "--gemm", "--ttft", "--memory", "--peak_memory",
                                    "--seq_length", "16", "--gen_length", "8", "--target_bits", str(usage),
                                    "--arch_path", str(stats), "--file_name", "out.json"])
    finally:
        arch.MODEL_CONFIGS.pop("tiny-512", None)
    row = f"{usage}bit"
    assert set(res) == {"fp16", row, "args"}
    for r in ("fp16", row):
        for mode in ("tps", "gemv", "gemm", "ttft"):
            assert list(res[r][mode]) == ["1.16.8"] and res[r][mode]["1.16.8"] > 0
        assert res[r]["memory"] > 0 and "peak_memory" in res[r]
    assert res[row]["memory"] < res["fp16"]["memory"]
    import json, os
    assert json.load(open(tmp_path / "benchmark" / "outputs" / "out.json"))["args"]["gen_length"] == 8


def test_reference_checkpoints_mixed_arch_logits():
    """tests/golden/ckpt: HQQ checkpoints written by the reference's own quantize_model/save_quantized (2/3/4 bit) and
    the logits its CPU forward gives for a mixed arch assembled like amq_speed_benchmark.py:231-251.  The loader +
    prefill (MFMA GEMM) + graph decode (GEMV) must reproduce them."""
    import json, os
    from amq_amd.checkpoint import load_mixed
    root = os.path.join(os.path.dirname(__file__), "golden", "ckpt")
    exp = np.load(os.path.join(root, "expected.npz"))
    arch_linear = json.loads(str(exp["arch"]))
    m = load_mixed({b: os.path.join(root, f"{b}bit") for b in (2, 3, 4)}, arch_linear, max_seq=64)
    assert (m.H, m.I, m.nb, m.vocab) == (256, 512, 2, 256)
    ids = torch.from_numpy(exp["ids"]).to(_dev())
    ref = exp["logits"]                                            # [5, vocab] fp32
    scale = np.abs(ref).max()
    lg = m.prefill(ids).float().cpu().numpy()
    assert np.abs(lg - ref[0]).max() <= 2e-2 * scale
    toks = [int(m.token.item())]
    for step in range(1, 5):
        # feed the REFERENCE's greedy token so both see the same prefix even if an argmax tie flips
        m.set_token(int(exp["tokens"][step - 1]))
        m.decode_step(use_graph=(step >= 2))
        lg = m.logits.float().cpu().numpy()
        assert np.abs(lg - ref[step]).max() <= 2e-2 * scale, step
        toks.append(int(m.token.item()))
    assert toks == [int(t) for t in exp["tokens"]]


def test_speed_benchmark_cli_with_reference_checkpoints(tmp_path, monkeypatch):
    """--save_path: the quantized row is assembled from HQQ checkpoint directories named like the reference's
    ({save_path}/{model}_{n}bit_128gs_1axis); here they are the golden checkpoints the real reference wrote"""
    import json, os, shutil
    from amq_amd import speed_benchmark
    root = os.path.join(os.path.dirname(__file__), "golden", "ckpt")
    save = tmp_path / "hqq"
    for b in (2, 3, 4):
        shutil.copytree(os.path.join(root, f"{b}bit"), save / f"tiny-llama_{b}bit_128gs_1axis")
    exp = np.load(os.path.join(root, "expected.npz"))
    arch_linear = json.loads(str(exp["arch"]))
    stats = tmp_path / "iter_0.stats"
    with open(stats, "w") as f:
        json.dump({"archive": [[{"linear": arch_linear}, 0.0, 3.0]], "candidates": []}, f)
    monkeypatch.chdir(tmp_path)
    res = speed_benchmark.main(["--model_name", "tiny-llama", "--save_path", str(save), "--gemv", "--ttft", "--memory", "--skip_fp16",
                                "--seq_length", "8", "--gen_length", "4", "--target_bits", "3.0", "--arch_path", str(stats)])
    row = "3.0bit"
    assert res[row]["gemv"]["1.8.4"] > 0 and res[row]["ttft"]["1.8.4"] > 0 and res[row]["memory"] > 0
    # a bit-width the arch needs but the directory lacks -> loud failure
    shutil.rmtree(save / "tiny-llama_2bit_128gs_1axis")
    with pytest.raises(FileNotFoundError):
        speed_benchmark.main(["--model_name", "tiny-llama", "--save_path", str(save), "--gemv", "--skip_fp16",
                              "--seq_length", "8", "--gen_length", "4", "--target_bits", "3.0", "--arch_path", str(stats)])


@pytest.mark.parametrize("group", [64, 32])
def test_runner_over_finer_group_layers(group):
    """the hipGraph runner over HQQ layers quantized with groups of 64 / 32: prompt pass (9 rows: GEMV kernel; 40 rows: dequantize-once + the
    fp16 GEMM per linear, not the fragment-ordered few-row kernels) and decode steps (the same five launches per block) against the fp16 runner
    on the oracle-exact dequantized weights; graph and eager forms agree bit for bit"""
    from amq_amd import arch, ops
    from amq_amd.hqq_format import random_hqq
    from amq_amd.llama import QuantLlama, DenseLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2, 1, vocab=1024))
    bits = dict(zip(cfg["linear"], [4, 2, 3, 3, 2, 4, 3]))
    al = {name: [b, b] for name, b in bits.items()}
    layers = {(blk, name): random_hqq(*cfg["linear_shape"][name], bits[name], seed=17 * blk + i, group=group)
              for blk in range(2) for i, name in enumerate(cfg["linear"])}
    m = QuantLlama(cfg, al, device="cuda:0", max_seq=64, hqq_layers=layers, seed=3)
    assert m.fine and m.engine is None
    d = DenseLlama(cfg, device="cuda:0", max_seq=64, seed=3)
    for b in range(2):
        for name in cfg["linear"]:
            l = m.blocks[b][name]
            d.blocks[b][name] = ops.dequantize(l.qn, l.mn, l.bits, l.mode, l.N, l.K)
        d.blocks[b]["ln1"], d.blocks[b]["ln2"] = m.blocks[b]["ln1"], m.blocks[b]["ln2"]
    d.embed, d.lm_head, d.norm = m.embed, m.lm_head, m.norm
    for S in (9, 40):
        ids = torch.randint(0, 1024, (S,), generator=torch.Generator().manual_seed(S)).to(_dev())
        m.reset(); d.reset()
        lg = m.prefill(ids, use_graph=False).float().clone()
        ld = d.prefill(ids, use_graph=False).float().clone()
        assert torch.isfinite(lg).all() and (lg - ld).abs().max() <= 1e-2 * ld.abs().max()
        m.decode_step(use_graph=False); d.decode_step(use_graph=False)          # one decode step each (same next token unless the argmax is a near-tie)
        if int(m.token.item()) == int(d.token.item()):
            m.decode_step(use_graph=False); d.decode_step(use_graph=False)
            assert (m.logits.float() - d.logits.float()).abs().max() <= 1e-2 * d.logits.float().abs().max()
        m.reset()
        lg2 = m.prefill(ids, use_graph=True).float().clone()
        assert torch.equal(lg2, lg)
        toks = []
        for _ in range(4):
            m.decode_step(); toks.append(int(m.token.item()))
        m.reset(); m.prefill(ids, use_graph=False)
        toks2 = []
        for _ in range(4):
            m.decode_step(use_graph=False); toks2.append(int(m.token.item()))
        assert toks == toks2


@pytest.mark.parametrize("gqa", [False, True])
def test_prefill_batch_matches_single_sequence_passes(gqa):
    """the batched prompt pass (GeMM mode at batch_size > 1) gives each sequence the logits of its own single-sequence pass"""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama, DenseLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2 if gqa else 4, 1, vocab=1024))
    m = QuantLlama(cfg, None, device="cuda:0", max_seq=64, seed=4)
    ids = torch.randint(0, 1024, (3, 40), generator=torch.Generator().manual_seed(9)).to(_dev())
    got = m.prefill_batch(ids)
    assert got.shape == (3, 1024)
    for b in range(3):
        m.reset()
        ref = m.prefill(ids[b], use_graph=False).float()
        assert (got[b].float() - ref).abs().max() <= 1e-2 * ref.abs().max()
    d = DenseLlama(cfg, device="cuda:0", max_seq=64, seed=4)
    gd = d.prefill_batch(ids)
    d.reset()
    rd = d.prefill(ids[1], use_graph=False).float()
    assert (gd[1].float() - rd).abs().max() <= 1e-2 * rd.abs().max()


@pytest.mark.parametrize("gqa,split", [(False, (40, 24)), (True, (7, 90)), (False, (70, 3))])
def test_prefill_in_chunks_matches_one_pass(gqa, split):
    """prefill(ids[a:], start_pos=a) behind prefill(ids[:a]) (the reference's patched forward takes the same start_pos,
    ftllama_modeling.py:76-104): same cache rows and last-token logits as one pass over the whole prompt (different row counts
    pick different GEMM kernels: fp16-rounding distance), the decode steps that follow agree, graph and eager forms agree."""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama, DenseLlama
    a, b = split
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2 if gqa else 4, 1, vocab=1024))
    ids = torch.randint(0, 1024, (a + b,), generator=torch.Generator().manual_seed(a)).to(_dev())
    for cls in (QuantLlama, DenseLlama):
        m = cls(cfg, None, device="cuda:0", max_seq=a + b + 8, seed=4) if cls is QuantLlama else cls(cfg, device="cuda:0", max_seq=a + b + 8, seed=4)
        whole = m.prefill(ids, use_graph=False).float().clone()
        kc_whole = [blk["kc"].clone() for blk in m.blocks]
        m.decode_step(use_graph=False)
        next_whole = m.logits.float().clone()
        m.reset()
        m.prefill(ids[:a], use_graph=False)
        got = m.prefill(ids[a:], use_graph=False, start_pos=a).float().clone()
        assert m.host_pos == a + b and int(m.pos.item()) == a + b
        scale = whole.abs().max()
        assert (got - whole).abs().max() <= 1e-2 * scale
        for blk, ref in zip(m.blocks, kc_whole):
            d = (blk["kc"][:, :, :a + b].float() - ref[:, :, :a + b].float()).abs().max()
            assert d <= 2e-2 * ref.float().abs().max()
        m.decode_step(use_graph=False)
        assert (m.logits.float() - next_whole).abs().max() <= 1e-2 * scale
        # the graph form of the second chunk (keyed by length and start position) gives the eager form's bits
        m.reset()
        m.prefill(ids[:a], use_graph=False)
        eager = m.prefill(ids[a:], use_graph=False, start_pos=a).clone()
        m.reset()
        m.prefill(ids[:a])
        graph = m.prefill(ids[a:], start_pos=a).clone()
        assert torch.equal(graph, eager) and m.host_pos == a + b
    with pytest.raises(ValueError):
        m.prefill(ids, start_pos=20)                       # does not fit the cache


@pytest.mark.parametrize("B,gqa", [(2, False), (4, True), (8, False)])
def test_batched_decode_matches_single_sequence_runs(B, gqa):
    """QuantLlama(batch=B): B sequences decoded together (one step = the same launches with B rows, weights streamed once) give
    each sequence what a batch-1 runner gives it alone: same greedy tokens, logits to fp16 rounding (M = B rows go through the
    general x staging of the GEMV kernel instead of the one-row register path); graph replay == eager; dense runner too."""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama, DenseLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2 if gqa else 4, 1, vocab=1024))
    ids = torch.randint(0, 1024, (B, 24), generator=torch.Generator().manual_seed(B)).to(_dev())
    steps = 6
    for dense in (False, True):
        mk = (lambda b: DenseLlama(cfg, device="cuda:0", max_seq=48, seed=4, batch=b)) if dense else \
             (lambda b: QuantLlama(cfg, None, device="cuda:0", max_seq=48, seed=4, batch=b))
        mb = mk(B)
        out_b = mb.generate(ids, steps, use_graph=False).clone()                 # [B, steps]
        logits_b = mb.logits.float().clone()
        assert out_b.shape == (B, steps) and int(mb.pos.item()) == 24 + steps - 1
        m1 = mk(1)
        for b in range(B):
            m1.reset()
            out_1 = m1.generate(ids[b], steps, use_graph=False)
            ref = m1.logits.float()
            assert (logits_b[b] - ref).abs().max() <= 1e-2 * ref.abs().max()
            agree = (out_1 == out_b[b]).float().mean().item()
            assert agree >= 0.8, (b, out_1.tolist(), out_b[b].tolist())         # (a near-tie may flip a greedy choice)
        mb.reset()
        out_g = mb.generate(ids, steps, use_graph=True)
        assert torch.equal(out_g, out_b) and torch.equal(mb.logits.float(), logits_b)
        mb.check()
    with pytest.raises(ValueError):
        mb.prefill(ids[:1])                                # wrong number of prompts


@pytest.mark.parametrize("B", [3, 5, 7, 8])
def test_batched_decode_at_7b_width(B):
    """the runner's per-row-count routes at Llama-2-7B layer widths (hidden 4096, intermediate 11008; one block): SiLU*mul unfused from 2 rows, RMSNorms
    from 5, down_proj's x in two K phases at 7 - 8 rows -- every sequence against its own batch-1 run, graph replay == eager"""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(1, 4096, 11008, 32, 32, 1, vocab=1024))
    ids = torch.randint(0, 1024, (B, 16), generator=torch.Generator().manual_seed(B)).to(_dev())
    steps = 4
    mb = QuantLlama(cfg, None, device="cuda:0", max_seq=32, seed=4, batch=B)
    out_b = mb.generate(ids, steps, use_graph=False).clone()
    logits_b = mb.logits.float().clone()
    m1 = QuantLlama(cfg, None, device="cuda:0", max_seq=32, seed=4)
    for b in range(B):
        m1.reset()
        m1.generate(ids[b], steps, use_graph=False)
        ref = m1.logits.float()
        assert (logits_b[b] - ref).abs().max() <= 1e-2 * ref.abs().max(), b
    mb.reset()
    out_g = mb.generate(ids, steps, use_graph=True)
    assert torch.equal(out_g, out_b) and torch.equal(mb.logits.float(), logits_b)
    mb.check()


def test_batched_decode_long_cache_and_chunked_prompt():
    """batch 3 over a cache long enough for the split attention kernel (several workgroups per head and sequence), the prompt fed in
    two chunks: every sequence agrees with its own batch-1 run (whole prompt, single-workgroup attention forced)"""
    from amq_amd import arch, ops
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2, 1, vocab=1024))
    B, S, max_seq, steps = 3, 300, 640, 5
    assert ops.attn_decode_splits(max_seq) > 1
    ids = torch.randint(0, 1024, (B, S), generator=torch.Generator().manual_seed(5)).to(_dev())
    mb = QuantLlama(cfg, None, device="cuda:0", max_seq=max_seq, seed=4, batch=B)
    mb.prefill(ids[:, :200], use_graph=False)
    mb.prefill(ids[:, 200:], use_graph=False, start_pos=200)
    toks = [mb.token.clone()]
    for _ in range(steps):
        mb.decode_step()
        toks.append(mb.token.clone())
    got = torch.stack(toks, 1)                                   # [B, steps + 1]
    logits_b = mb.logits.float().clone()
    mb.check()
    m1 = QuantLlama(cfg, None, device="cuda:0", max_seq=max_seq, seed=4)
    for b in range(B):
        m1.reset()
        ref_t = m1.generate(ids[b], steps + 1, use_graph=False)
        ref = m1.logits.float()
        assert (logits_b[b] - ref).abs().max() <= 1e-2 * ref.abs().max()
        assert (ref_t == got[b]).float().mean().item() >= 0.8


def test_batched_lm_head_and_tail_kernels():
    """amq_gemv_f16w_rows (lm_head for 2 .. 8 rows, W streamed once) row by row == the one-row kernel; amq_decode_tail_batch_f16:
    per-row first-maximum argmax + embedding gather, the shared position advanced once"""
    from amq_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    n, k = 1000, 1024
    w = (torch.randn(n, k, generator=g) * 0.03).half().to(dev)
    gamma = (1 + 0.1 * torch.randn(k, generator=g)).half().to(dev)
    for M in (2, 3, 8):
        x = torch.randn(M, k, generator=g).half().to(dev)
        for gm in (None, gamma):
            y = ops.gemv_f16w(x, w, gamma=gm, eps=1e-5)
            assert y.shape == (M, n)
            for m in range(M):
                assert torch.equal(y[m], ops.gemv_f16w(x[m].contiguous(), w, gamma=gm, eps=1e-5))
    vocab, hidden, B = 1024, 256, 5
    embed = torch.randn(vocab, hidden, generator=g).half().to(dev)
    logits = torch.randn(B, vocab, generator=g).half().to(dev)
    logits[2, 100] = 9.0; logits[2, 700] = 9.0                      # tie: the first one wins
    token = torch.zeros(B, dtype=torch.int64, device=dev)
    tab = ops.rope_table(50, 10000.0, dev)
    cur, pos, _err = ops.new_step_state(dev)
    pos.fill_(41)
    x = torch.zeros(B, hidden, dtype=torch.float16, device=dev)
    ops.decode_tail(logits, embed, token, pos, x, table=tab, cur=cur)
    assert torch.equal(token.cpu(), torch.argmax(logits.float().cpu(), dim=1)) and int(token[2]) == 100
    assert int(pos.item()) == 42 and torch.equal(cur, tab.view(50, 128)[42])
    assert torch.equal(x, embed[token])


def test_speed_harness_gemm_mode_batched():
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    from amq_amd.speed import benchmark_speed
    cfg = dict(arch._cfg(2, 512, 1024, 4, 4, 1, vocab=1024))
    m = QuantLlama(cfg, None, device="cuda:0", max_seq=64, seed=4)
    r = benchmark_speed(m, iteration=2, sizes=(4, 32, 8), mode="GeMM", get_peak_memory=True)
    assert r["gemm"]["4.32.8"] > 0 and r["peak_memory"]["4.32.8"] > 0
    with pytest.raises(NotImplementedError):
        benchmark_speed(m, iteration=1, sizes=(4, 32, 8), mode="TPS", get_peak_memory=False)


def test_decode_past_the_cache_is_refused_on_host_and_skipped_on_device():
    """A decode step whose position is outside the KV cache must not touch the cache or LDS: decode_step() refuses on the
    host; a raw graph replay (bypassing the host mirror) is a no-op in the attention kernel, which raises the sticky
    error word of the step state; the position saturates at max_seq."""
    from amq_amd import arch, ops
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(2, 512, 1024, 4, 2, 1, vocab=1024))
    m = QuantLlama(cfg, None, device="cuda:0", max_seq=16, seed=4)
    ids = torch.randint(0, 1024, (13,), generator=torch.Generator().manual_seed(1)).to(_dev())
    m.prefill(ids)
    guard = [torch.full((4096,), 7.0, dtype=torch.float16, device=_dev()) for _ in range(2)]   # neighbours in the allocator
    for _ in range(3):                       # positions 13, 14, 15: the last rows of the cache
        m.decode_step()
    m.check()
    assert int(m.pos.item()) == 16 and m.host_pos == 16
    with pytest.raises(ValueError, match="does not fit the KV cache"):
        m.decode_step()
    kc = [b["kc"].clone() for b in m.blocks]
    vc = [b["vc"].clone() for b in m.blocks]
    for _ in range(3):                       # what a caller replaying the captured graph blindly would do
        m.graph.replay()
    torch.cuda.synchronize()
    assert int(m.pos.item()) == 16           # saturated by the step's tail kernel
    assert int(m.step_err.item()) == 1       # raised by the attention kernel
    with pytest.raises(Exception, match="outside the KV cache"):
        m.check()
    for b, k0, v0 in zip(m.blocks, kc, vc):
        assert torch.equal(b["kc"], k0) and torch.equal(b["vc"], v0)
    assert all(torch.all(g == 7.0) for g in guard)
    # the plain entry point (position from a device int) skips as well
    nh, nkv, max_seq = 4, 2, 16
    q = torch.randn(1, nh * 128, device=_dev()).half()
    k = torch.randn(1, nkv * 128, device=_dev()).half()
    kcache = torch.zeros(1, nkv, max_seq, 128, device=_dev(), dtype=torch.float16)
    vcache = torch.zeros_like(kcache)
    out = torch.full((1, nh * 128), 3.0, device=_dev(), dtype=torch.float16)
    for bad in (16, 1000, -1):
        ops.attn_decode(q, k, k.clone(), kcache, vcache, out, torch.tensor([bad], dtype=torch.int32, device=_dev()), nh, nkv)
    torch.cuda.synchronize()
    assert torch.all(out == 3.0) and torch.all(kcache == 0) and torch.all(vcache == 0)
    with pytest.raises(ValueError):
        ops.attn_decode(q, k, k.clone(), kcache, vcache, out, 16, nh, nkv)


def test_prefill_graph_survives_scratch_growth():
    """a per-length prefill graph has the split-K workspace / dequant scratch pointers baked in; a later, larger
    request must not free what an older graph still writes.  prefill(64), prefill(300) (grows the workspace), empty_cache,
    then the replayed 64-token graph must still equal the eager pass."""
    from amq_amd import arch, ops
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(2, 1024, 2816, 8, 8, 1, vocab=1024))
    m = QuantLlama(cfg, None, device="cuda:0", max_seq=512, seed=6)
    g = torch.Generator().manual_seed(3)
    ids64 = torch.randint(0, 1024, (64,), generator=g).to(_dev())
    ids300 = torch.randint(0, 1024, (300,), generator=g).to(_dev())
    eager64 = m.prefill(ids64, use_graph=False).clone()
    a = m.prefill(ids64).clone()
    assert torch.equal(a, eager64)
    ws_before = [t.data_ptr() for t in ops._SPLITK_WS._cur.values()]
    m.prefill(ids300)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 20,), float("nan"), device=_dev()) for _ in range(8)]     # reuse whatever was freed
    b = m.prefill(ids64).clone()
    torch.cuda.synchronize()
    assert torch.equal(b, eager64)
    kept = {t.data_ptr() for t in ops._SPLITK_WS._keep} | {t.data_ptr() for t in ops._SPLITK_WS._cur.values()}
    assert all(p in kept for p in ws_before)                   # nothing a graph may reference was released
    del junk


@pytest.mark.parametrize("S,nh,nkv,pos0", [(5, 4, 4, 0), (64, 4, 2, 0), (65, 8, 2, 0), (200, 32, 8, 0), (33, 4, 2, 70), (130, 4, 1, 0)])
def test_attn_prefill_xfrag_output(S, nh, nkv, pos0):
    """amq_attn_prefill_xfrag_f16: the attention result written straight in fragment order == amq_xfrag_f16 of the row-major
    result, bit for bit, including the zero rows that pad the last 64-row group"""
    from amq_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(S + nh)
    H = nh * 128
    q = torch.randn(S, H, generator=g).half().to(dev)
    T = pos0 + S
    kc = torch.zeros(1, nkv, T + 5, 128, dtype=torch.float16, device=dev); vc = torch.zeros_like(kc)
    kc[:, :, :T] = torch.randn(1, nkv, T, 128, generator=g).half().to(dev)
    vc[:, :, :T] = torch.randn(1, nkv, T, 128, generator=g).half().to(dev)
    rows = ops.attn_prefill(q, kc, vc, torch.empty_like(q), S, nh, nkv, pos0=pos0, kv_cache=True)
    want = ops.xfrag(rows, S, H)
    got = ops.attn_prefill(q, kc, vc, torch.full_like(want, float("nan")), S, nh, nkv, pos0=pos0, kv_cache=True, out_xfrag=True)
    assert torch.equal(got, want)


@pytest.mark.parametrize("S,nh,nkv,batch,pos0", [(5, 4, 4, 1, 0), (64, 4, 2, 1, 0), (65, 8, 2, 2, 0), (200, 4, 4, 1, 0),
                                                  (130, 4, 1, 3, 0), (33, 4, 2, 1, 70), (257, 2, 2, 1, 0),
                                                  (300, 32, 8, 3, 0), (200, 32, 32, 4, 100),      # (many workgroups: several per CU)
                                                  (300, 32, 8, 24, 0), (200, 32, 32, 36, 100)])    # (>= 2048 128-row workgroups: QB = 2 kernel)
def test_attn_prefill_matches_eager_formula(S, nh, nkv, batch, pos0):
    """amq_attn_prefill_f16 (flash-style MFMA kernel: transposed products, hardware transpose read of V, online softmax)
    against the eager HF formula in fp32: softmax(mask(q k^T / sqrt(d))) v, GQA, ragged prompt lengths, several sequences,
    and a prompt chunk appended behind pos0 cached keys (cache layout) -- element-wise to fp16 accuracy."""
    from amq_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(S * 7 + nh + pos0)
    H, KV = nh * 128, nkv * 128
    q = torch.randn(batch * S, H, generator=g).half().to(dev)
    T = pos0 + S
    if pos0:
        max_seq = T + 9
        kc = torch.zeros(batch, nkv, max_seq, 128, dtype=torch.float16, device=dev)
        vc = torch.zeros_like(kc)
        kc[:, :, :T] = torch.randn(batch, nkv, T, 128, generator=g).half().to(dev)
        vc[:, :, :T] = torch.randn(batch, nkv, T, 128, generator=g).half().to(dev)
        kc[:, :, T:] = float("nan"); vc[:, :, T:] = float("nan")    # rows past the context must never be read into the result
        out = ops.attn_prefill(q, kc, vc, torch.empty_like(q), S, nh, nkv, batch=batch, pos0=pos0, kv_cache=True)
        kf = kc[:, :, :T].float()                                   # [B, nkv, T, 128]
        vf = vc[:, :, :T].float()
    else:
        k = torch.randn(batch * S, KV, generator=g).half().to(dev)
        v = torch.randn(batch * S, KV, generator=g).half().to(dev)
        out = ops.attn_prefill(q, k, v, torch.empty_like(q), S, nh, nkv, batch=batch)
        kf = k.view(batch, S, nkv, 128).transpose(1, 2).float()
        vf = v.view(batch, S, nkv, 128).transpose(1, 2).float()
    qf = q.view(batch, S, nh, 128).transpose(1, 2).float()          # [B, nh, S, 128]
    kf = kf.repeat_interleave(nh // nkv, dim=1)
    vf = vf.repeat_interleave(nh // nkv, dim=1)
    sc = qf @ kf.transpose(-1, -2) / (128 ** 0.5)
    mask = torch.arange(T, device=dev)[None, :] > (pos0 + torch.arange(S, device=dev))[:, None]
    sc = sc.masked_fill(mask[None, None], float("-inf"))
    ref = (torch.softmax(sc, dim=-1) @ vf).transpose(1, 2).reshape(batch * S, H)
    err = (out.float() - ref).abs()
    assert torch.isfinite(out.float()).all()
    assert err.max() <= 4e-3 * ref.abs().max() + 1e-3, err.max()    # fp16 probabilities + fp16 output rounding
    # deterministic
    again = ops.attn_prefill(q, kc if pos0 else k, vc if pos0 else v, torch.empty_like(q), S, nh, nkv, batch=batch, pos0=pos0,
                             kv_cache=bool(pos0))
    assert torch.equal(again, out)


def test_set_token_kernel_equals_the_framework_ops():
    """amq_set_token_f16 (next input token + embedding row + the position's cos/sin row in one launch: what hf_fast's forward pays per fed token) leaves the
    step state the five framework ops leave; one id broadcast over a batch; ids outside the vocabulary are clamped, not gathered out of bounds"""
    from amq_amd import arch
    from amq_amd.llama import QuantLlama
    cfg = dict(arch._cfg(1, 512, 1024, 4, 4, 1, vocab=1024))
    for B in (1, 3):
        m = QuantLlama(cfg, None, device="cuda:0", max_seq=64, seed=2, batch=B)
        for pos, toks in ((0, [5] * B), (17, list(range(7, 7 + B))), (63, [1023] * B)):
            m.set_pos(pos)
            for t in (toks[0],) if B == 1 else ():
                m.set_token(int(t))                                   # the framework-op path (a Python int)
                want = (m.token.clone(), m.x.clone(), m.rope_cur.clone())
                m.x.zero_(); m.token.zero_(); m.rope_cur.zero_()
                m.set_token(torch.tensor([t], dtype=torch.int64, device="cuda:0"))      # the kernel
                assert torch.equal(m.token, want[0]) and torch.equal(m.x, want[1]) and torch.equal(m.rope_cur, want[2])
            tt = torch.tensor(toks, dtype=torch.int64, device="cuda:0")
            m.set_token(tt)
            assert torch.equal(m.token, tt) and torch.equal(m.x, m.embed[tt]) and torch.equal(m.rope_cur.view(-1), m.rope_tab.view(64, 128)[pos])
            m.set_token(tt[:1])                                       # one id for every sequence
            assert torch.equal(m.token, tt[:1].expand(B)) and torch.equal(m.x, m.embed[tt[:1]].expand(B, -1))
        m.set_token(torch.tensor([5000] * B, dtype=torch.int64, device="cuda:0"))
        assert int(m.token.max()) == 1023
