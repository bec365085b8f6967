#!/usr/bin/env python3
"""Random sweep of the decode attention entry points against the eager fp32 formula (GPU box; test infrastructure, not product): random (query heads,
kv heads, cache size, position, sequences, splits) through ops.attn_decode (and, for half as many cases, ops.attn_prefill: the prompt kernel) -- the single-workgroup kernel, the per-head split kernel and the
grouped-query kernel (stage / tile boundaries, the first token, the last row of the cache, one .. many active chunks) -- at the tests' bound; the appended
row must be HF's rotation of the new key, nothing else in the caches may change, the tickets must be left zero, a second run must give the same bits.
usage: fuzz_attn.py [cases=200] [seed=0]   -- prints every failing case and exits non-zero if there was one."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0


def rope_ref(t, pos):
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32, device=t.device) / 128.0))
    fr = torch.tensor([float(pos)], device=t.device)[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().half(), emb.sin().half()
    return t * cos + torch.cat([-t[..., 64:], t[..., :64]], -1) * sin


for c in range(cases):
    G = rng.choice([1, 1, 2, 3, 4, 4, 5, 7, 8, 16, 20])
    nkv = rng.choice([1, 2, 3, 4, 8])
    nh = G * nkv
    if nh > 128:
        nkv = max(1, 128 // G); nh = G * nkv
    max_seq = rng.choice([200, 512, 513, 700, 1024, 1500, 2048, 3000, 4096, 8192, 12000])
    edge = rng.random()
    if edge < 0.15:
        pos = rng.choice([0, 1, 31, 32, 63, 64, 65, 127, 128, 129, 255, 256, 257])
    elif edge < 0.3:
        pos = max_seq - 1 - rng.choice([0, 1, 2, 63, 64, 65])
    elif edge < 0.45:
        pos = 128 * rng.randrange(1, max(2, max_seq // 128)) + rng.choice([-2, -1, 0, 1])
    else:
        pos = rng.randrange(0, max_seq)
    pos = max(0, min(max_seq - 1, pos))
    batch = rng.choice([1, 1, 1, 2, 3])
    n_splits = rng.choice([0, 0, 0, 1, 2, 3, 5, 8, 16, 33]) if max_seq > 512 else rng.choice([0, 1, 2])
    what = f"nh {nh} nkv {nkv} max_seq {max_seq} pos {pos} batch {batch} n_splits {n_splits}"
    try:
        g = torch.Generator().manual_seed(1000 + c)
        kc = torch.zeros(batch, nkv, max_seq, 128, dtype=torch.float16, device=dev)
        vc = torch.zeros_like(kc)
        kc[:, :, :pos] = torch.randn(batch, nkv, pos, 128, generator=g).half().to(dev)
        vc[:, :, :pos] = torch.randn(batch, nkv, pos, 128, generator=g).half().to(dev)
        kc[:, :, pos:] = float("nan"); vc[:, :, pos:] = float("nan")
        q = torch.randn(batch, nh * 128, generator=g).half().to(dev)
        k = torch.randn(batch, nkv * 128, generator=g).half().to(dev)
        v = torch.randn(batch, nkv * 128, generator=g).half().to(dev)
        tab = ops.rope_table(max_seq, 10000.0, dev)
        outs = []
        for rep in range(2):
            kc_, vc_ = kc.clone(), vc.clone()
            out = torch.zeros(batch, nh * 128, dtype=torch.float16, device=dev)
            if rep == 0:
                posd = torch.full((1,), pos, dtype=torch.int32, device=dev)
                ops.attn_decode(q, k, v, kc_, vc_, out, posd, nh, nkv, table=tab, n_splits=n_splits)
            else:
                cur, pos_state, err = ops.new_step_state(dev)
                cur.copy_(tab.view(max_seq, 128)[pos]); pos_state.fill_(pos)
                ops.attn_decode(q, k, v, kc_, vc_, out, pos_state, nh, nkv, cur=cur, n_splits=n_splits)
                assert int(err.item()) == 0, "error word"
            outs.append(out)
            for b_ in range(batch):
                assert torch.equal(kc_[b_, :, pos], rope_ref(k[b_].view(nkv, 128), pos)), "appended key"
                assert torch.equal(vc_[b_, :, pos], v[b_].view(nkv, 128)), "appended value"
            assert torch.equal(kc_[:, :, :pos], kc[:, :, :pos]) and torch.equal(vc_[:, :, :pos], vc[:, :, :pos]), "cached rows changed"
            assert torch.isnan(kc_[:, :, pos + 1:]).all() and torch.isnan(vc_[:, :, pos + 1:]).all(), "rows past the position written"
        assert torch.equal(outs[0], outs[1]), "table / step-state runs differ"
        assert all(int(t.abs().sum().item()) == 0 for t in ops._ATTN_TICKETS._cur.values()), "tickets"
        K = kc_[:, :, :pos + 1].repeat_interleave(G, 1).float()
        V = vc_[:, :, :pos + 1].repeat_interleave(G, 1).float()
        qr = torch.stack([rope_ref(q[b_].view(nh, 128), pos) for b_ in range(batch)]).float()
        w = torch.einsum("bhd,bhtd->bht", qr, K) * (128 ** -0.5)
        ref = torch.einsum("bht,bhtd->bhd", torch.softmax(w, -1), V).reshape(batch, -1)
        err_ = (outs[0].float() - ref).abs().max().item()
        bar = 4e-3 * ref.abs().max().item() + 1e-3
        assert torch.isfinite(outs[0].float()).all(), "non-finite output"
        assert err_ <= bar, f"error {err_:.3e} over the bar {bar:.3e}"
    except Exception as e:                                  # noqa: BLE001 (a sweep reports and goes on)
        fails += 1
        print("FAIL", what, "--", repr(e)[:200], flush=True)
    if c % 25 == 24:
        print(f"... {c + 1} cases, {fails} failures", flush=True)
print(f"decode: {cases} cases, {fails} failures", flush=True)
dec_fails = fails

# ---- the prompt kernel (amq_attn_prefill_f16): random (rows, heads, kv heads, sequences, cached prefix) in both K / V layouts and with the result in
# fragment order, against the eager causal formula
pcases = cases // 2
for c in range(pcases):
    G = rng.choice([1, 1, 2, 4, 5, 7, 8])
    nkv = rng.choice([1, 2, 3, 4, 8])
    nh = G * nkv
    S = rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 100, 127, 128, 129, 200, 256, 300, 511, 700, 1024])
    batch = rng.choice([1, 1, 2, 3]) if S * nh <= 4096 * 8 else 1
    pos0 = rng.choice([0, 0, 1, 40, 63, 64, 65, 300, 1000]) if rng.random() < 0.6 else 0
    xfrag = batch == 1 and rng.random() < 0.3
    what = f"prefill S {S} nh {nh} nkv {nkv} batch {batch} pos0 {pos0} xfrag {xfrag}"
    try:
        g = torch.Generator().manual_seed(5000 + c)
        H, KV = nh * 128, nkv * 128
        T = pos0 + S
        q = torch.randn(batch * S, H, generator=g).half().to(dev)
        use_cache = pos0 > 0 or rng.random() < 0.5
        if use_cache:
            max_seq = T + rng.choice([0, 1, 9, 100])
            kc = torch.full((batch, nkv, max_seq, 128), float("nan"), dtype=torch.float16, device=dev)
            vc = torch.full_like(kc, float("nan"))
            kc[:, :, :T] = torch.randn(batch, nkv, T, 128, generator=g).half().to(dev)
            vc[:, :, :T] = torch.randn(batch, nkv, T, 128, generator=g).half().to(dev)
            kk, vv = kc, vc
            kf, vf = kc[:, :, :T].float(), vc[:, :, :T].float()
        else:
            kk = torch.randn(batch * S, KV, generator=g).half().to(dev)
            vv = torch.randn(batch * S, KV, generator=g).half().to(dev)
            kf = kk.view(batch, S, nkv, 128).transpose(1, 2).float()
            vf = vv.view(batch, S, nkv, 128).transpose(1, 2).float()
        if xfrag:
            xf = ops.attn_prefill(q, kk, vv, None, S, nh, nkv, batch=1, pos0=pos0, kv_cache=use_cache, out_xfrag=True)
            out = ops.attn_prefill(q, kk, vv, torch.empty_like(q), S, nh, nkv, batch=1, pos0=pos0, kv_cache=use_cache)
            assert torch.equal(xf, ops.xfrag(out, S, H)), "fragment-ordered result differs from the row-major one"
        else:
            out = ops.attn_prefill(q, kk, vv, torch.empty_like(q), S, nh, nkv, batch=batch, pos0=pos0, kv_cache=use_cache)
        qf = q.view(batch, S, nh, 128).transpose(1, 2).float()
        kf = kf.repeat_interleave(G, dim=1); vf = vf.repeat_interleave(G, dim=1)
        sc = qf @ kf.transpose(-1, -2) / (128 ** 0.5)
        mask = torch.arange(T, device=dev)[None, :] > (pos0 + torch.arange(S, device=dev))[:, None]
        ref = (torch.softmax(sc.masked_fill(mask[None, None], float("-inf")), dim=-1) @ vf).transpose(1, 2).reshape(batch * S, H)
        assert torch.isfinite(out.float()).all(), "non-finite output"
        err_ = (out.float() - ref).abs().max().item()
        bar = 4e-3 * ref.abs().max().item() + 1e-3
        assert err_ <= bar, f"error {err_:.3e} over the bar {bar:.3e}"
        again = ops.attn_prefill(q, kk, vv, torch.empty_like(q), S, nh, nkv, batch=batch, pos0=pos0, kv_cache=use_cache)
        assert torch.equal(again, out), "not deterministic"
    except Exception as e:                                  # noqa: BLE001
        fails += 1
        print("FAIL", what, "--", repr(e)[:200], flush=True)
print(f"prefill: {pcases} cases, {fails - dec_fails} failures")
print(f"{cases + pcases} cases, {fails} failures")
sys.exit(1 if fails else 0)
