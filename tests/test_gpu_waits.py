"""GPU: the product library against its conservative-waits twin, bit for bit.

Every hand-counted wait of the product kernels (`AMQ_WAIT_VM`, amq_common.cuh: the LDS-DMA pipelines' `s_waitcnt vmcnt(N)`) is a full drain in
`libamq_hip_safe.so` (`make -C amq_amd/csrc safe`, -DAMQ_WAITS_CONSERVATIVE).  A count that is one transfer short shows -- when the transfer is late --
as a result that differs from the twin's.  The corpus covers every kernel family with a counted wait (GEMV rows staged by LDS-DMA incl. the
two-K-phase form, the few-row streams, the ring / wave-specialised / ping-pong GEMMs in fp16 and bf16, split attention at 600 - 4000 keys, the
prompt attention); each case runs 20 times on either library, once on a quiet GPU and once with a second stream streaming gigabytes through the
memory system to perturb the order in which transfers land.  The twin is test infrastructure: nothing in amq_amd/ loads it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPS = 20


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _layer(bits, n, k, seed, mode_fma=False):
    from amq_amd import ops
    from amq_amd.hqq_format import random_hqq
    h = random_hqq(n, k, bits, seed=seed).to(_dev())
    qn, mn = ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), bits, n, k)
    return qn, mn


def _corpus():
    """[(name, fn)]: fn() launches on the library `_lib.load()` currently hands out and returns the result tensor(s)"""
    from amq_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(2024)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(torch.float16).to(dev)
    cases = []

    # ---- GEMV: one row, rows by LDS-DMA (RS = 64: 2 - 4 rows, RS = 128: 5 - 8), two K phases (8 rows of K = 11008), generic staging (9, 16 rows)
    for bits, m, k, n in ((4, 1, 4096, 1024), (3, 2, 4096, 1024), (2, 3, 4096, 768), (4, 4, 4096, 1024), (3, 5, 4096, 1024), (3, 8, 4096, 1024),
                          (2, 8, 8192, 512), (4, 6, 11008, 512), (3, 7, 11008, 1024), (3, 8, 11008, 4096), (4, 9, 4096, 512), (2, 16, 2048, 512)):
        qn, mn = _layer(bits, n, k, seed=bits + m)
        x, up, gamma, res = rnd(m, k), rnd(m, k), (1.0 + 0.1 * torch.randn(k, generator=gen)).to(torch.float16).to(dev), rnd(m, n)
        norm_ok = m <= ops.gemv_max_rows(k, plain=True)

        def gv(pro, qn=qn, mn=mn, bits=bits, n=n, k=k, x=x, up=up, gamma=gamma, res=res, m=m):
            y = torch.empty(m, n, dtype=torch.float16, device=dev)
            kw = dict(gamma=gamma, eps=1e-5) if pro == ops.PRO_RMSNORM else dict(x2=up) if pro == ops.PRO_SILU_MUL else {}
            ops.gemv_grouped(x, [dict(qn=qn, mn=mn, bits=bits, mode=ops.MODE_HQQ, N=n, y=y, residual=res)], k, prologue=pro, **kw)
            return y
        cases.append((f"gemv {bits}b {m}x{k}->{n} plain", lambda gv=gv: gv(ops.PRO_NONE)))
        cases.append((f"gemv {bits}b {m}x{k}->{n} silu*mul", lambda gv=gv: gv(ops.PRO_SILU_MUL)))
        if norm_ok:
            cases.append((f"gemv {bits}b {m}x{k}->{n} rmsnorm", lambda gv=gv: gv(ops.PRO_RMSNORM)))
    # grouped q/k/v-like launch of three bit-widths at 8 rows (segments dealt by row-tile share)
    segs = [(b,) + _layer(b, nn, 4096, seed=40 + b) + (nn,) for b, nn in ((2, 1024), (3, 512), (4, 2048))]
    x8, g8 = rnd(8, 4096), (1.0 + 0.1 * torch.randn(4096, generator=gen)).to(torch.float16).to(dev)

    def grouped():
        ys = [torch.empty(8, nn, dtype=torch.float16, device=dev) for _, _, _, nn in segs]
        ops.gemv_grouped(x8, [dict(qn=q, mn=mt, bits=b, mode=ops.MODE_HQQ, N=nn, y=y) for (b, q, mt, nn), y in zip(segs, ys)], 4096,
                         prologue=ops.PRO_RMSNORM, gamma=g8, eps=1e-5)
        return torch.cat(ys, dim=1)
    cases.append(("gemv grouped 2/3/4 bit, 8 rows, rmsnorm", grouped))
    # the partial-sum RMSNorm of the 5 .. 8-row steps: producer (sums_out in the epilogue) and consumer (sums_in: rows by LDS-DMA, partials, transform)
    qo, mo = _layer(3, 4096, 4096, seed=55)
    a8, r8 = rnd(8, 4096), rnd(8, 4096)

    def sums_pair():
        y1 = torch.empty(8, 4096, dtype=torch.float16, device=dev)
        ss = torch.empty(8, 256, dtype=torch.float32, device=dev)
        ops.gemv_grouped_sums(a8, [dict(qn=qo, mn=mo, bits=3, mode=ops.MODE_HQQ, N=4096, y=y1, residual=r8)], 4096, sums_out=ss)
        ys = [torch.empty(8, nn, dtype=torch.float16, device=dev) for _, _, _, nn in segs]
        ops.gemv_grouped_sums(y1, [dict(qn=q, mn=mt, bits=b, mode=ops.MODE_HQQ, N=nn, y=y) for (b, q, mt, nn), y in zip(segs, ys)], 4096,
                              gamma=g8, eps=1e-5, sums_in=ss)
        return torch.cat([y1] + ys, dim=1)
    cases.append(("gemv partial-sum rmsnorm, 8 rows", sums_pair))

    def sums_pair3():
        y1 = torch.empty(3, 4096, dtype=torch.float16, device=dev)
        ss = torch.empty(3, 256, dtype=torch.float32, device=dev)
        ops.gemv_grouped_sums(a8[:3], [dict(qn=qo, mn=mo, bits=3, mode=ops.MODE_HQQ, N=4096, y=y1, residual=r8[:3])], 4096, sums_out=ss)
        ys = [torch.empty(3, nn, dtype=torch.float16, device=dev) for _, _, _, nn in segs]
        ops.gemv_grouped_sums(y1, [dict(qn=q, mn=mt, bits=b, mode=ops.MODE_HQQ, N=nn, y=y) for (b, q, mt, nn), y in zip(segs, ys)], 4096,
                              gamma=g8, eps=1e-5, sums_in=ss)
        return torch.cat([y1] + ys, dim=1)
    cases.append(("gemv partial-sum rmsnorm, 3 rows", sums_pair3))

    # ---- few-row launches over fragment-ordered x: the tile form and the streaming form (3 / 6 column blocks per workgroup)
    k2, n2 = 4096, 2048
    for bits in (3, 4):
        qn, mn = _layer(bits, n2, k2, seed=60 + bits)
        x64 = rnd(64, k2)
        xf = ops.xfrag(x64, 64, k2)
        for form, blocks in ((1, 0), (2, 3), (2, 6)):
            def few(qn=qn, mn=mn, bits=bits, xf=xf, form=form, blocks=blocks):
                y = torch.empty(64, n2, dtype=torch.float16, device=dev)
                ops.gemm_xfrag_grouped(xf, 64, [dict(qn=qn, mn=mn, bits=bits, mode=ops.MODE_HQQ, N=n2, y=y)], k2, form=form, blocks_per_wg=blocks)
                return y
            cases.append((f"few-row {bits}b form {form}/{blocks}", few))

    # ---- many-row GEMMs: every hand-written route with a counted wait
    for bits in (2, 3, 4):
        qn, mn = _layer(bits, n2, k2, seed=70 + bits)
        for m, routes in ((1024, (ops.GEMM_RING, ops.GEMM_RING128, ops.GEMM_DEQ) + ((ops.GEMM_WS,) if hasattr(ops, "GEMM_WS") else ())), (320, (ops.GEMM_RING128, ops.GEMM_DEQ))):
            xm = rnd(m, k2)
            for route in routes:
                cases.append((f"gemm {bits}b {m} rows route {route}", lambda xm=xm, qn=qn, mn=mn, bits=bits, route=route: ops.gemm(xm, qn, mn, bits, ops.MODE_HQQ, n2, k2, route=route)))
    # the dense ping-pong GEMM directly (several tiles per workgroup: the per-tile drain), fp16 and bf16
    xd, wd = rnd(1536, 1024), (torch.randn(2304, 1024, generator=gen) * 0.05).to(torch.float16).to(dev)
    cases.append(("gemm_f16w 1536x1024x2304", lambda: ops.gemm_f16w(xd, wd)))
    from amq_amd.hqq_format import random_hqq
    hb = random_hqq(1024, 2048, 4, seed=91).to(dev)
    qb, mb = ops.repack_from_hqq(hb.W_q, hb.scale.reshape(-1).float().to(torch.bfloat16), hb.zero.reshape(-1).float().to(torch.bfloat16), 4, 1024, 2048)
    xb = torch.randn(512, 2048, generator=gen).to(torch.bfloat16).to(dev)
    cases.append(("gemm bf16 512 rows", lambda: ops.linear_bf16(xb, qb, mb, 4, 1024, 2048)))
    xb8 = xb[:8].contiguous()
    cases.append(("gemv bf16 8 rows", lambda: ops.linear_bf16(xb8, qb, mb, 4, 1024, 2048)))

    # ---- decode attention, split over the context (ring of row loads, ticket + last-arriver combine) and the single-workgroup kernel
    # (grouped-query heads: the staged MFMA kernel -- LDS-DMA stage buffers, one drain per stage -- and its combine launch)
    for max_seq, pos, B, hq, hkv in ((600, 599, 1, 32, 32), (2048, 2047, 1, 32, 32), (4000, 3999, 1, 32, 8), (1024, 700, 4, 32, 32), (512, 300, 2, 32, 32),
                                     (8192, 8000, 1, 28, 4), (2048, 1500, 3, 8, 2)):
        q, kk, vv = rnd(B, hq * 128), rnd(B, hkv * 128), rnd(B, hkv * 128)
        kc, vc = rnd(B, hkv, max_seq, 128), rnd(B, hkv, max_seq, 128)
        table = ops.rope_table(max_seq, 10000.0, dev)

        def att(q=q, kk=kk, vv=vv, kc=kc, vc=vc, table=table, pos=pos, B=B, hq=hq, hkv=hkv):
            out = torch.empty(B, hq * 128, dtype=torch.float16, device=dev)
            ops.attn_decode(q, kk, vv, kc, vc, out, pos, hq, hkv, table=table)
            return out
        cases.append((f"attn decode {max_seq} keys x{B}", att))
    # ---- prompt attention (MFMA kernel, K / V tiles by LDS-DMA)
    S, hq, hkv = 320, 8, 2
    q, kk, vv = rnd(S, hq * 128), rnd(S, hkv * 128), rnd(S, hkv * 128)

    def pre():
        out = torch.empty(S, hq * 128, dtype=torch.float16, device=dev)
        ops.attn_prefill(q, kk, vv, out, S, hq, hkv)
        return out
    cases.append(("attn prefill 320", pre))
    return cases


def _equal(a, b):
    return torch.equal(a.view(torch.int16) if a.dtype in (torch.float16, torch.bfloat16) else a, b.view(torch.int16) if b.dtype in (torch.float16, torch.bfloat16) else b)


def test_product_equals_its_conservative_twin_quiet_and_perturbed():
    from amq_amd import _lib
    twin = _lib.open_twin()
    assert twin.amq_version() == _lib.load().amq_version()
    dev = _dev()
    cases = _corpus()
    assert len(cases) >= 60
    print(f"twin corpus: {len(cases)} cases x {REPS} repetitions x 2 modes x 2 libraries")
    side = torch.cuda.Stream(device=dev)
    big_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    big_b = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    bad = []
    for name, fn in cases:
        ref = fn()
        torch.cuda.synchronize()
        for perturbed in (False, True):
            for rep in range(REPS):
                if perturbed:
                    with torch.cuda.stream(side):                 # 4 GB through the memory system beside the launch
                        big_b.copy_(big_a, non_blocking=True)
                        big_a.copy_(big_b, non_blocking=True)
                got_p = fn()
                with _lib.routed_to(twin):
                    got_s = fn()
                if not _equal(got_p, ref):
                    bad.append(f"{name}: product differs from its first run ({'perturbed' if perturbed else 'quiet'}, rep {rep})")
                    break
                if not _equal(got_s, ref):
                    bad.append(f"{name}: product differs from the conservative twin ({'perturbed' if perturbed else 'quiet'}, rep {rep})")
                    break
            torch.cuda.synchronize()
    assert not bad, f"{len(bad)} of {len(cases)} cases:\n" + "\n".join(bad)
