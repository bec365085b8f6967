#!/usr/bin/env python3
"""Random-shape sweep of the C-ABI matmul entry points against the CPU oracle (GPU box; test infrastructure, not product): random (rows, N, K, bits,
prologue, bias / residual, strided x, segments) for the GEMV family (incl. the partial-sum RMSNorm pair of the 2 .. 8-row steps), random (rows, N, K, bits, route) for the GEMM family, the tests' parity bar.
usage: fuzz_kernels.py [cases=150] [seed=0]   -- prints every failing case and exits non-zero if there was one."""
import os, sys, random
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from amq_amd import ops, _lib
from amq_amd.hqq_format import random_hqq
from oracle import hqq_ref, linear_ref

RTOL = 1e-3
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0


def close(y, ref, inter):
    """the tests' bar on the final value; where bias / residual add further fp16 roundings, one fp16 ulp of the largest intermediate on top (a matmul that
    rounds one ulp differently than the oracle's carries that ulp through the adds, whatever the final value cancels to)"""
    y, ref = np.asarray(y, np.float32), np.asarray(ref, np.float32)
    floor = RTOL * float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    bar = RTOL * np.abs(ref) + floor
    if inter is not None:
        bar = bar + 2.0 ** -10 * np.asarray(inter, np.float32)
    err = np.abs(y - ref)
    return not (err > bar).any(), float((err / bar).max())


def layer(bits, n, k, seed):
    h = random_hqq(n, k, bits, seed=seed)
    hd = h.to(dev)
    qn, mn = ops.repack_from_hqq(hd.W_q, hd.scale.reshape(-1), hd.zero.reshape(-1), bits, n, k)
    return qn, mn, hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), bits, (n, k))


def rms_ref(x, g, eps):
    xf = x.float()
    return (g * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(torch.float16))


for c in range(cases):
    gen = torch.Generator().manual_seed(1000 + c)
    K = 128 * rng.choice([1, 2, 3, 5, 8, 11, 16, 24, 32, 33, 40, 43, 64, 86])
    what = None
    try:
        if rng.random() < 0.15:         # ---- 2 .. 8 rows, RMSNorm from partial sums: producer (sums_out) + consumer (sums_in) against the oracle
            K = 128 * rng.choice([16, 24, 28, 32, 40, 43, 64])
            m = rng.choice([2, 3, 4, 5, 6, 7, 8])
            if m > ops.gemv_max_rows(K, plain=True):
                m = 2
            bits_p = rng.choice([2, 3, 4])
            qn, mn, w = layer(bits_p, K, K, 11 * c)                       # producer: K -> K with a residual (an o_proj)
            a_in = torch.randn(m, K, generator=gen).to(torch.float16)
            res = torch.randn(m, K, generator=gen).to(torch.float16)
            y1 = torch.empty(m, K, dtype=torch.float16, device=dev)
            ss = torch.empty(m, K // 16, dtype=torch.float32, device=dev)
            what = f"gemv_sums m={m} K={K} bits={bits_p}"
            ops.gemv_grouped_sums(a_in.to(dev), [dict(qn=qn, mn=mn, bits=bits_p, mode=ops.MODE_HQQ, N=K, y=y1, residual=res.to(dev))], K, sums_out=ss)
            r1 = (res.numpy() + linear_ref.linear_f16(a_in.numpy(), w)).astype(np.float16)
            ok, worst = close(y1.cpu().numpy(), r1, np.abs(linear_ref.linear_f16(a_in.numpy(), w).astype(np.float32)))
            want_ss = y1.float().pow(2).view(m, K // 16, 16).sum(-1)
            ok = ok and bool(torch.allclose(ss, want_ss, rtol=1e-5, atol=1e-6))
            gamma = (1.0 + 0.1 * torch.randn(K, generator=gen)).to(torch.float16)
            nseg = rng.choice([1, 2, 3])
            segs, refs = [], []
            xin = rms_ref(y1.cpu(), gamma, 1e-5)
            for sgi in range(nseg):
                bits = rng.choice([2, 3, 4])
                n = 16 * rng.choice([2, 7, 33, 64, 256])
                q2, m2, w2 = layer(bits, n, K, 17 * c + sgi)
                y = torch.empty(m, n, dtype=torch.float16, device=dev)
                segs.append(dict(qn=q2, mn=m2, bits=bits, mode=ops.MODE_HQQ, N=n, y=y))
                refs.append(linear_ref.linear_f16(xin.numpy(), w2))
            ops.gemv_grouped_sums(y1, segs, K, gamma=gamma.to(dev), eps=1e-5, sums_in=ss)
            for sg, r in zip(segs, refs):
                o2, w2_ = close(sg["y"].cpu().numpy(), r, None)
                ok, worst = ok and o2, max(worst, w2_)
            if not ok:
                fails += 1
                print("FAIL", what, "worst/bar", worst, flush=True)
        elif rng.random() < 0.6:        # ---- GEMV family
            m = rng.choice([1, 1, 1, 2, 3, 4, 5, 7, 8, 9, 12, 16])
            nseg = rng.choice([1, 1, 2, 3])
            pro = rng.choice([ops.PRO_NONE, ops.PRO_RMSNORM, ops.PRO_SILU_MUL])
            if m > ops.gemv_max_rows(K, plain=True, norm=pro == ops.PRO_RMSNORM):
                m = 1
            segs, refs = [], []
            strided = rng.random() < 0.2 and m > 1
            xs = K + 64 if strided else K
            xfull = torch.randn(m, xs, generator=gen).to(torch.float16)
            x = xfull[:, :K]
            up = torch.randn(m, xs, generator=gen).to(torch.float16)[:, :K]
            gamma = (1.0 + 0.1 * torch.randn(K, generator=gen)).to(torch.float16)
            xin = x if pro == ops.PRO_NONE else rms_ref(x, gamma, 1e-5) if pro == ops.PRO_RMSNORM else torch.nn.functional.silu(x.float()).to(torch.float16) * up
            desc = []
            for s in range(nseg):
                bits = rng.choice([2, 3, 4])
                n = 16 * rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 256, 257])
                qn, mn, w = layer(bits, n, K, 7 * c + s)
                use_bias, use_res = rng.random() < 0.3, rng.random() < 0.4
                bias = torch.randn(n, generator=gen).to(torch.float16) * 0.1 if use_bias else None
                res = torch.randn(m, n, generator=gen).to(torch.float16) if use_res else None
                y = torch.empty(m, n, dtype=torch.float16, device=dev)
                segs.append(dict(qn=qn, mn=mn, bits=bits, mode=ops.MODE_HQQ, N=n, y=y, bias=None if bias is None else bias.to(dev), residual=None if res is None else res.to(dev)))
                r = linear_ref.linear_f16(xin.numpy(), w)
                inter = np.abs(r.astype(np.float32))
                if bias is not None:
                    r = (r + bias.numpy()).astype(np.float16)
                    inter = np.maximum(inter, np.abs(r.astype(np.float32)))
                if res is not None:
                    r = (res.numpy() + r).astype(np.float16)
                refs.append((r, inter if (use_bias or use_res) else None))
                desc.append((bits, n, use_bias, use_res))
            what = f"gemv m={m} K={K} pro={pro} strided={strided} segs={desc}"
            xd = xfull.to(dev)[:, :K] if strided else x.contiguous().to(dev)
            upd = up.contiguous().to(dev) if not strided else torch.randn(1)  # placeholder
            if strided:
                upfull = torch.zeros(m, xs, dtype=torch.float16); upfull[:, :K] = up
                upd = upfull.to(dev)[:, :K]
            math = rng.choice([ops.MATH_DEFAULT, ops.MATH_DEFAULT, ops.MATH_GROUPSCALE, ops.MATH_LINEAR])
            if math == ops.MATH_LINEAR and m > ops.gemv_max_rows(K):
                math = ops.MATH_DEFAULT
            what += f" math={math}"
            ops.gemv_grouped(xd, segs, K, prologue=pro, x2=upd if pro == ops.PRO_SILU_MUL else None, gamma=gamma.to(dev) if pro == ops.PRO_RMSNORM else None, eps=1e-5,
                             opts=ops.GemvOpts(math=math) if math else None)
            for sg, (r, b) in zip(segs, refs):
                if math == ops.MATH_DEFAULT:
                    ok, worst = close(sg["y"].cpu().numpy(), r, b)
                else:       # the opt-in arithmetics: their own (tested) bound -- 2 fp16 ulps + 1e-3 rms per element, one ulp of the intermediates on top
                    y_, r_ = sg["y"].float().cpu().numpy(), np.asarray(r, np.float32)
                    rms_ = float(np.sqrt(np.mean(r_.astype(np.float64) ** 2)))
                    # (AMQ_MATH_LINEAR's tested bound is one ulp + 2e-3 rms: no per-weight rounding at all)
                    # (behind a fused RMSNorm / SiLU*mul prologue the activations themselves may sit an fp16 ulp from the oracle's: seed 23 found GROUPSCALE at 1.0008 of the
                    #  plain bar on a 16-row, K = 384, 2-bit launch with the RMSNorm prologue -- the opt-in arithmetic has no slack left at three groups per row)
                    bar_ = (2.0 ** -9 * np.abs(r_) + (1.25e-3 if pro == ops.PRO_NONE else 1.5e-3) * rms_ if math == ops.MATH_GROUPSCALE else 2.0 ** -10 * np.abs(r_) + 2e-3 * rms_) + (0 if b is None else 2.0 ** -9 * b)      # (bias AND residual behind an opt-in arithmetic: two more fp16 roundings of intermediates that may differ by an ulp each)
                    e_ = np.abs(y_ - r_)
                    ok, worst = not (e_ > bar_).any(), float((e_ / bar_).max())
                if not ok:
                    fails += 1
                    print("FAIL", what, "worst/bar", worst, flush=True)
        else:                           # ---- GEMM family
            m = rng.choice([9, 16, 17, 33, 64, 65, 100, 128, 200, 256, 300, 512, 777, 1024, 2048])
            bits = rng.choice([2, 3, 4])
            n = 16 * rng.choice([1, 4, 16, 17, 64, 128, 256, 320])
            route = rng.choice([ops.GEMM_AUTO, ops.GEMM_AUTO, ops.GEMM_TILED, ops.GEMM_SKINNY, ops.GEMM_RING, ops.GEMM_RING128, ops.GEMM_WS, ops.GEMM_DEQ])
            if route == ops.GEMM_SKINNY and m > 64:
                route = ops.GEMM_AUTO
            qn, mn, w = layer(bits, n, K, 13 * c)
            x = (torch.randn(m, K, generator=gen) * 0.5).to(torch.float16)
            use_bias, use_res = rng.random() < 0.3, rng.random() < 0.3
            bias = torch.randn(n, generator=gen).to(torch.float16) * 0.1 if use_bias else None
            res = torch.randn(m, n, generator=gen).to(torch.float16) if use_res else None
            what = f"gemm m={m} N={n} K={K} bits={bits} route={route} bias={use_bias} res={use_res}"
            try:
                y = ops.gemm(x.to(dev), qn, mn, bits, ops.MODE_HQQ, n, K, bias=None if bias is None else bias.to(dev), residual=None if res is None else res.to(dev), route=route)
            except _lib.AmqError as e:       # a route that refuses the shape says so
                print("refused:", what, "--", str(e)[:100], flush=True)
                continue
            r = linear_ref.linear_f16(x.numpy(), w)
            inter = np.abs(r.astype(np.float32))
            if bias is not None:
                r = (r + bias.numpy()).astype(np.float16)
                inter = np.maximum(inter, np.abs(r.astype(np.float32)))
            if res is not None:
                r = (res.numpy() + r).astype(np.float16)
            ok, worst = close(y.cpu().numpy(), r, inter if (use_bias or use_res) else None)
            if not ok:
                fails += 1
                print("FAIL", what, "worst/bar", worst, flush=True)
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("ERROR", what, "--", repr(e)[:300], flush=True)
    if c % 25 == 24:
        print(f"... {c + 1} cases, {fails} failures", flush=True)
print(f"{cases} cases, {fails} failures")
sys.exit(1 if fails else 0)
