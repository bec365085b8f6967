#!/usr/bin/env python3
"""Random-shape sweep of the bfloat16 entry points against the CPU oracle (GPU box; test infrastructure, not product): random (rows, N, K, bits, bias,
residual, strides) through amq_gemm_bf16 / amq_gemv_bf16, the tests' bar (weights bit-exact, y within one bf16 ulp; a further ulp of the largest
intermediate per extra bf16 add).  usage: fuzz_bf16.py [cases=120] [seed=0]   -- prints every failing case and exits non-zero if there was one."""
import os, sys, random
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from amq_amd import ops, _lib
from amq_amd.hqq_format import random_hqq
from oracle import hqq_ref

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = _lib.load()
fails = 0


def bits_of(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


for c in range(cases):
    gen = torch.Generator().manual_seed(5000 + c)
    K = 128 * rng.choice([1, 2, 3, 5, 8, 11, 16, 24, 32, 33, 43, 64, 86])
    N = 16 * rng.choice([1, 2, 3, 7, 16, 33, 64, 65, 128, 256, 257, 688])
    bits = rng.choice([2, 3, 4])
    m = rng.choice([1, 1, 1, 2, 3, 5, 8, 15, 16, 17, 31, 64, 100, 256, 257, 700])
    use_bias, use_res = rng.random() < 0.4, rng.random() < 0.4
    xs = K + 8 * rng.choice([0, 0, 1, 8])
    ys = N + 4 * rng.choice([0, 0, 1, 8])
    what = f"case {c}: rows {m} N {N} K {K} bits {bits} bias {use_bias} residual {use_res} x_stride {xs} y_stride {ys}"
    try:
        h = random_hqq(N, K, bits, seed=c)
        sb, zb = h.scale.float().to(torch.bfloat16), h.zero.float().to(torch.bfloat16)
        wb = hqq_ref.dequantize_bf16(h.W_q.numpy(), bits_of(sb), bits_of(zb), bits, (N, K), 128)
        qn, mn = ops.repack_from_hqq(h.W_q.to(dev), sb.reshape(-1).to(dev), zb.reshape(-1).to(dev), bits, N, K)
        if not np.array_equal(bits_of(ops.dequantize_bf16(qn, mn, bits, N, K)), wb):
            raise AssertionError("dequantized weights differ from the oracle")
        x = torch.randn(m, xs, generator=gen).to(torch.bfloat16)
        bias = (torch.randn(N, generator=gen) * 0.1).to(torch.bfloat16) if use_bias else None
        res = torch.randn(m, ys, generator=gen).to(torch.bfloat16) if use_res else None
        xd = x.to(dev)
        yd = torch.zeros(m, ys, dtype=torch.bfloat16, device=dev)
        rd = res.to(dev) if use_res else None
        need = int(lib.amq_gemm_bf16_workspace_bytes(m, N, K))
        ws = torch.empty(max(need, 2) // 2, dtype=torch.bfloat16, device=dev)
        rc = lib.amq_gemm_bf16(bits, _lib.ptr(xd), _lib.ptr(qn), _lib.ptr(mn), _lib.ptr(bias.to(dev)) if use_bias else None, _lib.ptr(rd), _lib.ptr(yd),
                               m, N, K, 128, xs, ys, _lib.ptr(ws), need, _lib.current_stream())
        if rc != 0:
            raise AssertionError("rc %d: %s" % (rc, lib.amq_last_error().decode()))
        torch.cuda.synchronize()
        w = torch.from_numpy(wb.view(np.int16)).view(torch.bfloat16)
        y0 = torch.nn.functional.linear(x[:, :K].contiguous(), w)
        ref, inter = y0, y0.float().abs()
        if use_bias:
            ref = ref + bias
            inter = torch.maximum(inter, ref.float().abs())
        if use_res:
            ref = res[:, :N] + ref
        got = yd[:, :N].float().cpu().double()
        refd = ref.float().double()
        bar = 2.0 ** -7 * refd.abs() + 2.0 ** -8 * refd.pow(2).mean().sqrt() + ((use_bias + use_res) * 2.0 ** -7) * inter.double()
        worst = float(((got - refd).abs() / bar).max())
        if not worst <= 1.0:
            raise AssertionError("worst element at %.3f of the bar" % worst)
        if ys > N and int(torch.count_nonzero(yd[:, N:])) != 0:
            raise AssertionError("wrote past a row's N outputs")
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("FAIL", what, "--", repr(e), flush=True)
print(f"{cases} cases, {fails} failures")
sys.exit(1 if fails else 0)
