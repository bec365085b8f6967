#!/usr/bin/env python3
"""The dequantize-once route (amq_gemm_f16.hip behind amq_dequantize_f16) against the fused ring kernel and against
dequantize + library GEMM, one process, interleaved rounds, HIP events on the launch stream, random operands.
Checks first: the dense kernel against an fp32 matmul (whole output, ragged M / N, every epilogue), then every route.
usage: f16pp_bench.py [--shapes N,K;N,K] [--m 2048,32768] [--bits 3] [--rounds 5] [--check-only]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops
from amq_amd.llama import _synthetic_linear

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="13824,5120;5120,13824;5120,5120")
ap.add_argument("--m", default="2048,8192,32768")
ap.add_argument("--bits", default="3")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--routes", default="3,6,lib,dense,torch")
ap.add_argument("--check-only", action="store_true")
ap.add_argument("--no-check", action="store_true", help="timing-only ablation builds compute garbage")
args = ap.parse_args()
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)


def check_dense():
    worst = 0.0
    for (m, n, k) in ((256, 256, 128), (512, 512, 256), (300, 272, 384), (1000, 1040, 1152), (4096, 5120, 5120), (257, 16, 128)):
        x = (torch.randn(m, k, device=dev, generator=gen) * 0.5).half()
        w = (torch.randn(n, k, device=dev, generator=gen) * 0.05).half()
        ref = x.float() @ w.float().t()
        rms = ref.pow(2).mean().sqrt().item()
        bias = (torch.randn(n, device=dev, generator=gen) * 0.1).half()
        res = (torch.randn(m, n, device=dev, generator=gen)).half()
        gate = (torch.randn(m, n, device=dev, generator=gen)).half()
        def nerr(got, exp):          # max over elements of |error| / (2 fp16 ulps of the value + 2e-3 rms): must stay below 1
            exp = exp.float()
            bar = exp.abs() * 2.0 ** -9 + 2e-3 * exp.pow(2).mean().sqrt()
            return ((got.float() - exp).abs() / bar).max().item()
        y = ops.gemm_f16w(x, w)
        e0 = nerr(y, ref)
        yb = ops.gemm_f16w(x, w, bias=bias)
        e1 = nerr(yb, ref.half() + bias)
        yr = ops.gemm_f16w(x, w, bias=bias, residual=res)
        e2 = nerr(yr, res + (ref.half() + bias))
        yg = ops.gemm_f16w(x, w, gate=gate)
        e3 = nerr(yg, torch.nn.functional.silu(gate.float()).half() * ref.half())
        # in-place forms
        r2 = res.clone(); ops.gemm_f16w(x, w, bias=bias, residual=r2, out=r2)
        g2 = gate.clone(); ops.gemm_f16w(x, w, gate=g2, out=g2)
        same = bool((r2 == yr).all().item()) and bool((g2 == yg).all().item())
        det = bool((ops.gemm_f16w(x, w) == y).all().item())
        print(json.dumps({"check": "dense", "M": m, "N": n, "K": k, "err": round(e0, 6), "err_bias": round(e1, 6), "err_res": round(e2, 6),
                          "err_gate": round(e3, 6), "inplace_same": same, "repeat_same": det}), flush=True)
        worst = max(worst, e0, e1, e2, e3)
        assert same and det
    assert worst < 1.0, worst


def run(route, x, l, w, y):
    if route == "lib":
        ops.LIB_GEMM_ROWS = 1
        ops.gemm(x, l.qn, l.mn, l.bits, l.mode, l.N, l.K, out=y)
        ops.LIB_GEMM_ROWS = 0
    elif route == "dense":
        ops.gemm_f16w(x, w, out=y)
    elif route == "torch":
        torch.matmul(x, w.t(), out=y)
    else:
        ops.gemm(x, l.qn, l.mn, l.bits, l.mode, l.N, l.K, out=y, route=int(route))


if not args.no_check:
    check_dense()
ops.LIB_GEMM_ROWS = 0
routes = args.routes.split(",")
for shp in args.shapes.split(";"):
    n, k = (int(v) for v in shp.split(","))
    for bits in (int(v) for v in args.bits.split(",")):
        l = _synthetic_linear(n, k, bits, gen, dev)
        w = ops.dequantize(l.qn, l.mn, bits, l.mode, n, k)
        for m in (int(v) for v in args.m.split(",")):
            x = (torch.randn(m, k, device=dev, generator=gen) * 0.5).half()
            y = torch.empty(m, n, device=dev, dtype=torch.float16)
            ref = x[:512].float() @ w.float().t()
            rms = ref.pow(2).mean().sqrt().item()
            times = {r: [] for r in routes}
            errs, outs = {}, {}
            for r in routes:
                y.zero_()
                run(r, x, l, w, y)
                torch.cuda.synchronize()
                errs[r] = ((y[:512].float() - ref).abs().max().item()) / rms
                outs[r] = y[-300:].clone()
            out = {"N": n, "K": k, "bits": bits, "M": m}
            if "3" in outs and "6" in outs:
                out["ring_eq_deq"] = bool((outs["3"] == outs["6"]).all().item())
            if "6" in outs and "dense" in outs:
                out["deq_eq_dense"] = bool((outs["6"] == outs["dense"]).all().item())
            if not args.check_only:
                for _ in range(args.rounds):
                    for r in routes:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        run(r, x, l, w, y)
                        e1.record()
                        torch.cuda.synchronize()
                        times[r].append(e0.elapsed_time(e1) * 1e3)
            fl = 2.0 * m * n * k
            for r in routes:
                if times[r]:
                    t = sorted(times[r])
                    out[f"us_{r}"] = round(t[len(t) // 2], 1)
                    out[f"TF_{r}"] = round(fl / t[len(t) // 2] / 1e6, 1)
                out[f"err_{r}"] = round(errs[r], 5)
            print(json.dumps(out), flush=True)
