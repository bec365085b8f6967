#!/usr/bin/env python3
"""Per-layer microbenchmark of the weight-streaming kernels (GPU only).

Rotates over enough distinct weight buffers (>= 512 MiB) that neither L2 nor
the 256 MiB Infinity Cache can serve the stream; times with HIP events on the
launch stream.  Prints one JSON line per case: achieved algorithmic GB/s
(BASELINE.md section 3 byte count) and us per call."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops  # noqa: E402
from amq_amd.hqq_format import random_hqq  # noqa: E402


HOT = False
ZERO = False
ROUTE = 0


def layer_bytes(n, k, bits, m=1):
    return n * k * bits // 8 + 4 * n * k // 128 + 2 * m * k + 2 * m * n


def bench_case(n, k, bits, m, iters, fn_name="gemv"):
    dev = torch.device("cuda:0")
    h = random_hqq(n, k, bits, seed=1).to(dev)
    qn0, mn0 = ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), bits, n, k)
    if ZERO:
        qn0.zero_()
    per = qn0.numel() * 4 + mn0.numel() * 2
    copies = max(2, min(64, (768 << 20) // per + 1))
    if HOT:
        copies = 1
    bufs = [(qn0.clone(), mn0.clone()) for _ in range(copies)]
    x = torch.randn(m, k, device=dev).half()
    y = torch.empty(m, n, device=dev, dtype=torch.float16)
    fn = getattr(ops, fn_name)
    if fn_name == "gemm" and ROUTE:
        def fn(x_, q, mt, bits_, mode, n_, k_, out=None):
            return ops.gemm(x_, q, mt, bits_, mode, n_, k_, out=out, route=ROUTE)
    if fn_name == "gemm_xfrag":                       # x pre-arranged in fragment order (ops.xfrag), as a fused producer would
        xf = ops.xfrag(x, m, k)
        ref = ops.gemm(x, qn0, mn0, bits, ops.MODE_HQQ, n, k)
        got = ops.gemm_xfrag(xf, m, qn0, mn0, bits, ops.MODE_HQQ, n, k)
        err = (got.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
        assert err < 2e-3, err

        def fn(x_, q, mt, bits_, mode, n_, k_, out=None):
            return ops.gemm_xfrag(xf, m, q, mt, bits_, mode, n_, k_, out=out)
    for i in range(copies):
        fn(x, bufs[i][0], bufs[i][1], bits, ops.MODE_HQQ, n, k, out=y)
    torch.cuda.synchronize()
    # capture the launch sequence in a HIP graph: replay cost is device-side
    # (kernel + ~1.2-1.5 us inter-kernel boundary), not Python/ctypes time
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for i in range(iters):
                q, mt = bufs[i % copies]
                fn(x, q, mt, bits, ops.MODE_HQQ, n, k, out=y)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (iters * reps)
    b = layer_bytes(n, k, bits, m)
    return {"kernel": fn_name, "N": n, "K": k, "bits": bits, "M": m, "us": round(us, 3),
            "GBps": round(b / us / 1e3, 1), "bytes": b, "copies": copies}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--dot", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--rpt", type=int, default=0)
    ap.add_argument("--math", type=int, default=0)
    ap.add_argument("--route", type=int, default=0, help="GEMM kernel family: 0 auto, 1 tiled, 2 skinny, 3 ring")
    ap.add_argument("--gemv", type=int, default=1)
    ap.add_argument("--gemm", type=int, default=1)
    ap.add_argument("--zero", type=int, default=0, help="1: all-zero packed weights (data-dependent clock check)")
    ap.add_argument("--hot", type=int, default=0, help="1: a single weight buffer (stays in L2 / Infinity Cache)")
    ap.add_argument("--only", default="", help="N,K: just this shape")
    ap.add_argument("--gemm_m", default="64,256,512,1024,4096,16384", help="row counts of the GEMM table")
    ap.add_argument("--gemm_shape", default="5120,5120", help="N,K of the GEMM table")
    ap.add_argument("--gemm_bits", default="4,3,2")
    ap.add_argument("--gemm_fn", default="gemm", help="gemm | gemm_xfrag")
    ap.add_argument("--lib_gemm_rows", type=int, default=0, help="ops.LIB_GEMM_ROWS (0: always time the own kernels)")
    args = ap.parse_args()
    global HOT, ZERO
    HOT = bool(args.hot)
    ZERO = bool(args.zero)
    from amq_amd import _lib
    ops.LIB_GEMM_ROWS = args.lib_gemm_rows
    # per-call options, set explicitly from this tool's command line (the library has no option state)
    if args.dot or args.waves or args.depth or args.rpt or args.math:
        ops.DEFAULT_GEMV_OPTS = ops.GemvOpts(math=args.math, waves=args.waves, depth=args.depth, rpt=args.rpt, dot=args.dot)
    global ROUTE
    ROUTE = args.route
    shapes = [(4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (22016, 4096)]
    if args.quick:
        shapes = shapes[:2]
    if args.only:
        shapes = [tuple(int(v) for v in sh.split(",")) for sh in args.only.split(";")]
    if args.gemv:
        for n, k in shapes:
            for bits in (4, 3, 2):
                print(json.dumps(bench_case(n, k, bits, 1, args.iters)), flush=True)
        for m in (() if args.only else (2, 4, 8)):
            print(json.dumps(bench_case(4096, 4096, 4, m, args.iters)), flush=True)
    if not args.gemm:
        return
    gn, gk = (int(v) for v in args.gemm_shape.split(","))
    for m in (int(v) for v in args.gemm_m.split(",")):
        for bits in (int(v) for v in args.gemm_bits.split(",")):
            r = bench_case(gn, gk, bits, m, max(20, args.iters // 10), args.gemm_fn)
            r["TFLOPs"] = round(2.0 * m * gn * gk / r["us"] / 1e6, 1)
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
