#!/usr/bin/env python3
"""Per-workgroup phase timeline of one GEMV launch (diagnostic build only).

    make -C amq_amd/csrc variant TAG=stamp EXTRA=-DAMQ_STAMP
    python tools/with_variant.py stamp tools/stamp_gemv.py N K bits [M [pro]] -> gpurun_out/stamps_<N>x<K>_b<bits>.npz
    (M: x rows, default 1; pro: 0 none / 1 RMSNorm / 2 SiLU*mul prologue through amq_gemv_grouped_f16, default 0; build the variant with
     `make -C amq_amd/csrc gemvvariant TAG=stamp EXTRA=-DAMQ_STAMP`)

Runs a rotation of launches over distinct weight buffers (cold weights), stamps the
last few, and prints phase statistics in us relative to the earliest workgroup entry."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import _lib, ops  # noqa: E402
from amq_amd.hqq_format import random_hqq  # noqa: E402


def main():
    n, k, bits = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    M = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    pro = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    dev = torch.device("cuda:0")
    lib = _lib.load()
    setst = ctypes.CDLL(_lib.LIB_PATH).amq_debug_set_stamps
    setst.argtypes = [ctypes.c_void_p]
    h = random_hqq(n, k, bits, seed=1).to(dev)
    qn0, mn0 = ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), bits, n, k)
    per = qn0.numel() * 4 + mn0.numel() * 2
    copies = max(2, min(64, (768 << 20) // per + 1))
    bufs = [(qn0.clone(), mn0.clone()) for _ in range(copies)]
    x = torch.randn(M, k, device=dev).half()
    x2 = torch.randn(M, k, device=dev).half()
    gamma = torch.ones(k, device=dev, dtype=torch.float16)
    y = torch.empty(M, n, device=dev, dtype=torch.float16)

    opts = ops.GemvOpts(waves=int(os.environ["GEMV_WAVES"])) if os.environ.get("GEMV_WAVES") else None      # A/B: e.g. 520 = WAVES_2X8

    def launch(q, mt):
        ops.gemv_grouped(x, [dict(qn=q, mn=mt, bits=bits, mode=ops.MODE_HQQ, N=n, y=y)], k, prologue=pro, x2=x2, gamma=gamma, eps=1e-5, opts=opts)
    nstamp = 4
    stamps = torch.zeros(nstamp, 4096, 128, dtype=torch.int64, device=dev)
    for rep in range(3):
        for i in range(copies):
            launch(bufs[i][0], bufs[i][1])
    torch.cuda.synchronize()
    def seq():
        for j in range(nstamp):
            for i in range(8):
                launch(*bufs[(j * 9 + i) % copies])
            setst(ctypes.c_void_p(stamps[j].data_ptr()))
            launch(*bufs[(j * 9 + 8) % copies])
            setst(None)
    if os.environ.get("STAMP_GRAPH", "0") == "1":
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                seq()
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            graph.replay()
        print("mode: hipGraph replay")
    else:
        seq()
        print("mode: eager")
    torch.cuda.synchronize()
    st = stamps.cpu().numpy()
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed(f"gpurun_out/stamps_{n}x{k}_b{bits}.npz", stamps=st)
    for j in range(nstamp):
        s = st[j]
        live = s[:, 0] != 0
        s = s[live]
        t0 = s[:, 0].min()
        us = lambda a: (a - t0) / 100.0
        ent, primed, staged, ex = us(s[:, 0]), us(s[:, 1]), us(s[:, 2]), us(s[:, 4])
        wend = us(s[:, 8:24].astype(np.int64))
        wend = np.where(s[:, 8:24] != 0, wend, np.nan)
        hw = s[:, 3]
        xcc = (hw >> 32) & 0xF
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
        ncu = len(np.unique(cu))
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        q = lambda a: "min %.2f p50 %.2f p90 %.2f max %.2f" % (np.nanmin(a), np.nanpercentile(a, 50), np.nanpercentile(a, 90), np.nanmax(a))
        print(f"launch {j}: {len(s)} WGs on {ncu} CUs (WGs/CU min {per_cu.min()} max {per_cu.max()})")
        print("  entry        ", q(ent))
        print("  primed-entry ", q(primed - ent))
        print("  staged-entry ", q(staged - ent))
        print("  stream (wave end - staged)", q(wend - staged[:, None]))
        print("  wave-end skew within WG  ", q(np.nanmax(wend, 1) - np.nanmin(wend, 1)))
        print("  exit-last wave end       ", q(ex - np.nanmax(wend, 1)))
        print("  exit         ", q(ex), " => kernel span %.2f us" % ex.max())
        # per-CU busy: last exit on the CU
        order = np.unique(cu, return_inverse=True)[1]
        last = np.zeros(ncu)
        np.maximum.at(last, order, ex)
        print("  per-CU last exit", q(last))
        if (s[:, 6] != 0).all():
            clk = (s[:, 6] - s[:, 5]) / np.maximum(s[:, 4] - s[:, 0], 1) * 100.0   # shader ticks per 10 ns -> MHz
            print("  shader clock MHz (memtime / memrealtime per WG)", q(clk))
        fw = np.where(s[:, 96:112] != 0, us(s[:, 96:112].astype(np.int64)), np.nan)
        fa = np.where(s[:, 112:128] != 0, us(s[:, 112:128].astype(np.int64)), np.nan)
        if np.isfinite(fa).any():
            print("  loop entry (wave) - staged   ", q(fw - staged[:, None]))
            print("  first tile arrived - primed  ", q(fa - primed[:, None]))
            print("  first tile arrived - staged  ", q(fa - staged[:, None]))
        ph = s[:, 32:96].reshape(len(s), 16, 4).astype(np.float64)
        act = ph[:, :, 3] > 0
        if act.any():
            tiles = ph[:, :, 3][act]
            for nm, col in (("wait", 0), ("math+issue", 1), ("rowend", 2)):
                v = ph[:, :, col][act] / tiles
                print(f"  main-loop cycles per tile: {nm:11s}", q(v))
            tot = (ph[:, :, 0] + ph[:, :, 1] + ph[:, :, 2])[act] / tiles
            print("  main-loop cycles per tile: total      ", q(tot), " (tiles in main loop per wave: %.1f)" % tiles.mean())


if __name__ == "__main__":
    main()
