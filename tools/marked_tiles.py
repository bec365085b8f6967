#!/usr/bin/env python3
"""VERDICT r5 item 4(i), priced before it is built: how many 16-row x 128-column tiles could take the one-shift-per-dword exact unpack (`AMQ_SD_E9`)?

The E = -9 body uses every field where the packing left it (one shift per dword instead of one per pair: +2.6 % on the 70B replica, +1.7 % on 7B,
HISTORY R5) and stays bit-identical to `Quantizer.dequantize` only while the first rounding RN16((q - z) 2^E) neither loses bits nor meets a
subnormal: |z| >= 2^-5 and |q - z| >= 2^-5 for every code q the field can hold.  Deciding per (row, group) at repack time costs 2^bits
evaluations; a TILE (one wave's unit: a wave-uniform branch) may take the body only if all of its 16 (row, group) pairs qualify.

This tool counts, on the CPU (no GPU needed): the share of qualifying groups and of fully qualifying tiles for (a) the bench's synthetic layers
(hqq_format.random_hqq: fractional zeros), (b) the tiny checkpoints the REFERENCE's quantizer wrote (tests/golden/ckpt: HQQ-optimised zeros) and
(c) the reference's own golden layers (tests/golden/hqq_b*.npz), and prints the bound on what the marked-tile kernel could gain.
usage: marked_tiles.py [out.txt]"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from amq_amd.hqq_format import random_hqq
from amq_amd.checkpoint import load_hqq_dir
from amq_amd.hqq_format import HQQWeights

GAIN_ALL_TILES = {"70B": 0.026, "7B": 0.017}          # measured with EVERY tile on the E9 body (HISTORY R5; amq_gemv_body.cuh AMQ_SD_E9)
lines = []


def emit(s=""):
    print(s)
    lines.append(s)


def shares(zero, bits, n, k, group):
    """zero: fp16 [n * k / group] (row-major groups) -> (share of qualifying groups, share of 16 x 128 tiles whose groups all qualify)"""
    z = np.asarray(zero, np.float32).reshape(n, k // group)
    q = np.arange(2 ** bits, dtype=np.float32)
    ok = (np.abs(z) >= 2.0 ** -5) & (np.abs(q[None, None, :] - z[:, :, None]).min(-1) >= 2.0 ** -5)
    per128 = ok if group == 128 else ok.reshape(n, k // 128, 128 // group).all(-1) if group < 128 else np.repeat(ok, group // 128, axis=1)
    tiles = per128.reshape(n // 16, 16, -1).all(1)
    return float(ok.mean()), float(tiles.mean())


emit("share of (row, group) pairs / of 16 x 128 tiles that qualify for the one-shift-per-dword exact unpack (|z|, |q - z| >= 2^-5 for all codes)")
emit(f"{'weights':44s} {'bits':>4s} {'groups ok':>10s} {'tiles ok':>9s}")
tile_share = {}
for bits in (2, 3, 4):
    h = random_hqq(4096, 4096, bits, seed=bits)
    g, t = shares(h.zero.numpy(), bits, 4096, 4096, 128)
    tile_share[("synthetic", bits)] = t
    emit(f"{'synthetic (random_hqq 4096 x 4096)':44s} {bits:4d} {g:10.3f} {t:9.3f}")
for bits in (2, 3, 4):
    d = os.path.join(ROOT, "tests", "golden", "ckpt", f"{bits}bit")
    if not os.path.isdir(d):
        continue
    _, mods = load_hqq_dir(d)
    gs, ts, w = [], [], []
    for name, m in mods.items():
        if isinstance(m, HQQWeights):
            n, k = m.shape
            g, t = shares(m.zero.float().numpy(), bits, n, k, m.group_size)
            gs.append(g); ts.append(t); w.append(n * k)
    emit(f"{'reference-written checkpoint (tiny Llama)':44s} {bits:4d} {np.average(gs, weights=w):10.3f} {np.average(ts, weights=w):9.3f}")
    tile_share[("ckpt", bits)] = float(np.average(ts, weights=w))
for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "hqq_b*.npz"))):
    d = np.load(p)
    bits, (n, k) = int(d["nbits"]), tuple(int(v) for v in d["shape"])
    g, t = shares(d["zero"].astype(np.float32).reshape(-1), bits, n, k, 128)
    emit(f"{'reference layer ' + os.path.basename(p):44s} {bits:4d} {g:10.3f} {t:9.3f}")
emit()
best = max(tile_share.values())
emit(f"best case over the rows above: {best:.3f} of the tiles on the faster body")
for model, gain in GAIN_ALL_TILES.items():
    emit(f"  {model}: all tiles on it measured +{100 * gain:.1f} %  ->  marked tiles alone: at most +{100 * gain * best:.2f} % (before the cost of the per-tile branch and of a second body's registers)")
emit("decision: not built -- the bound is below what one launch-time measurement resolves (~2 %), an order of magnitude short of the 0.48 -> 0.52 asked for")
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
