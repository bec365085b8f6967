#!/usr/bin/env python3
"""Per-workgroup phase timeline of the streaming few-row kernel (diagnostic build: make -C amq_amd/csrc tuvariant TU=amq_gemm_fewrow TAG=fsstamp EXTRA=-DAMQ_FS_STAMP;
python tools/with_variant.py fsstamp tools/stamp_fewrow.py [rows] [blocks per workgroup] [qkv|gateup])."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import _lib, ops
from amq_amd.llama import _synthetic_linear

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
which = sys.argv[3] if len(sys.argv) > 3 else "gateup"
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
K = 4096
segs = [(4096, K)] * 3 if which == "qkv" else [(11008, K)] * 2
_lib.load()
setst = ctypes.CDLL(_lib.LIB_PATH).amq_debug_set_fs_stamps
setst.argtypes = [ctypes.c_void_p]
copies = 12
w = [[_synthetic_linear(n, k, 3, gen, dev) for n, k in segs] for _ in range(copies)]
x = torch.randn(rows, K, device=dev, generator=gen).half()
xf = ops.xfrag(x, rows, K)
ys = [torch.empty(rows, n, device=dev, dtype=torch.float16) for n, _ in segs]
stamps = torch.zeros(2048, 64, dtype=torch.int64, device=dev)


def launch(i):
    ops.gemm_xfrag_grouped(xf, rows, [l.seg(y) for l, y in zip(w[i % copies], ys)], K, form=2, blocks_per_wg=blocks)


for i in range(2 * copies):
    launch(i)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for i in range(8):
            launch(i)
        setst(ctypes.c_void_p(stamps.data_ptr()))
        launch(8)
        setst(None)
        launch(9)
torch.cuda.current_stream().wait_stream(side)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
s = stamps.cpu().numpy()
s = s[s[:, 0] != 0]
t0 = s[:, 0:8].min()
us = lambda a: (a - t0) / 100.0
ent, primed, loop_end, done, summed = us(s[:, 0:8]), us(s[:, 8:16]), us(s[:, 16:24]), us(s[:, 24:32]), us(s[:, 32:40])
hw = s[:, 40]
xcc = (hw >> 32) & 0xF
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
q = lambda a: "min %.2f p50 %.2f p90 %.2f max %.2f" % (a.min(), np.percentile(a, 50), np.percentile(a, 90), a.max())
print(f"{which} rows {rows} blocks/wg {blocks}: {len(s)} workgroups on {len(np.unique(cu))} CUs")
print("  entry (wave)              ", q(ent))
print("  primed loads landed - entry", q(primed - ent))
print("  main loop (end - primed)   ", q(loop_end - primed))
print("  loop-end skew inside a WG  ", q(loop_end.max(1) - loop_end.min(1)))
print("  last loop end -> partials in LDS (first barrier)", q(summed.min(1) - loop_end.max(1)))
print("  first barrier -> workgroup done", q(done.max(1) - summed.min(1)))
print("  workgroup exit             ", q(done.max(1)), " => span %.2f us" % done.max())
