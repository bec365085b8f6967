#!/usr/bin/env python3
"""Per-wave cycle accounting of amq::gemm_f16_pp_kernel's two-phase loop (diagnostic build:
make -C amq_amd/csrc tuvariant TU=amq_gemm_f16 TAG=stamp EXTRA="-DAMQ_PP_STAMP -DAMQ_PP_BUF=1";
run: python tools/with_variant.py stamp tools/f16pp_stamps.py [M N K]).  Shader cycles per phase, mean over the first 256
workgroups' waves, split by wave group (wr = 0 / 1) and phase (X: 16 reads + 2 pieces, Y: 8 reads + 6 pieces):
issue = operand reads + LDS-DMA pieces issued | vm = counted vmcnt wait | lgkm = operand wait | bar1 = barrier before the MFMAs |
mfma = 32 MFMAs issued | tail = stamp wait + barrier after the MFMAs + loop overhead."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops, _lib

m, n, k = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 13824, 5120)
dev = torch.device("cuda:0")
x = (torch.randn(m, k, device=dev) * 0.5).half()
w = (torch.randn(n, k, device=dev) * 0.05).half()
y = torch.empty(m, n, device=dev, dtype=torch.float16)
for _ in range(2):
    ops.gemm_f16w(x, w, out=y)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(256 * 8 * 16, dtype=np.uint32)
fn = lib.amq_debug_pp_stamps
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
a = buf.reshape(256, 8, 16).astype(np.float64)
T = a[0, 0, 12]
nt = ((m + 255) // 256) * ((n + 255) // 256)
grid = min(nt, torch.cuda.get_device_properties(0).multi_processor_count)
tiles = np.array([len(range(b, nt, grid)) for b in range(256)], dtype=np.float64)          # the kernel is persistent: tiles per workgroup
a[:, :, :12] /= np.maximum(tiles, 1.0)[:, None, None]
a = a[:min(grid, 256)]
names = ["issue", "vm", "lgkm", "bar1", "mfma", "tail"]
print(f"M={m} N={n} K={k}: {int(T)} K-tiles per tile; cycles per phase (mean over the workgroups' 4 waves and tiles), ideal: 512 per phase = 32 MFMAs x 16")
for wr in (0, 1):
    for p, pn in ((0, "X"), (1, "Y")):
        v = a[:, 4 * wr:4 * wr + 4, 6 * p:6 * p + 6].mean(axis=(0, 1)) / T
        print(f"  wr={wr} phase {pn}: " + "  ".join(f"{nm} {c:7.1f}" for nm, c in zip(names, v)) + f"   sum {v.sum():7.1f}")
tot = a[:, :, :12].sum(axis=2).mean() / T
print(f"  per K-tile and wave: {tot:.0f} cycles (MFMA-bound floor 2048)")
