#!/usr/bin/env python3
"""MFMA-segment timeline of amq::gemm_f16_pp_kernel (diagnostic build: make -C amq_amd/csrc tuvariant TU=amq_gemm_f16 TAG=trace
EXTRA="-DAMQ_PP_TRACE -DAMQ_PP_BUF=1"; run: python tools/with_variant.py trace tools/f16pp_trace.py).  For the first 128 phases of
16 workgroups: begin / end (shader cycles) of every wave's MFMA burst; prints burst length, the skew of the four waves of a group,
and the hand-over gap = first begin of the next group's burst - last end of this group's."""
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
buf = np.zeros(16 * 8 * 128 * 4, dtype=np.uint64)
fn = lib.amq_debug_pp_trace
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(16, 8, 128, 4).astype(np.int64)      # [.., 0] burst begin, 1 burst end, 2 end of the read segment that preceded the burst
res = {k_: [] for k_ in ("len", "skew_begin", "skew_end", "gap_01", "gap_10", "same_simd_gap", "period", "reader_slack_01", "release_after_last_arrival")}
for g in range(16):
    b, e = t[g, :, 8:120, 0], t[g, :, 8:120, 1]            # [wave, phase]; waves 0-3: wr = 0, 4-7: wr = 1
    res["len"].append((e - b).mean())
    for grp in (slice(0, 4), slice(4, 8)):
        res["skew_begin"].append((b[grp].max(0) - b[grp].min(0)).mean())
        res["skew_end"].append((e[grp].max(0) - e[grp].min(0)).mean())
    # group 0's burst p is followed by group 1's burst p, then group 0's burst p + 1
    res["gap_01"].append((b[4:8].min(0) - e[0:4].max(0)).mean())
    res["gap_10"].append((b[0:4, 1:].min(0) - e[4:8, :-1].max(0)).mean())
    res["same_simd_gap"].append((b[4:8] - e[0:4]).mean())   # the partner's begin minus this wave's end, same SIMD (waves w and w + 4)
    res["period"].append((b[0, 1:] - b[0, :-1]).mean())
    rdy = t[g, :, 8:120, 2]
    # hand-over group 0 -> group 1: the barrier needs group 0's bursts ended AND group 1's read segments ended
    res["reader_slack_01"].append((e[0:4].max(0) - rdy[4:8].max(0)).mean())       # > 0: the readers were there first
    res["release_after_last_arrival"].append((b[4:8].min(0) - np.maximum(e[0:4].max(0), rdy[4:8].max(0))).mean())
for k_, v in res.items():
    print(f"{k_:14s} mean {np.mean(v):8.1f}   min {np.min(v):8.1f}   max {np.max(v):8.1f}")
g = 3
b, e = t[g, :, 40:44, 0], t[g, :, 40:44, 1]
base = b.min()
print("workgroup 3, phases 40-43: [begin, end) per wave, cycles from the first begin")
for wv in range(8):
    print(f"  wave {wv} (wr={wv >> 2}, SIMD {wv & 3}): " + "  ".join(f"[{int(b[wv, p] - base):5d},{int(e[wv, p] - base):5d})" for p in range(4)))
