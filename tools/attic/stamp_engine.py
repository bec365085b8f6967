#!/usr/bin/env python3
"""Per-workgroup phase timeline of the decode engine (diagnostic build -DAMQ_ENG_STAMP):
    make -C amq_amd/csrc abvariant TAG=engstamp EXTRA=-DAMQ_ENG_STAMP      (the engine is an A/B route: libraries that carry it are built with abvariant)
    python tools/with_variant.py engstamp tools/attic/stamp_engine.py [block]
Stamps are the 100 MHz realtime counter (10 ns), thread 0 of every workgroup, for ONE decoder block of a graph-less step of
the bench workload; printed: per stage, median / max over workgroups of each phase, relative to the stage's barrier release."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from amq_amd import _lib

dev = torch.device("cuda:0")
block = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m, a, usage = bench.build_model(dev, seed=0, max_seq=64 + 64, engine=True)
lib = _lib.load()
P = torch.cuda.get_device_properties(dev).multi_processor_count
st = torch.zeros(P * 64, dtype=torch.int64, device=dev)
lib_raw = ctypes.CDLL(_lib.LIB_PATH)
lib_raw.amq_debug_engine_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib_raw.amq_debug_engine_stamps(ctypes.c_void_p(st.data_ptr()), block)
ids = torch.randint(0, m.vocab - 1, (64,), generator=torch.Generator().manual_seed(0)).to(dev)
m.prefill(ids)
for _ in range(6):
    m.decode_step(use_graph=False)
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(P, 64).astype(np.float64) * 0.01      # us
names = ["qkv", "o", "gate/up", "down"]
t0 = s[:, 1 + 0].min()
print("block %d, us; per stage: [release spread] | x staged | tiles done (wave 0) | partials barrier | stores drained   (median / max over workgroups, from the stage's earliest release)" % block)
prev_end = None
for k in range(4):
    base = 1 + 8 * k
    rel = s[:, base + 0]
    r0 = rel.min()
    line = "%-8s release spread %.2f" % (names[k], rel.max() - r0)
    for j, nm in ((1, "x"), (2, "tiles"), (3, "bar"), (4, "drained")):
        v = s[:, base + j] - r0
        line += " | %s %.2f / %.2f" % (nm, np.median(v), v.max())
    if k == 1:
        att0, att1 = s[:, base + 5], s[:, base + 6]
        nh = m.nh
        line += "   [attention: start %.2f (after qkv's earliest drain %.2f), body %.2f / %.2f on %d wgs, release after last body end %.2f]" % (
            att0.min() - t0, att0.min() - prev_end, np.median((att1 - att0)[:nh]), (att1 - att0)[:nh].max(), nh, r0 - att1[:nh].max())
    if prev_end is not None and k != 1:
        line += "   [sync: last drain -> earliest release %.2f]" % (r0 - prev_last)
    print(line)
    prev_end = s[:, base + 4].min()
    prev_last = s[:, base + 4].max()
    print("         stage span (earliest release -> last drain) %.2f" % (s[:, base + 4].max() - r0))
print("whole block: %.2f us" % (s[:, 1 + 8 * 3 + 4].max() - t0))
c = st.cpu().numpy().reshape(P, 64)[:, 40:47].astype(np.float64)
nt = np.maximum(c[:, 4], 1)
if c[:, 4].max() > 0:          # (only builds with -DAMQ_ENG_CYCLES fill these; the s_memtime pairs perturb the loop by 10-20 %)
  print("wave 0 of every workgroup, whole kernel, shader-clock cycles per tile (median over workgroups): ring wait %.0f | LDS reads %.0f | issue %.0f (of which run changes %.0f) | dequant + MFMA %.0f | tiles %d | kernel cycles / tile %.0f"
      % (np.median(c[:, 0] / nt), np.median(c[:, 1] / nt), np.median(c[:, 2] / nt), np.median(c[:, 5] / nt), np.median(c[:, 3] / nt), int(np.median(c[:, 4])), np.median(c[:, 6] / nt)))
