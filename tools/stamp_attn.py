#!/usr/bin/env python3
"""Phase timeline of the decode attention kernel (diagnostic build -DAMQ_STAMP).  usage: python tools/with_variant.py stamp tools/stamp_attn.py [pos]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import _lib, ops

pos = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
_lib.load()
setst = ctypes.CDLL(_lib.LIB_PATH).amq_debug_set_stamps
setst.argtypes = [ctypes.c_void_p]
nh = nkv = 32
max_seq = 512
nlayer = 8
kcs = [torch.randn(1, nkv, max_seq, 128, device=dev).half() for _ in range(nlayer)]
vcs = [torch.randn(1, nkv, max_seq, 128, device=dev).half() for _ in range(nlayer)]
q = torch.randn(1, nh * 128, device=dev).half(); k = torch.randn(1, nkv * 128, device=dev).half(); v = torch.randn(1, nkv * 128, device=dev).half()
out = torch.zeros(1, nh * 128, dtype=torch.float16, device=dev)
tab = ops.rope_table(max_seq, 10000.0, dev)
cur, posd, _err = ops.new_step_state(dev)              # the graph-replay form: cos/sin row + position in one block
cur.copy_(tab.view(max_seq, 128)[pos]); posd.fill_(pos)
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
stamps = torch.zeros(nlayer, nh, 16, dtype=torch.int64, device=dev)
for rep in range(3):
    junk.fill_(rep)                                  # push the caches out of L2 / Infinity Cache
    for l in range(nlayer):
        setst(ctypes.c_void_p(stamps[l].data_ptr()))
        ops.attn_decode(q, k, v, kcs[l], vcs[l], out, posd, nh, nkv, cur=cur)
        setst(None)
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.float64)
names = ["entry->pos", "pos->rope barrier", "scores", "softmax", "PV+barrier", "reduce+store"]
for l in range(nlayer):
    s = st[l]
    t0 = s[:, 0].min()
    d = np.diff(s[:, :7], axis=1) / 100.0
    print("layer %d: entry skew %.2f us; " % (l, (s[:, 0].max() - t0) / 100) + "  ".join("%s %.2f" % (n, np.median(d[:, i])) for i, n in enumerate(names)) + "  | span %.2f us" % ((s[:, 6].max() - t0) / 100))
