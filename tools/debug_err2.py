import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops, _lib
from amq_amd.hqq_format import random_hqq
from oracle import hqq_ref, linear_ref
dev = torch.device("cuda:0")
bits, n, k = 2, 11008, 4096
h = random_hqq(n, k, bits, seed=7 * bits + 1)
hd = h.to(dev)
qn, mn = ops.repack_from_hqq(hd.W_q, hd.scale.reshape(-1), hd.zero.reshape(-1), bits, n, k)
w_ref = hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), bits, (n, k))
q = hqq_ref.unpack(h.W_q.numpy(), bits, (n, k))
x = torch.randn(1, k, generator=torch.Generator().manual_seed(n + k + 1)).to(torch.float16)
y64 = linear_ref.matmul_f64(x.numpy(), w_ref.T)[0]
y = ops.gemv(x.to(dev), qn, mn, bits, 0, n, k).cpu().numpy().astype(np.float64)[0]
e = np.abs(y - y64) - 2.0 ** -10 * np.abs(y64)
idx = np.argsort(-e)[:5]
for i in idx:
    print("row", i, "y", y[i], "y64", y64[i], "err", y[i] - y64[i], "sum|xw|", np.sum(np.abs(x.numpy()[0].astype(np.float64) * w_ref[i].astype(np.float64))))
eye = torch.eye(k, dtype=torch.float16, device=dev)[:512]
wm = ops.gemm(eye, qn, mn, bits, 0, n, k).cpu().numpy().T
ws = w_ref[:, :512]
bad = np.argwhere(wm.view(np.uint16) != ws.view(np.uint16))
z = h.zero.numpy().reshape(n, -1); s = h.scale.numpy().reshape(n, -1)
for (r, c) in bad[:12]:
    print("w mism row", r, "k", c, "q", q[r, c], "z", float(z[r, c // 128]), "s", float(s[r, c // 128]), "ref", float(ws[r, c]), "mm", float(wm[r, c]))
print("---- isolate")
eye = torch.eye(k, dtype=torch.float16, device=dev)
wm_full = torch.cat([ops.gemm(eye[i:i+512], qn, mn, bits, 0, n, k) for i in range(0, k, 512)]).cpu().numpy().T  # [n,k]
ymm64 = (x.numpy().astype(np.float64) @ wm_full.astype(np.float64).T)[0]
for i in idx:
    print("row", i, "gpu y", y[i], "y64(ref w)", y64[i], "y64(mm w)", ymm64[i])
from amq_amd import _lib
lib = _lib.load()
for name, opt in (("dot", 1), ("mfma", 0)):
    lib.amq_set_option(1, opt)
    yy = ops.gemv(x.to(dev), qn, mn, bits, 0, n, k).cpu().numpy().astype(np.float64)[0]
    print(name, [float(yy[i]) for i in idx])
lib.amq_set_option(1, 0)
xs = x.numpy()[0]
print("subnormal x count", int(((np.abs(xs) < 6.2e-5) & (xs != 0)).sum()), "min |x|", np.abs(xs[xs != 0]).min())
i = idx[0]
contrib = xs.astype(np.float64) * w_ref[i].astype(np.float64)
print("row", i, "largest |contrib|", np.sort(np.abs(contrib))[-3:])
# per-wave partial sums (NW waves, tile g = wave + j*NW) in fp32 vs fp64
print("---- worst weights")
dw = np.abs(wm_full.astype(np.float64) - w_ref.astype(np.float64))
flat = np.argsort(-dw.ravel())[:10]
for f in flat:
    r_, c_ = divmod(int(f), k)
    print("row", r_, "k", c_, "k%128", c_ % 128, "q", q[r_, c_], "z", float(z[r_, c_ // 128]), "s", float(s[r_, c_ // 128]), "ref", float(w_ref[r_, c_]), "mm", float(wm_full[r_, c_]), "diff", dw[r_, c_])
print("mismatch frac full", (wm_full.view(np.uint16) != w_ref.view(np.uint16)).mean())
rowdiff = dw[2374].reshape(-1, 128).sum(1)
print("row 2374 per-group sum|dw|", np.round(rowdiff * 1e6, 2)[:32], "z:", z[2374][:32])
