import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops, _lib
from amq_amd.hqq_format import random_hqq
from oracle import hqq_ref, linear_ref
dev = torch.device("cuda:0")
lib = _lib.load()
for bits, n, k in ((2, 11008, 4096), (4, 4096, 4096), (3, 4096, 11008)):
    h = random_hqq(n, k, bits, seed=7 * bits + 1)
    hd = h.to(dev)
    qn, mn = ops.repack_from_hqq(hd.W_q, hd.scale.reshape(-1), hd.zero.reshape(-1), bits, n, k)
    w_ref = hqq_ref.dequantize(h.W_q.numpy(), h.scale.numpy(), h.zero.numpy(), bits, (n, k))
    x = torch.randn(1, k, generator=torch.Generator().manual_seed(n + k + 1)).to(torch.float16)
    y64 = linear_ref.matmul_f64(x.numpy(), w_ref.T)
    rms = np.sqrt(np.mean(y64 ** 2))
    for name, dot in (("mfma", 0), ("dot", 1)):
        lib.amq_set_option(1, dot)
        y = ops.gemv(x.to(dev), qn, mn, bits, 0, n, k).cpu().numpy().astype(np.float64)
        e = np.abs(y - y64)
        ulp = np.abs(np.spacing(y64.astype(np.float16)).astype(np.float64))
        print(bits, n, k, name, "max|e|/rms %.3e" % (e.max() / rms), "max e/ulp %.3f" % (e / ulp).max(), "mean e/ulp %.3f" % (e / ulp).mean())
    lib.amq_set_option(1, 0)
    # weights as the kernels see them
    wd = ops.dequantize(qn, mn, bits, 0, n, k).cpu().numpy()
    print("  dequant kernel exact:", np.array_equal(wd.view(np.uint16), w_ref.view(np.uint16)))
    eye = torch.eye(k, dtype=torch.float16, device=dev)[:256]
    wm = ops.gemm(eye, qn, mn, bits, 0, n, k).cpu().numpy().T   # [n, 256]
    same = wm.view(np.uint16) == w_ref[:, :256].view(np.uint16)
    print("  matmul-weights mismatch frac %.4f" % (1 - same.mean()), "max abs diff %.3e" % np.abs(wm.astype(np.float64) - w_ref[:, :256].astype(np.float64)).max(),
          "scale ~ %.3e" % float(h.scale.float().mean()))
