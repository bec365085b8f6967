import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops
from amq_amd.llama import _synthetic_linear
dev = torch.device("cuda:0")
for bits, n, k in ((3, 4096, 11008), (3, 4096, 4096), (4, 4096, 11008), (3, 5120, 13824)):
    l = _synthetic_linear(n, k, bits, torch.Generator(device=dev).manual_seed(n + k + bits), dev)
    x = torch.randn(1, k, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).half()
    ys = [ops.gemv(x, l.qn, l.mn, bits, 0, n, k).clone() for _ in range(4)]
    print(bits, n, k, "deterministic:", all(torch.equal(ys[0], y) for y in ys))
    y2 = ops.gemv(x * 2, l.qn, l.mn, bits, 0, n, k)
    d = (y2.float() - 2 * ys[0].float()).abs()
    print("   homog mismatches:", int((d > 0).sum()), "max", float(d.max()), "x absmax", float(x.abs().max()), "y absmax", float(ys[0].abs().max()))
    bad = (d > 0).nonzero()[:5]
    for b in bad:
        i = int(b[1]); print("    idx", i, float(ys[0][0, i]), float(y2[0, i]))
