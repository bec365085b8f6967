"""few-row GEMM launches over a layer quantized with groups of 128 / 64: what ops.gemm runs (auto) against dequantize-once forced (profiles/r04_fine_groups.txt)"""
import sys, os, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import ops
from amq_amd.hqq_format import random_hqq
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (n, k) in ((4096, 4096), (11008, 4096), (4096, 11008)):
    for group in (128, 64):
        h = random_hqq(n, k, 3, seed=1, group=group).to(dev)
        qn, mn = ops.repack_from_hqq(h.W_q, h.scale.reshape(-1), h.zero.reshape(-1), 3, n, k, group=group)
        for m in (24, 64, 128):
            x = torch.randn(m, k, device=dev).half()
            y = torch.empty(m, n, device=dev, dtype=torch.float16)
            row = {"N": n, "K": k, "group": group, "M": m, "auto_us": round(t(lambda: ops.gemm(x, qn, mn, 3, ops.MODE_HQQ, n, k, out=y)), 1)}
            if group != 128:
                row["deq_us"] = round(t(lambda: ops.gemm(x, qn, mn, 3, ops.MODE_HQQ, n, k, out=y, route=ops.GEMM_DEQ)), 1)
                a = ops.gemm(x, qn, mn, 3, ops.MODE_HQQ, n, k); b = ops.gemm(x, qn, mn, 3, ops.MODE_HQQ, n, k, route=ops.GEMM_DEQ)
                row["max_diff_over_rms"] = round(((a.float() - b.float()).abs().max() / b.float().pow(2).mean().sqrt()).item(), 5)
            print(json.dumps(row), flush=True)
