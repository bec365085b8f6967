#!/usr/bin/env python3
"""Group-scale GEMV arithmetic (AMQ_MATH_GROUPSCALE) against the exact two-rounding arithmetic: (1) distance from the CPU
nn.Linear-on-dequantized-weights result, per bit-width / shape / rows, in units of the parity bar and of rms(y); (2) decode tokens/s
A/B on the 7B avg-3 workload (graph replay, alternating arms).  usage: gs_ab.py [parity|speed|both] [steps]
(imports oracle/: a measuring tool, like bench.py's parity gate -- never on the product path)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import arch, ops                      # noqa: E402
from amq_amd.hqq_format import random_hqq          # noqa: E402
from amq_amd.llama import QuantLlama               # noqa: E402

dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ARMS = {"exact": ops.GemvOpts(math=ops.MATH_EXACT), "groupscale": ops.GemvOpts(math=ops.MATH_GROUPSCALE), "linear": ops.GemvOpts(math=ops.MATH_LINEAR)}


def parity():
    from oracle import cpu_baseline as cb
    rows_out = []
    for (n, k) in ((4096, 4096), (11008, 4096), (4096, 11008)):
        for bits in (2, 3, 4):
            h = random_hqq(n, k, bits, seed=11 + bits)
            w_deq = cb.dequantize_torch(h.W_q, h.scale, h.zero, bits, (n, k))
            hd = h.to(dev)
            qn, mn = ops.repack_from_hqq(hd.W_q, hd.scale.reshape(-1), hd.zero.reshape(-1), bits, n, k)
            for rows in ((1, 5, 8) if k <= 4096 else (1, 5)):
                x = torch.randn(rows, k, generator=torch.Generator().manual_seed(rows)).to(torch.float16)
                y_cpu = torch.nn.functional.linear(x.float(), w_deq.float()).to(torch.float16).float()
                rms = float(y_cpu.pow(2).mean().sqrt())
                rec = {"shape": [n, k], "bits": bits, "rows": rows}
                for arm, o in ARMS.items():
                    y = ops.gemv(x.to(dev), qn, mn, bits, ops.MODE_HQQ, n, k, opts=o).float().cpu()
                    err = (y - y_cpu).abs()
                    rec[arm] = {"max_err_over_bar": round(float((err / (1e-3 * y_cpu.abs() + 1e-3 * rms)).max()), 3),
                                "rms_err_over_rms": round(float(err.pow(2).mean().sqrt()) / rms, 6),
                                "max_err_over_rms": round(float(err.max()) / rms, 6)}
                rows_out.append(rec)
                print(json.dumps(rec), flush=True)
    worst = {arm: max(r[arm]["max_err_over_bar"] for r in rows_out) for arm in ARMS}
    print("WORST max_err_over_bar:", json.dumps(worst), flush=True)


def speed():
    name = os.environ.get("SWEEP_MODEL", "Llama-2-7b-hf")
    cfg = arch.MODEL_CONFIGS[name]
    for label in ("avg-3", "uniform-4"):
        a, usage = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=arch.PINNED_7B if "7b" in name else ())
        linear = a["linear"] if label == "avg-3" else arch.uniform_arch(cfg, 4)["linear"]
        res = {arm: [] for arm in ("exact", "groupscale")}
        for rep in range(3):
            for arm in ("exact", "groupscale"):
                ops.DEFAULT_GEMV_OPTS = ARMS[arm]
                m = QuantLlama(cfg, linear, device=dev, max_seq=64 + steps + 24, seed=0)
                ids = torch.randint(0, m.vocab - 1, (64,), generator=torch.Generator().manual_seed(0)).to(dev)
                m.prefill(ids, use_graph=False)
                m.capture()
                for _ in range(8):
                    m.decode_step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    m.decode_step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                m.check()
                res[arm].append(1 / dt)
                del m
                torch.cuda.empty_cache()
        ops.DEFAULT_GEMV_OPTS = None
        print(name, label, {k: [round(v, 1) for v in vs] for k, vs in res.items()}, flush=True)


if what in ("parity", "both"):
    parity()
if what in ("speed", "both"):
    speed()
