#!/usr/bin/env python3
"""amq_attn_prefill_f16 against the framework's SDPA (AOTriton flash attention) on prompt-pass shapes: us and TFLOP/s
(causal: 2 * S^2 * 128 * heads * batch flop for both products together), one process, interleaved rounds."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import ops

dev = torch.device("cuda:0")
for name, (B, S, nh, nkv) in {"7B 1x64": (1, 64, 32, 32), "7B 1x512": (1, 512, 32, 32), "7B 1x2048": (1, 2048, 32, 32),
                             "13B 16x2048 (configs[3])": (16, 2048, 40, 40), "70B 1x2048 (GQA 64/8)": (1, 2048, 64, 8)}.items():
    H, KV = nh * 128, nkv * 128
    q = torch.randn(B * S, H, device=dev).half(); k = torch.randn(B * S, KV, device=dev).half(); v = torch.randn(B * S, KV, device=dev).half()
    out = torch.empty_like(q)

    def ours():
        ops.attn_prefill(q, k, v, out, S, nh, nkv, batch=B)

    def sdpa():
        qh = q.view(B, S, nh, 128).transpose(1, 2)
        kh, vh = k.view(B, S, nkv, 128).transpose(1, 2), v.view(B, S, nkv, 128).transpose(1, 2)
        if nkv != nh:
            kh, vh = kh.repeat_interleave(nh // nkv, dim=1), vh.repeat_interleave(nh // nkv, dim=1)
        return torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, is_causal=True).transpose(1, 2).reshape(B * S, H).contiguous()

    ref = sdpa(); ours()
    err = (out.float() - ref.float()).abs().max().item()
    t = {"ours": [], "sdpa": []}
    for _ in range(5):
        for nm, fn in (("ours", ours), ("sdpa", sdpa)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            t[nm].append(e0.elapsed_time(e1) * 1e3 / 5)
    fl = 2.0 * S * S * 128 * nh * B
    r = {"case": name, "max_abs_diff_vs_sdpa": round(err, 5)}
    for nm in t:
        us = sorted(t[nm])[2]
        r[nm + "_us"] = round(us, 1); r[nm + "_TFLOPs"] = round(fl / us / 1e6, 1)
    print(json.dumps(r), flush=True)
