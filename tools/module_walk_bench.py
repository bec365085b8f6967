#!/usr/bin/env python3
"""Drop-in path vs runner: decode tokens/s of
  (a) QuantLlama            grouped launches, fused norms / SiLU / residuals, hipGraph      (what bench.py reports)
  (b) ModuleWalkLlama eager HF-style walk over HIPQuantLinear.forward calls, one launch per module, host-driven
  (c) ModuleWalkLlama graph the same walk captured into a hipGraph
on the bench workload (Llama-2-7B shapes, avg-3-bit arch), plus the host cost of one HIPQuantLinear.forward.
usage: module_walk_bench.py [steps]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from amq_amd.module_walk import ModuleWalkLlama

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
m, a, usage = bench.build_model(dev, max_seq=64 + 12 * steps + 64)
mw = ModuleWalkLlama(m)                       # q/k/v and gate/up grouped, as prepare_for_inference(backend="hip") leaves a swapped model
mw_plain = ModuleWalkLlama(m, group_siblings=False)
ids = torch.randint(0, m.vocab - 1, (64,), generator=torch.Generator().manual_seed(0)).to(dev)


def run(fn, n):
    m.prefill(ids)                              # every arm starts from the same context (the attention cost grows with the position)
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


m.prefill(ids)
out = {"workload": "Llama-2-7B avg-3-bit (bits_usage %.3f), batch 1 decode after a 64-token prefill" % usage,
       "module_calls_per_token": mw.n_module_calls()}
# same next token from both step implementations (the fused runner rounds some intermediates differently: compare logits)
m.decode_step(use_graph=False)
la, ta = m.logits.float().clone(), int(m.token.item())
m.prefill(ids)
mw.decode_step(use_graph=False)
lb = m.logits.float().clone()
out["logit_distance_runner_vs_walk"] = float((la - lb).abs().max() / la.abs().max())
out["runner_graph_tokens_per_s"] = run(lambda: m.decode_step(True), steps)
out["module_walk_eager_tokens_per_s"] = run(lambda: mw.decode_step(False), steps)
out["module_walk_graph_tokens_per_s"] = run(lambda: mw.decode_step(True), steps)
out["module_walk_ungrouped_eager_tokens_per_s"] = run(lambda: mw_plain.decode_step(False), steps)
out["module_walk_ungrouped_graph_tokens_per_s"] = run(lambda: mw_plain.decode_step(True), steps)
from amq_amd import _ext
out["torch_extension"] = _ext.get() is not None
# host cost of one forward (no device wait in the loop: measures the enqueue path; the queue is drained every 512 calls)
lin = mw.layers[0].self_attn.o_proj
x = torch.randn(1, m.H, device=dev).half()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2048):
    lin(x)
    if i % 512 == 511:
        torch.cuda.synchronize()
out["forward_host_us"] = (time.perf_counter() - t0) / 2048 * 1e6
print(json.dumps(out, indent=1))
