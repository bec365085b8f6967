import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amq_amd import arch, ops
from amq_amd.llama import QuantLlama
cfg = arch.MODEL_CONFIGS["Llama-2-7b-hf"]
a, usage = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=arch.PINNED_7B)
t0 = time.time()
m = QuantLlama(cfg, a["linear"], max_seq=256, seed=0)
torch.cuda.synchronize(); print("build s", time.time() - t0, "bits_usage", usage, "linear GB/token", m.linear_bytes_per_token() / 1e9)
ids = torch.randint(0, 31999, (64,), device="cuda:0")
t0 = time.time(); m.prefill(ids); torch.cuda.synchronize(); print("prefill ms (first)", (time.time() - t0) * 1e3)
t0 = time.time(); m.prefill(ids); torch.cuda.synchronize(); print("prefill ms", (time.time() - t0) * 1e3)
m.capture()
for _ in range(5): m.decode_step()
torch.cuda.synchronize()
K = 100
t0 = time.perf_counter()
for _ in range(K): m.decode_step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("decode tokens/s", K / dt, "ms/token", dt / K * 1e3, "finite logits", bool(torch.isfinite(m.logits.float()).all()))
bytes_tok = m.total_bytes_per_token(64 + 50)
print("GB/s effective", bytes_tok / (dt / K) / 1e9)

# external events inside a graph
try:
    blk = m.blocks[3]
    evs = [(torch.cuda.Event(enable_timing=True, external=True), torch.cuda.Event(enable_timing=True, external=True)) for _ in range(4)]
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    def body():
        H = m.H
        evs[0][0].record(); ops.gemv_grouped(m.x, [blk["self_attn.q_proj"].seg(m.q), blk["self_attn.k_proj"].seg(m.k), blk["self_attn.v_proj"].seg(m.v)], H, prologue=ops.PRO_RMSNORM, gamma=blk["ln1"], eps=1e-5); evs[0][1].record()
        evs[1][0].record(); ops.gemv_grouped(m.att, [blk["self_attn.o_proj"].seg(m.x, residual=m.x)], H); evs[1][1].record()
        evs[2][0].record(); ops.gemv_grouped(m.x, [blk["mlp.gate_proj"].seg(m.gate), blk["mlp.up_proj"].seg(m.up)], H, prologue=ops.PRO_RMSNORM, gamma=blk["ln2"], eps=1e-5); evs[2][1].record()
        evs[3][0].record(); ops.gemv_grouped(m.gate, [blk["mlp.down_proj"].seg(m.x, residual=m.x)], m.I, prologue=ops.PRO_SILU_MUL, x2=m.up); evs[3][1].record()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            body()
    torch.cuda.synchronize()
    for rep in range(3):
        g.replay(); torch.cuda.synchronize()
        print("graph external-event us:", [round(a.elapsed_time(b) * 1e3, 2) for a, b in evs])
except Exception as ex:
    print("external events in graph failed:", repr(ex))
