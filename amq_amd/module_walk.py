"""Module-walk decode: the drop-in path as the reference harness actually drives it.

``amq_speed_benchmark.py:231-256`` deep-copies an HF Llama and ``setattr``s every linear to a kernel-backed module; a
token step is then HF's forward: a Python walk over 7 quantized ``forward`` calls per block (plus the monkeypatched FT
RMSNorm / single-query attention, ``ftllama_modeling.py:39-46, 127-155``) -- 225 module calls and ~13 launches per block.
``QuantLlama`` (llama.py) instead groups q/k/v and gate/up into single launches and fuses norms / SiLU / residuals into
them (5 launches per block).  This module is the first shape, built from :class:`HIPQuantLinear` modules over the SAME
native weights, so the two can be measured side by side (tools/module_walk_bench.py): what a user gets by only swapping
modules, eager and captured into a hipGraph.
"""
import torch
import torch.nn as nn

from . import _ext, ops
from .quant_linear import HIPLlamaMLP, HIPQuantLinear, HIPRMSNorm


class _RMSNorm(nn.Module):
    def __init__(self, weight, eps):
        super().__init__()
        self.weight, self.eps = weight, eps          # (a plain tensor of the runner: shared, not a Parameter)

    def forward(self, x):
        ext = _ext.get()
        if ext is not None:
            return ext.rmsnorm(x, self.weight, self.eps)
        return ops.rmsnorm(x, self.weight, self.eps)


class _Attention(nn.Module):
    """q/k/v/o projections around the single-query attention kernel (static batch-1 KV cache, position on the device)"""

    def __init__(self, blk, runner):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (_module_of(blk["self_attn." + n]) for n in ("q_proj", "k_proj", "v_proj", "o_proj"))
        self.kc, self.vc, self.r = blk["kc"], blk["vc"], runner

    def forward(self, h):
        r = self.r
        q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
        ext = _ext.get()
        if ext is not None and r.rope_cur is not None and ops.attn_decode_splits(self.kc.shape[2]) == 1:
            return self.o_proj(ext.attn_decode_cur(q, k, v, self.kc, self.vc, r.rope_cur, r.nh, r.nkv))
        att = torch.empty_like(q)
        ops.attn_decode(q, k, v, self.kc, self.vc, att, r.pos, r.nh, r.nkv, r.theta, cur=r.rope_cur)
        return self.o_proj(att)


class _MLP(nn.Module):
    """LlamaMLP as HF writes it (what an unfused swap leaves)"""

    def __init__(self, blk):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = (_module_of(blk["mlp." + n]) for n in ("gate_proj", "up_proj", "down_proj"))
        self.act_fn = nn.SiLU()

    def forward(self, h):
        ext = _ext.get()
        g, u = self.gate_proj(h), self.up_proj(h)
        return self.down_proj(ext.silu_mul(g, u) if ext is not None else ops.silu_mul(g, u))      # act_fn(gate) * up, LlamaMLP


class _Block(nn.Module):
    def __init__(self, blk, runner):
        super().__init__()
        self.input_layernorm = _RMSNorm(blk["ln1"], runner.eps)
        self.post_attention_layernorm = _RMSNorm(blk["ln2"], runner.eps)
        self.self_attn = _Attention(blk, runner)
        self.mlp = _MLP(blk)

    def forward(self, x):
        x = x + self.self_attn(self.input_layernorm(x))
        return x + self.mlp(self.post_attention_layernorm(x))


def _module_of(lin):
    """HIPQuantLinear over a runner's native weights (shared storage, no copy)"""
    mod = HIPQuantLinear(lin.bits, 128, lin.K, lin.N, bias=None)
    mod._set_native(lin.qn, lin.mn, lin.mode)
    return mod


class ModuleWalkLlama(nn.Module):
    """HF-shaped decoder stack over a QuantLlama's weights, caches and step state (``runner`` keeps owning them: prefill
    with the runner, then step with either)."""

    def __init__(self, runner, group_siblings=True, fuse_norms=True, fuse_layers=True):
        super().__init__()
        if getattr(runner, "B", 1) != 1:
            raise ValueError("the module walk mirrors the reference's batch-1 step")
        self.r = runner
        self.layers = nn.ModuleList(_Block(blk, runner) for blk in runner.blocks)
        self.graph = None
        if group_siblings:                      # what prepare_for_inference(backend="hip") does to a swapped model
            from .patching import fuse_llama_mlps, fuse_llama_norms, group_sibling_linears
            group_sibling_linears(self)
            fuse_llama_mlps(self)
            if fuse_norms:
                fuse_llama_norms(self)
            if fuse_layers:
                from .patching import fuse_llama_layers
                fuse_llama_layers(self)

    @torch.inference_mode()
    def _step(self):
        r = self.r
        x = r.x
        for layer in self.layers:
            x = layer(x)
        ops.gemv_f16w(x.reshape(-1), r.lm_head, gamma=r.norm, eps=r.eps, out=r.logits)     # final norm + fp16 lm_head
        ops.decode_tail(r.logits, r.embed, r.token, r.pos, r.x, table=r.rope_tab, cur=r.rope_cur)

    def n_module_calls(self):
        return sum(1 for m in self.modules() if isinstance(m, (HIPQuantLinear, HIPLlamaMLP, HIPRMSNorm, _RMSNorm, _Attention, _MLP, _Block)))

    def capture(self):
        if self.graph is not None:
            return
        r = self.r
        side = torch.cuda.Stream(device=r.dev)
        side.wait_stream(torch.cuda.current_stream(r.dev))
        saved = (r.token.clone(), r.host_pos)
        with torch.cuda.stream(side):
            self._step()
            side.synchronize()
            r.set_pos(saved[1]); r.set_token(saved[0])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                self._step()
        torch.cuda.current_stream(r.dev).wait_stream(side)
        torch.cuda.synchronize(r.dev)
        r.set_pos(saved[1]); r.set_token(saved[0])
        self.graph = g

    def decode_step(self, use_graph=False):
        r = self.r
        if r.host_pos >= r.max_seq:
            raise ValueError(f"decode step at position {r.host_pos} does not fit the KV cache (max_seq={r.max_seq})")
        if use_graph and self.graph is None:
            self.capture()                      # (before the host mirror moves: capture() saves and restores the step state)
        r.host_pos += 1
        if use_graph:
            self.graph.replay()
        else:
            self._step()
