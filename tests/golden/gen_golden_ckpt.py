#!/usr/bin/env python3
"""Golden HQQ checkpoints + reference logits, produced by the REAL reference on CPU.

Build container only (needs /root/reference and transformers).  Drives the reference's own pipeline
(amq/amq_quantization_proxy.py:22-42): AutoHQQHFModel.quantize_model(nbits, group_size=128, axis=1) ->
save_quantized, for nbits in {2,3,4}, on a tiny random LlamaConfig (head_dim 128).  Then assembles a
mixed-precision model the way amq_speed_benchmark.py:231-251 does (setattr of the per-bit layers according to an
arch) -- with the HQQLinear layers themselves, whose CPU forward is matmul(x, dequantize().T) -- and records its
logits for a fixed prompt.  Output (data only): tests/golden/ckpt/{2,3,4}bit/{qmodel.pt,config.json},
tests/golden/ckpt/expected.npz (ids, arch, logits of the last prompt token and of 4 greedy steps).
"""
import json
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference/amq/kernel/hqq"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stubs():
    d = tempfile.mkdtemp(prefix="amq_stubs_")
    os.makedirs(f"{d}/termcolor")
    with open(f"{d}/termcolor/__init__.py", "w") as f:
        f.write("def colored(s,*a,**k): return s\n")
    os.makedirs(f"{d}/faster_transformer")
    with open(f"{d}/faster_transformer/__init__.py", "w") as f:
        f.write("def gemv_4bit(*a,**k): raise NotImplementedError\ndef gemm_4bit(*a,**k): raise NotImplementedError\n")
    return d


def main():
    sys.dont_write_bytecode = True
    sys.path[:0] = [_stubs(), REF]
    import copy
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM
    from hqq.core.quantize import BaseQuantizeConfig
    from hqq.models.hf.base import AutoHQQHFModel

    torch.manual_seed(7)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=2, vocab_size=256, max_position_embeddings=128, rms_norm_eps=1e-5,
                      rope_theta=10000.0, tie_word_embeddings=False, attention_bias=False, mlp_bias=False)
    base = LlamaForCausalLM(cfg).half().eval()
    out_dir = os.path.join(HERE, "ckpt")
    os.makedirs(out_dir, exist_ok=True)
    models = {}
    for bits in (2, 3, 4):
        m = copy.deepcopy(base)
        qc = BaseQuantizeConfig(nbits=bits, group_size=128)          # axis defaults to 1 (quantize.py:1083)
        AutoHQQHFModel.quantize_model(m, quant_config=qc, compute_dtype=torch.float16, device="cpu")
        d = os.path.join(out_dir, f"{bits}bit")
        AutoHQQHFModel.save_quantized(m, d)
        for fn in os.listdir(d):                                      # keep only what from_quantized reads
            if fn not in ("qmodel.pt", "config.json"):
                os.remove(os.path.join(d, fn))
        models[bits] = m
    # mixed arch, assembled like amq_speed_benchmark.py:231-251
    linears = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
               "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]
    arch = {"self_attn.q_proj": [4, 2], "self_attn.k_proj": [3, 3], "self_attn.v_proj": [4, 4], "self_attn.o_proj": [2, 3],
            "mlp.gate_proj": [3, 2], "mlp.up_proj": [2, 4], "mlp.down_proj": [4, 3]}
    mixed = copy.deepcopy(models[4])
    for i in range(2):
        for name in linears:
            mod, lin = name.split(".")
            src = getattr(getattr(models[arch[name][i]].model.layers[i], mod), lin)
            setattr(getattr(mixed.model.layers[i], mod), lin, src)
    ids = torch.randint(0, 256, (1, 9), generator=torch.Generator().manual_seed(3))
    logits = []
    with torch.no_grad():
        cur = ids
        for _ in range(5):
            lg = mixed(cur).logits[0, -1].float()
            logits.append(lg.numpy())
            cur = torch.cat([cur, lg.argmax().reshape(1, 1)], dim=1)
    np.savez_compressed(os.path.join(out_dir, "expected.npz"), ids=ids.numpy()[0], logits=np.stack(logits),
                        tokens=cur.numpy()[0, 9:], arch=json.dumps(arch))
    print("wrote", out_dir, "greedy tokens", cur.numpy()[0, 9:])


if __name__ == "__main__":
    main()
