#!/usr/bin/env python3
"""The reference driver's own object through the reference-compatible harness, at size: an HF ``LlamaForCausalLM`` with Llama-2-7B
shapes whose 224 decoder linears are HIPQuantLinear modules (synthetic avg-3-bit payloads), prepared by ``prepare_for_inference`` and
handed to ``benchmark_speed`` as amq_speed_benchmark.py:253-277 does -- once with ``use_ft=True`` (the fused hipGraph runner built over
the model's buffers, QuantLlama.from_hf) and once with ``use_ft=False`` (HF's own forward / generate over the fused modules: the
reference's non-FT loops, amq/utils/speed.py:22-46, 93-125).
usage: speed_harness_hf.py [out.json] [--gen 128] [--seq 64]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?", default=None)
ap.add_argument("--seq", type=int, default=64)
ap.add_argument("--gen", type=int, default=128)
ap.add_argument("--model", default="Llama-2-7b-hf")
args = ap.parse_args()

from transformers import LlamaConfig, LlamaForCausalLM
from amq_amd import arch, ops
from amq_amd.llama import _synthetic_linear
from amq_amd.patching import prepare_for_inference
from amq_amd.quant_linear import HIPQuantLinear
from amq_amd.speed import SyntheticTokenizer, benchmark_speed

dev = torch.device("cuda:0")
cfg = arch.MODEL_CONFIGS[args.model]
a, usage = arch.synthesize_arch(cfg, 3.0, seed=0, pinned=arch.PINNED_7B if "7b" in args.model else ())
H, I, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["n_block"]
hf_cfg = LlamaConfig(hidden_size=H, intermediate_size=I, num_hidden_layers=L, num_attention_heads=cfg["num_heads"],
                     num_key_value_heads=cfg["num_kv_heads"], vocab_size=cfg["vocab_size"], max_position_embeddings=4096,
                     rms_norm_eps=1e-5, attn_implementation="sdpa")
t0 = time.time()
old = torch.get_default_dtype()
torch.set_default_dtype(torch.float16)
with torch.device("meta"):
    model = LlamaForCausalLM(hf_cfg)
torch.set_default_dtype(old)
model.to_empty(device=dev)                                                         # (before the swap: to_empty would wipe the payloads too)
gen = torch.Generator(device=dev).manual_seed(0)
for li, layer in enumerate(model.model.layers):                                    # the decoder linears: HIPQuantLinear over synthetic payloads
    for name in cfg["linear"]:
        mod_name, lin_name = name.split(".")
        n, k = cfg["linear_shape"][name]
        bits = arch.arch_bits(a["linear"], name, li)
        s = _synthetic_linear(n, k, bits, gen, dev)
        q = HIPQuantLinear(bits, 128, k, n, bias=False, name=lin_name)
        q._set_native(s.qn, s.mn, ops.MODE_HQQ)
        setattr(getattr(layer, mod_name), lin_name, q)
with torch.no_grad():                                                              # embeddings, norms, lm_head: random fp16
    for n_, p in model.named_parameters():
        if "norm" in n_:
            p.fill_(1.0)
        else:
            p.copy_((torch.randn(p.shape, device=dev, generator=gen) * 0.02).to(p.dtype))
    for n_, b_ in model.named_buffers():
        if "inv_freq" in n_:                                                       # (rotary table: recomputed, to_empty left it undefined)
            dim = H // cfg["num_heads"]
            b_.copy_(1.0 / (10000.0 ** (torch.arange(0, dim, 2, device=dev).float() / dim)))
model = model.eval()
prepare_for_inference(model, backend="hip")
print(f"built {args.model} HF model with {sum(isinstance(m, HIPQuantLinear) for m in model.modules())} HIPQuantLinear modules "
      f"(bits_usage {usage:.3f}) in {time.time() - t0:.1f} s", flush=True)
tok = SyntheticTokenizer(cfg["vocab_size"])
sizes = [1, args.seq, args.gen]
result = {}

# ---- the reference's own loops over the HF OBJECT (amq/utils/speed.py:22-46 TPS, 50-127 GeMV / GeMM with wrapped_ft, 186-239 TTFT): the caller's
# model.generate(...) and model(ids, start_pos=..., use_cache=False), every timed region bracketed by synchronize + perf_counter, medians as there
import gc
import numpy as np


def _cleanup():
    torch.cuda.empty_cache()
    gc.collect()


@torch.inference_mode()
def hf_object_rates(model, seq, gen, it_tok=5, it_gemm=20):
    ids = torch.randint(0, cfg["vocab_size"] - 1, (1, seq), dtype=torch.long).to(dev)
    mask = torch.ones_like(ids)
    model.generation_config.pad_token_id = model.generation_config.eos_token_id
    row = {}
    # TPS: gen / median(time of generate)
    ts = []
    for i in range(it_tok + 1):
        _cleanup()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _ = model.generate(ids, min_new_tokens=gen, max_new_tokens=gen, do_sample=False, num_beams=1, attention_mask=mask)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    row["tps"] = {f"1.{seq}.{gen}": float(gen / np.median(ts[1:]))}              # (first pass: runner build + graph capture, as the reference's first pass pays its JIT)
    # GeMM: 1 / median(prompt forward); GeMV: 1 / median(one-token forward), each token timed on its own
    tm, tv = [], []
    for i in range(it_gemm):
        _cleanup()
        start_pos = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model(ids, start_pos=start_pos, use_cache=False)
        torch.cuda.synchronize(); tm.append(time.perf_counter() - t0)
        if i >= it_tok:
            continue
        start_pos += out.logits.shape[1]
        nxt = torch.as_tensor([[out.logits[:, -1].max(1)[1].unsqueeze(1)]], device=dev)
        for _ in range(gen):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = model(nxt, start_pos=start_pos, use_cache=False)
            torch.cuda.synchronize(); tv.append(time.perf_counter() - t0)
            start_pos += out.logits.shape[1]
            nxt = torch.as_tensor([[out.logits[:, -1].max(1)[1].unsqueeze(1)]], device=dev)
    row["gemm"] = {f"1.{seq}.{gen}": float(1 / np.median(tm[1:]))}
    row["gemv"] = {f"1.{seq}.{gen}": float(1 / np.median(tv))}
    # TTFT: tokenizer encode + prompt forward + arg-max + tokenizer decode, ms
    text = tok.decode(ids[0])
    tt = []
    for _ in range(it_gemm):
        _cleanup()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        i2 = tok(text, return_tensors="pt", truncation=True, max_length=seq).input_ids.to(dev)
        out = model(i2, start_pos=0, use_cache=False)
        first = out.logits[:, -1].max(1)[1].unsqueeze(1)
        _ = tok.decode(first[0])
        torch.cuda.synchronize(); tt.append((time.perf_counter() - t0) * 1000)
    row["ttft"] = {f"1.{seq}.{gen}": float(np.median(tt[1:]))}
    return row


from amq_amd.hf_fast import convert_model_to_hip, revert_model_to_hf
convert_model_to_hip(model)                                                        # convert_model_to_ft + replace_generate_functions of the reference's driver
result["3.0bit HF object, converted"] = hf_object_rates(model, args.seq, args.gen)
print(result["3.0bit HF object, converted"], flush=True)
revert_model_to_hf(model)
for use_ft in (True, False):
    row = result[f"3.0bit use_ft={'true' if use_ft else 'false'}"] = {}
    for mode, it in (("TPS", 5), ("GeMM", 20), ("GeMV", 5), ("TTFT", 20)):
        r = benchmark_speed(model, tok if mode == "TTFT" else None, use_ft=use_ft, iteration=it if use_ft else max(2, it // 4), sizes=sizes,
                            mode=mode, get_peak_memory=False)
        row.update(r)
        print(row, flush=True)
result["args"] = {"model_name": args.model, "seq_length": args.seq, "gen_length": args.gen, "batch_size": 1, "bits_usage": usage,
                  "object": "transformers LlamaForCausalLM with HIPQuantLinear decoder linears after prepare_for_inference(backend='hip')",
                  "HF object, converted": "hf_fast.convert_model_to_hip(model): the reference's own loops -- model.generate(...), model(ids, start_pos=, use_cache=False) -- on the HF object",
                  "use_ft=true": "QuantLlama.from_hf: hipGraph token step over the model's own buffers",
                  "use_ft=false": "HF's forward / generate() over the fused modules (eager, DynamicCache, sdpa attention)"}
print(json.dumps(result))
if args.out:
    json.dump(result, open(args.out, "w"), indent=1)
