#!/usr/bin/env python3
"""amq_speed_benchmark.py for the HIP backend: same flags, same metric definitions, same result JSON
(amq/amq_speed_benchmark.py:99-293):

    python -m amq_amd.speed_benchmark --model_name Llama-2-7b-hf --tps --gemv --peak_memory \
        --target_bits 3 --arch_path iter_200.stats --file_name out.json

    {"fp16": {"tps": {"1.64.128": ...}, ...}, "<bits>bit": {...}, "args": {...}}

``--save_path`` works as in the reference: the quantized row is assembled from the HQQ checkpoint directories
``{save_path}/{model_name}_{n}bit_128gs_1axis`` (``config.json`` + ``qmodel.pt`` as written by
``AutoHQQHFModel.save_quantized``; amq_speed_benchmark.py:129-131, 231-251) -- only the bit-widths the arch uses have to
exist, and the model shape is taken from their ``config.json``.  Without it (no checkpoints ship with the reference and
there is no network) the weights are synthetic payloads of the model's real layer shapes.  The fp16 row is always
synthetic (the reference loads the HF fp16 checkpoint from ``--model_path``).  ``--use_ft`` is
accepted and ignored (there is one attention path), and without ``--arch_path`` the arch is uniform
``--target_bits`` in {2,3,4} exactly as in the reference; ``--synthesize_arch`` draws an arch at
``--target_bits`` with the SearchSpace.sample recipe when no searched ``.stats`` file is at hand.
"""
import argparse
import json
import os

import torch

from . import arch as arch_mod
from . import checkpoint
from .llama import DenseLlama, QuantLlama, get_memory_footprint
from .speed import benchmark_speed, cleanup


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--model_path", type=str, default="meta-llama")
    p.add_argument("--model_name", type=str, default="Llama-2-7b-hf")
    p.add_argument("--save_path", type=str, default=None)
    p.add_argument("--use_ft", action="store_true")
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--seq_length", type=int, default=64)
    p.add_argument("--gen_length", type=int, default=128)
    p.add_argument("--tps", action="store_true")
    p.add_argument("--gemm", action="store_true")
    p.add_argument("--gemv", action="store_true")
    p.add_argument("--ttft", action="store_true")
    p.add_argument("--memory", action="store_true")
    p.add_argument("--peak_memory", action="store_true")
    p.add_argument("--target_bits", type=float, default=4)
    p.add_argument("--arch_path", type=str, default=None)
    p.add_argument("--file_name", type=str, default=None)
    p.add_argument("--synthesize_arch", action="store_true", help="(extension) draw an arch at --target_bits")
    p.add_argument("--skip_fp16", action="store_true", help="(extension) do not run the fp16 baseline row")
    p.add_argument("--tokenizer", type=str, default="auto",
                   help="(extension) TTFT tokenizer: a tokenizer.json path, 'synthetic' (one word per vocabulary id), 'none' (ids "
                        "used directly), or 'auto' = tokenizer.json of the checkpoint directory if present, else synthetic")
    args = p.parse_args(argv)

    def ckpt_dir(bits):
        return os.path.join(args.save_path, f"{args.model_name}_{bits}bit_128gs_1axis")     # amq_speed_benchmark.py:129-131

    if args.save_path:
        have = [b for b in (2, 3, 4) if os.path.isdir(ckpt_dir(b))]
        if not have:
            raise FileNotFoundError(f"no HQQ checkpoint directory {ckpt_dir('{2,3,4}')} found")
        with open(os.path.join(ckpt_dir(have[0]), "config.json")) as f:
            cfg = checkpoint.runner_config(json.load(f))
    elif args.model_name in arch_mod.MODEL_CONFIGS:
        cfg = arch_mod.MODEL_CONFIGS[args.model_name]
    else:
        raise SystemExit(f"unknown model {args.model_name}; known: {sorted(arch_mod.MODEL_CONFIGS)} (or pass --save_path)")
    tokenizer = None
    if args.ttft and args.tokenizer != "none":
        from .speed import SyntheticTokenizer
        path = args.tokenizer if args.tokenizer not in ("auto", "synthetic") else None
        if path is None and args.tokenizer == "auto" and args.save_path:
            cand = os.path.join(ckpt_dir(have[0]), "tokenizer.json")
            path = cand if os.path.exists(cand) else None
        tokenizer = SyntheticTokenizer.from_file(path) if path else SyntheticTokenizer(cfg["vocab_size"])
    sizes = [args.batch_size, args.seq_length, args.gen_length]
    gemm_iteration = 20
    gemv_iteration = 5 if args.gen_length < 1024 else 2          # amq_speed_benchmark.py:168-169
    max_seq = args.seq_length + args.gen_length + 8
    # token modes at batch_size 2 .. 8 run a runner built for that batch (one step = the same launches with batch_size rows); larger
    # batches are served in GeMM mode only, which needs no cache
    run_batch = args.batch_size if 1 <= args.batch_size <= 8 and (args.tps or args.gemv) else 1
    result = {}

    def run(model, row):
        result[row] = {}
        if args.tps:
            r = benchmark_speed(model, None, use_ft=args.use_ft, iteration=gemv_iteration, sizes=sizes, mode="TPS",
                                get_peak_memory=args.peak_memory)
            result[row].update(r); print("Token per second : ", r)
        if args.gemm:
            r = benchmark_speed(model, None, use_ft=args.use_ft, iteration=gemm_iteration, sizes=sizes, mode="GeMM", get_peak_memory=False)
            result[row].update(r); print("GeMM : ", r)
        if args.gemv:
            r = benchmark_speed(model, None, use_ft=args.use_ft, iteration=gemv_iteration, sizes=sizes, mode="GeMV", get_peak_memory=False)
            result[row].update(r); print("GeMV : ", r)
        if args.ttft and args.batch_size != 1:
            print("TTFT : skipped (a batch-1 measurement)")
        elif args.ttft:
            r = benchmark_speed(model, tokenizer, use_ft=args.use_ft, iteration=gemm_iteration, sizes=sizes, mode="TTFT", get_peak_memory=False)
            result[row].update(r); print("TTFT : ", r)
        if args.memory:
            mem = get_memory_footprint(model) / 1024 ** 3
            result[row].update({"memory": mem}); print(f"Memory : {mem} GB")

    if not args.skip_fp16:
        print("Get Speed of original model...")
        base = DenseLlama(cfg, max_seq=max_seq, batch=run_batch)
        run(base, "fp16")
        del base
        cleanup()

    if args.arch_path is not None:
        if not os.path.exists(args.arch_path):
            raise FileNotFoundError(f"Arch file {args.arch_path} not found")
        linear = arch_mod.select_arch(args.arch_path, args.target_bits)
    elif args.synthesize_arch:
        pinned = arch_mod.PINNED_7B if args.model_name == "Llama-2-7b-hf" else ()
        linear = arch_mod.synthesize_arch(cfg, args.target_bits, seed=0, pinned=pinned)[0]["linear"]
    else:
        assert args.target_bits in [2, 3, 4], "target bits should be 2, 3, 4 if arch_path is not provided"
        linear = arch_mod.uniform_arch(cfg, int(args.target_bits))["linear"]

    print(f"Get Speed of {args.target_bits}bit model...")
    if args.save_path:
        used = sorted({int(b) for v in linear.values() for b in v})
        missing = [ckpt_dir(b) for b in used if not os.path.isdir(ckpt_dir(b))]
        if missing:
            raise FileNotFoundError(f"the arch needs HQQ checkpoints that are not there: {missing}")
        model = checkpoint.load_mixed({b: ckpt_dir(b) for b in used}, linear, max_seq=max_seq, batch=run_batch)
    else:
        model = QuantLlama(cfg, linear, max_seq=max_seq, batch=run_batch)
    run(model, f"{args.target_bits}bit")
    del model
    cleanup()

    result.update({"args": vars(args)})
    if args.file_name:
        out_path = os.path.join("benchmark/outputs", args.file_name)      # amq_speed_benchmark.py:290-293
        os.makedirs(os.path.dirname(out_path), exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(result, f, indent=4)
    print(json.dumps(result))
    return result


if __name__ == "__main__":
    main()
