"""benchmark_speed -- the reference's timing harness (amq/utils/speed.py:15-255) for QuantLlama.

Same metric definitions and result keys:
  TPS   gen_len / median(t_generate); t_generate covers prefill + gen_len greedy tokens,
        synchronize-bracketed perf_counter                          (speed.py:22-46)
  GeMV  1 / median(per-token forward latency) after one un-timed prefill; every token is
        timed individually with a device sync on both sides         (speed.py:50-127)
  GeMM  1 / median(prefill forward latency)                          (speed.py:61-71)
  TTFT  median ms of tokenizer encode + prefill + argmax + tokenizer decode of the first token (speed.py:186-239).
        With no tokenizer (None) the ids are used directly; SyntheticTokenizer is the stand-in when no Llama tokenizer
        files exist (offline): a `tokenizers` WordLevel model over one word per vocabulary id, same call surface.
Returns ``{mode: {'B.S.G': value}}`` (+ ``'peak_memory'``) like the reference.
The reference's static KV cache is batch-1 (ftllama_modeling.py:61-68); so is this runner.
"""
import gc
import time

import numpy as np
import torch


def cleanup():
    torch.cuda.empty_cache()
    gc.collect()


@torch.inference_mode()
def device_warmup(device):
    """speed.py:15-19"""
    w = torch.randn((4096, 4096)).to(device)
    for _ in range(100):
        torch.mm(w, w)


@torch.inference_mode()
def benchmark_tps(model, input_ids, gen_seq_len, iteration):
    times = []
    for _ in range(iteration):
        cleanup()
        model.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(input_ids, gen_seq_len)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return gen_seq_len / np.median(times)


@torch.inference_mode()
def benchmark_gemv_gemm(model, input_ids, gen_seq_len, iteration, mode="gemv"):
    times = []
    for _ in range(iteration):
        cleanup()
        model.reset()
        if mode.lower() == "gemm":
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        model.prefill(input_ids)
        if mode.lower() == "gemm":
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        if mode.lower() == "gemv":
            for _ in range(gen_seq_len):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.decode_step()
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
    return 1 / np.median(times)


@torch.inference_mode()
def _benchmark_gemm_batch(model, sizes, iteration, get_peak_memory):
    """GeMM mode at batch_size > 1 (speed.py:61-71, 95-105): 1 / median latency of the batched prompt pass."""
    batch_size, input_seq_len, gen_seq_len = sizes
    device = model.dev
    data = {"gemm": {}}
    if get_peak_memory:
        cleanup()
        torch.cuda.reset_peak_memory_stats(device=device)
        data["peak_memory"] = {}
    input_ids = torch.randint(0, model.vocab - 1, (batch_size, input_seq_len), dtype=torch.long).to(device)
    device_warmup(device)
    model.prefill_batch(input_ids)                       # allocator warm-up
    times = []
    for _ in range(iteration):
        cleanup()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.prefill_batch(input_ids)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    key = f"{batch_size}.{input_seq_len}.{gen_seq_len}"
    data["gemm"][key] = float(1 / np.median(times))
    if get_peak_memory:
        data["peak_memory"][key] = torch.cuda.max_memory_allocated(device=device) / 1024 ** 3
    torch.cuda.reset_peak_memory_stats(device=device)
    cleanup()
    return data


class _Encoding:
    def __init__(self, ids):
        self.input_ids = ids


class SyntheticTokenizer:
    """The slice of the HF tokenizer surface the reference's TTFT loop uses -- ``tok(text, return_tensors='pt',
    truncation=True, max_length=n).input_ids`` and ``tok.decode(ids)`` (speed.py:193, 214-217) -- over a `tokenizers`
    model.  ``SyntheticTokenizer(vocab)``: WordLevel, one word ("t<id>") per vocabulary id, whitespace pre-tokenizer -- what
    stands in when no Llama tokenizer files are available; ``SyntheticTokenizer.from_file(path)``: a real tokenizer.json."""

    def __init__(self, vocab_size=None, tokenizer=None):
        import tokenizers
        if tokenizer is None:
            from tokenizers import models, pre_tokenizers
            tokenizer = tokenizers.Tokenizer(models.WordLevel({f"t{i}": i for i in range(int(vocab_size))}, unk_token="t0"))
            tokenizer.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
        self.tk = tokenizer

    @classmethod
    def from_file(cls, path):
        import tokenizers
        return cls(tokenizer=tokenizers.Tokenizer.from_file(path))

    def __call__(self, text, return_tensors="pt", truncation=True, max_length=None):
        ids = self.tk.encode(text).ids
        if truncation and max_length is not None:
            ids = ids[:max_length]
        return _Encoding(torch.tensor([ids], dtype=torch.long))

    def decode(self, ids):
        if isinstance(ids, torch.Tensor):
            ids = ids.reshape(-1).tolist()
        return self.tk.decode(list(ids))


class HFForwardRunner:
    """``use_ft=False``: a (swapped) HF causal LM driven through ITS OWN forward, with the runner surface this harness times.  The
    reference's non-FT branches restated: TPS = ``model.generate(ids, min_new_tokens=G, max_new_tokens=G, do_sample=False, num_beams=1,
    attention_mask=...)`` (amq/utils/speed.py:31-36), GeMV / GeMM = the ``past_key_values`` loop with the arg-max token fed back
    (speed.py:93-125).  Nothing is captured or fused here beyond what ``prepare_for_inference`` left in the modules."""

    def __init__(self, model, batch=1, max_seq=None):
        self.model = model
        self.dev = next(model.parameters()).device
        self.vocab = int(model.config.vocab_size)
        self.max_seq = int(max_seq or getattr(model.config, "max_position_embeddings", 1 << 30))
        self.B = int(batch)
        self.reset()

    def reset(self):
        self.past, self.token, self.logits = None, None, None

    def capture(self):          # (the runner's hipGraph hook: HF's forward is timed eagerly, as the reference does)
        pass

    def _step(self, ids):
        out = self.model(ids, past_key_values=self.past, use_cache=True)
        self.past, self.logits = out.past_key_values, out.logits[:, -1]
        self.token = self.logits.max(1)[1].unsqueeze(1)             # speed.py:107 / 121
        return self.logits

    def prefill(self, ids):
        self.past = None
        return self._step(ids.view(1, -1) if ids.dim() == 1 else ids)

    def decode_step(self):
        return self._step(self.token)

    def generate(self, ids, n):
        ids = ids.view(1, -1) if ids.dim() == 1 else ids
        return self.model.generate(ids, min_new_tokens=n, max_new_tokens=n, do_sample=False, num_beams=1,
                                   attention_mask=torch.ones_like(ids))


@torch.inference_mode()
def benchmark_speed(model, tokenizer=None, use_ft=True, iteration=1, sizes=(1, 128, 128), mode="TPS", get_peak_memory=True):
    """speed.py:131-255.  ``model``: a runner with reset()/prefill()/decode_step()/generate() (QuantLlama or DenseLlama), or --
    the reference's calling convention (amq_speed_benchmark.py:152, 253) -- a swapped HF ``LlamaForCausalLM`` itself: with
    ``use_ft`` (default, the reference's fast path) the runner is built over its modules (QuantLlama.from_hf: shared weights,
    static cache, fused token step), as the reference's FT monkeypatch does to its HF model; with ``use_ft=False`` HF's own forward /
    ``generate`` is timed (HFForwardRunner: the reference's non-FT loops).  ``tokenizer``: used by TTFT mode as in
    the reference (None: ids are used directly)."""
    assert mode.lower() in ["tps", "gemv", "gemm", "ttft"], \
        "speed benchmark mode should be one of ['TPS', 'GeMV', 'GeMM', 'TTFT']"
    batch_size, input_seq_len, gen_seq_len = sizes
    if not hasattr(model, "decode_step") and hasattr(model, "lm_head") and hasattr(getattr(model, "model", None), "layers"):
        if use_ft:
            from .llama import QuantLlama
            model = QuantLlama.from_hf(model, max_seq=max(input_seq_len + gen_seq_len, 64), batch=batch_size if batch_size <= 8 else 1)
        else:
            model = HFForwardRunner(model, batch=batch_size)
    if batch_size != getattr(model, "B", 1):
        # a runner built for another batch (the reference's FT path has a batch-1 cache, ftllama_modeling.py:61-68): the
        # batched prompt pass (GeMM mode) needs no cache and is served anyway; token modes need QuantLlama(batch=batch_size)
        if mode.lower() != "gemm":
            raise NotImplementedError(f"batch_size {batch_size} needs a runner built with batch={batch_size} (this one: {getattr(model, 'B', 1)})")
        if input_seq_len > model.max_seq:
            raise ValueError("sizes do not fit the model's RoPE table")
        return _benchmark_gemm_batch(model, sizes, iteration, get_peak_memory)
    if input_seq_len + gen_seq_len > model.max_seq:
        raise ValueError("sizes do not fit the model's KV cache")
    device = model.dev
    data = {mode.lower(): {}}
    if get_peak_memory:
        cleanup()
        torch.cuda.reset_peak_memory_stats(device=device)
        data["peak_memory"] = {}
    input_ids = torch.randint(0, model.vocab - 1, (batch_size, input_seq_len), dtype=torch.long).to(device)   # speed.py:162
    if batch_size == 1:
        input_ids = input_ids[0]
    device_warmup(device)
    cleanup()
    if get_peak_memory:
        torch.cuda.reset_peak_memory_stats(device=device)
    if mode.lower() == "tps":
        model.capture()
        speed = benchmark_tps(model, input_ids, gen_seq_len, iteration)
    elif mode.lower() in ("gemv", "gemm"):
        model.capture()
        speed = benchmark_gemv_gemm(model, input_ids, gen_seq_len, iteration, mode)
    else:
        times = []
        if batch_size != 1:
            raise NotImplementedError("TTFT is a batch-1 measurement (as in the reference: one decoded text)")
        text = tokenizer.decode(input_ids) if tokenizer is not None else None        # speed.py:193
        for _ in range(iteration):
            cleanup()
            model.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if tokenizer is not None:                                                  # speed.py:214-217
                ids = tokenizer(text, return_tensors="pt", truncation=True, max_length=input_seq_len).input_ids[0].to(device)
            else:
                ids = input_ids
            model.prefill(ids)
            first = int(model.token.reshape(-1)[0].item())       # argmax token back on the host
            if tokenizer is not None:
                _ = tokenizer.decode([first])
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1000)
        speed = np.median(times)
    key = f"{batch_size}.{input_seq_len}.{gen_seq_len}"
    data[mode.lower()][key] = float(speed)
    if get_peak_memory:
        data["peak_memory"][key] = torch.cuda.max_memory_allocated(device=device) / 1024 ** 3
    torch.cuda.reset_peak_memory_stats(device=device)
    cleanup()
    return data


__all__ = ["benchmark_speed", "SyntheticTokenizer", "HFForwardRunner"]
