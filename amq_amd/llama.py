"""Mixed-precision Llama decode runner: the caller side of the hot path.

Counterpart of what ``amq_speed_benchmark.py:231-256`` assembles (a Llama whose
7 linears per block are 2/3/4-bit modules chosen by the arch JSON) and of the
patched forward in ``kernel/monkeypatch/ftllama_modeling.py:70-341`` (static
batch-1 KV cache, ``start_pos``), built MI355X-first:

  * one token step = 5 launches per block (fused-RMSNorm q/k/v GEMV with three
    bit-widths in one launch, RoPE+KV-append+attention, o_proj GEMV with
    residual epilogue, fused-RMSNorm gate/up GEMV, SiLU*mul-prologue down GEMV
    with residual epilogue) + fused final-norm lm_head GEMV + argmax;
  * the whole step is captured once into a hipGraph and replayed per token
    (position and token id live in device memory), so the ~165 launches cost
    device-side boundaries only, no Python/ctypes time;
  * prefill runs the same weights through the tiled MFMA GEMM.

Weights: real HQQ layers via ``from_hqq_weights`` (tests, real checkpoints) or
synthetic native payloads of the real shapes (``synthetic=True``; there is no
network for checkpoints -- values do not affect speed).
"""
import math

import torch

from . import ops
from .arch import MODEL_CONFIGS, arch_bits, rope_inv_freq, uniform_arch
from .hqq_format import HQQWeights

EPS = 1e-5
ROPE_THETA = 10000.0


class _Lin:
    """native weights of one linear (+ its fp16 bias: Qwen2's q / k / v projections carry one)"""
    __slots__ = ("qn", "mn", "bits", "mode", "N", "K", "bias")

    def __init__(self, qn, mn, bits, mode, N, K, bias=None):
        self.qn, self.mn, self.bits, self.mode, self.N, self.K, self.bias = qn, mn, bits, mode, N, K, bias

    def seg(self, y, residual=None):
        return dict(qn=self.qn, mn=self.mn, bits=self.bits, mode=self.mode, N=self.N, y=y, residual=residual, bias=self.bias)

    def nbytes(self):
        return self.qn.numel() * 4 + self.mn.numel() * 2 + (0 if self.bias is None else self.bias.numel() * 2)


def _synthetic_linear(n, k, bits, gen, device, group=128):
    """Random native payload + (scale, zero) giving roughly unit-gain layers:
    any bit pattern is a valid weight matrix in the native layout (group: 128, or 64 / 32 = two / four pairs per tile row)."""
    qb, mb = ops.native_sizes(bits, n, k, group)
    qn = torch.randint(-2 ** 31, 2 ** 31 - 1, (qb // 4,), dtype=torch.int32, device=device, generator=gen)
    std_q = math.sqrt((4.0 ** bits - 1.0) / 12.0)
    s0 = 0.5 / (math.sqrt(k) * std_q)
    r = torch.rand(mb // 4, 2, device=device, generator=gen)
    meta = torch.empty(mb // 4, 2, dtype=torch.float16, device=device)
    meta[:, 0] = (s0 * (0.75 + 0.5 * r[:, 0])).to(torch.float16)
    meta[:, 1] = ((2 ** bits - 1) / 2.0 + (r[:, 1] - 0.5)).to(torch.float16)
    return _Lin(qn, meta.reshape(-1).contiguous(), bits, ops.MODE_HQQ, n, k)


_GRAPH_STATE_PRIMED = set()


class _no_gc:
    """no cyclic garbage collection while a hipGraph is being captured: a collection that happens to run inside the capture may finalise a DEAD runner
    (HF models sit in reference cycles: they die only when the collector runs) -- its hipGraph and pool memory are then destroyed in the middle of the
    stream capture, which HIP answers by aborting the process.  (``torch.cuda.graph`` collects once on entry; this keeps it from happening again
    before the capture has ended.)"""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


def _prime_graph_state(dev):
    """torch allocates the per-device tensors its hipGraph captures keep the RNG seed / offset in at the FIRST capture, in whatever mode is current, and
    updates them in place at every later capture.  A first capture under ``torch.inference_mode()`` (the reference's harness decorates every loop with
    it, amq/utils/speed.py:14, 21, 49, 129) makes them inference tensors, and the next capture outside it raises "Inplace update to inference tensor":
    one empty capture outside inference mode, once per device, before any runner captures."""
    key = (dev.type, dev.index)
    if key in _GRAPH_STATE_PRIMED or dev.type != "cuda":
        return
    with torch.inference_mode(False):
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            keep = torch.zeros(1, device=dev)
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                keep += 1
        side.synchronize()
    _GRAPH_STATE_PRIMED.add(key)


class QuantLlama:
    # decode steps run as ONE persistent launch per token (ops.DecodeEngine) when the runner is batch 1 and its KV cache is in the
    # single-workgroup-per-head attention regime (the same bound as ops.ATTN_SPLIT_FROM); otherwise, and with engine=False, as
    # five launches per block
    ENGINE_MAX_SEQ = 512
    # rows up to which down_proj's SiLU*mul stays fused into its GEMV's prologue (beyond: one silu_mul launch + the GEMV without a prologue -- every
    # workgroup of a fused launch takes in gate AND up and repeats the transform on all rows, which from 2 rows on costs more than the extra launch:
    # profiles/r05_decode_batch.txt)
    DOWN_FUSED_ROWS = 1
    # rows up to which the two RMSNorms stay fused into the q/k/v and gate/up launches (beyond: one rmsnorm launch + the grouped GEMV without a prologue)
    NORM_FUSED_ROWS = 4
    NORM_SUMS = True            # (A/B switch of the partial-sum RMSNorm at 2 .. 8 rows: False = fused prologue up to NORM_FUSED_ROWS, one rmsnorm launch per norm beyond)
    # q/k/v + attention of a block as ONE launch (ops.gemv_qkv_attn; batch 1, short cache, hidden <= 8192): 4 launches per block
    # instead of 5.  Built, bit-identical (tests/test_gpu_qkv_attn.py) and SLOWER -- 15.7 us per fused launch against 9.2 + 5.1,
    # 755 vs 830 tokens/s (profiles/r03_qkv_attn_fused_negative.txt) -- so it is off unless a caller sets fuse_qkv_attn.
    FUSE_QKV_ATTN = False
    ENGINE_DEFAULT = False      # what engine=None means (the engine is opt-in until it beats the five-launch step: HISTORY.md 3.2b)
    # prompt passes also leave the logits of EVERY prompt row in self.logits_rows [B, S, vocab] (HF's forward returns them all; the runner's own
    # generate loop needs the last row only): final norm + one fp16 GEMM over the prompt rows, inside the captured prompt graph (hf_fast.py sets it)
    all_logits = False
    fine = False                # any layer with groups of 64 / 32 (set by __init__)

    def __init__(self, config, arch_linear=None, device="cuda:0", max_seq=256, seed=0, synthetic=True,
                 hqq_layers=None, dense=None, batch=1, engine=None, prebuilt=None, group=128, rope=None):
        """config: an entry of arch.MODEL_CONFIGS (or its name).
        arch_linear: {'self_attn.q_proj': [bits]*n_block, ...}; default uniform 4.
        hqq_layers: {(block, name): HQQWeights} real quantized layers (else synthetic).
        dense: {'embed','lm_head','norm','ln1'[n_block],'ln2'[n_block]} fp16 tensors (else synthetic).
        prebuilt: {(block, name): _Lin} linears already in the native layout (from_hf: shared with the modules that own them).
        group: group size of the SYNTHETIC layers (128; 64 / 32: see ``fine``).
        batch: sequences decoded together, 1 .. 8 (same prompt length; one step = the same launches with ``batch`` rows: the
        weights are streamed once per step for all of them).  batch = 1 is the reference's FT configuration.
        rope: (inv_freq fp32 [64], attention_scaling) of the rotary embedding when it is not the plain ``rope_theta`` form (from_hf hands over
        the HF module's own; otherwise derived from config["rope_scaling"]: Llama-3.1's "llama3")."""
        if isinstance(config, str):
            config = MODEL_CONFIGS[config]
        if not 1 <= int(batch) <= 8:
            raise ValueError("batch must be 1..8")
        self.B = int(batch)
        self.cfg = config
        self.dev = torch.device(device)
        _prime_graph_state(self.dev)
        self.H = config["hidden_size"]
        self.I = config["intermediate_size"]
        self.nh, self.nkv = config["num_heads"], config["num_kv_heads"]
        if config["head_dim"] != 128:
            raise ValueError("head_dim must be 128")
        self.kvd = self.nkv * 128
        self.nb = config["n_block"]
        self.vocab = config["vocab_size"]
        self.max_seq = max_seq
        self.eps = float(config.get("rms_norm_eps", EPS))
        self.theta = float(config.get("rope_theta", ROPE_THETA))
        arch_linear = arch_linear or uniform_arch(config, 4)["linear"]
        self.arch_linear = arch_linear
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        dev = self.dev

        def lin(block, name):
            n, k = config["linear_shape"][name]
            bits = arch_bits(arch_linear, name, block)
            if prebuilt is not None:
                l = prebuilt[(block, name)]
                assert l.bits == bits and (l.N, l.K) == (n, k) and l.qn.device == dev
                return l
            if hqq_layers is not None:
                h: HQQWeights = hqq_layers[(block, name)].to(dev)
                assert h.nbits == bits and tuple(h.shape) == (n, k)
                qn, mn = ops.repack_from_hqq(h.W_q.contiguous(), h.scale.reshape(-1).contiguous(),
                                             h.zero.reshape(-1).contiguous(), bits, n, k, group=h.group_size)
                return _Lin(qn, mn, bits, ops.MODE_HQQ, n, k, None if h.bias is None else h.bias.to(dev, torch.float16).contiguous())
            if not synthetic:
                raise ValueError("no weights given")
            l = _synthetic_linear(n, k, bits, gen, dev, group)
            if config.get("qkv_bias") and name in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"):
                l.bias = (0.1 * torch.randn(n, device=dev, generator=gen)).to(torch.float16)
            return l

        self.blocks = []
        for b in range(self.nb):
            blk = {name: lin(b, name) for name in config["linear"]}
            if dense is not None:
                blk["ln1"], blk["ln2"] = dense["ln1"][b].to(dev), dense["ln2"][b].to(dev)
            else:
                blk["ln1"] = (1.0 + 0.05 * torch.randn(self.H, device=dev, generator=gen)).to(torch.float16)
                blk["ln2"] = (1.0 + 0.05 * torch.randn(self.H, device=dev, generator=gen)).to(torch.float16)
            blk["kc"] = torch.zeros(self.B, self.nkv, max_seq, 128, dtype=torch.float16, device=dev)
            blk["vc"] = torch.zeros(self.B, self.nkv, max_seq, 128, dtype=torch.float16, device=dev)
            self.blocks.append(blk)
        if dense is not None:
            self.embed, self.lm_head, self.norm = dense["embed"].to(dev), dense["lm_head"].to(dev), dense["norm"].to(dev)
        else:
            self.embed = (torch.randn(self.vocab, self.H, device=dev, generator=gen)).to(torch.float16)
            self.lm_head = (torch.randn(self.vocab, self.H, device=dev, generator=gen) / math.sqrt(self.H)).to(torch.float16)
            self.norm = (1.0 + 0.05 * torch.randn(self.H, device=dev, generator=gen)).to(torch.float16)

        # layers with groups of 64 / 32 (HQQ's default group_size is 64): the decode step is the same five launches (amq_gemv_grouped_f16 takes the
        # group size); the prompt pass leaves the fragment-ordered few-row kernels alone (they read one (scale, zero) pair per tile) and runs
        # dequantize-once + the fp16 GEMM per linear; the A/B step forms are not offered
        self.fine = any(blk[n].mn.numel() != ops.native_sizes(blk[n].bits, blk[n].N, blk[n].K)[1] // 2 for blk in self.blocks for n in config["linear"])
        if self.fine and engine:
            raise ValueError("the decode engine serves groups of 128")
        # q/k/v and gate/up run as segments of ONE launch, which takes one group size: refuse a model that mixes them inside a sibling set here, at
        # build time (amq_gemv_grouped_f16 would refuse it at the first decode step; patching._same_group keeps such siblings apart on the HF side)
        for bi, blk in enumerate(self.blocks):
            for sibs in (("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), ("mlp.gate_proj", "mlp.up_proj")):
                pairs = {blk[n].mn.numel() * 128 // (2 * blk[n].N * blk[n].K) for n in sibs}      # (scale, zero) pairs per 128 columns: 1 / 2 / 4
                if len(pairs) != 1:
                    raise ValueError(f"block {bi}: {', '.join(sibs)} mix group sizes ({sorted(128 // p for p in pairs)}); the runner issues them as "
                                     "one grouped launch, which needs one group size per sibling set")

        f16 = dict(dtype=torch.float16, device=dev)
        B = self.B
        self.x = torch.zeros(B, self.H, **f16)
        self.xn = torch.zeros(B, self.H, **f16)          # normed rows (launches of more than NORM_FUSED_ROWS rows)
        self.q = torch.zeros(B, self.H, **f16)
        self.k = torch.zeros(B, self.kvd, **f16)
        self.v = torch.zeros(B, self.kvd, **f16)
        self.att = torch.zeros(B, self.H, **f16)
        self.gate = torch.zeros(B, self.I, **f16)
        self.up = torch.zeros(B, self.I, **f16)
        self.logits = torch.zeros(self.vocab, **f16) if B == 1 else torch.zeros(B, self.vocab, **f16)    # [vocab] | [B, vocab]
        self.token = torch.zeros(B, dtype=torch.int64, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        # the cos/sin table every rotating kernel reads: plain rope_theta frequencies, or the rotary embedding's own (rope_scaling)
        self.inv_freq, self.rope_scale = rope if rope is not None else rope_inv_freq(config)
        self.rope_tab = ops.rope_table(max_seq, self.theta, dev, inv_freq=self.inv_freq, scale=self.rope_scale)
        # step state: cos/sin row of self.pos + the position itself in one block (set_token / the step's tail keep it)
        self.rope_cur, self.pos, self.step_err = ops.new_step_state(dev)
        self.rope_cur.copy_(self.rope_tab.view(max_seq, 128)[0])
        self.graph = None
        self.host_pos = 0          # host mirror of self.pos (decode_step refuses to run past the cache without a device sync)
        # token ids the greedy choice never takes (8 slots, -1 = unused; read by the step's tail kernel): what HF's min_new_tokens does to the EOS ids
        # (set_suppressed; the values may change between replays of the captured step)
        self.suppress = torch.full((8,), -1, dtype=torch.int32, device=dev)
        self.has_bias = any(blk[n].bias is not None for blk in self.blocks for n in config["linear"])
        # down_proj's launch: the GEMV with the fused SiLU*mul prologue while the rows' x fits LDS whole; past that (7B: 7 - 8 rows of 11008) one
        # silu_mul launch + the GEMV without a prologue, x staged in two K phases (fusing the prologue there would make every workgroup take in gate
        # AND up -- 352 KB per CU at 8 rows: 23.9 us against 4.9 + ~11); past that too, the few-row MFMA kernel
        self._down_rows_fit = self.B <= min(ops.gemv_max_rows(self.I, plain=not self.fine), self.DOWN_FUSED_ROWS)
        self._down_rows_phased = not self._down_rows_fit and self.B <= ops.gemv_max_rows(self.I, plain=not self.fine, norm=False)
        self.can_fuse_qkv_attn = self.B == 1 and max_seq <= ops.ATTN_SPLIT_FROM and self.H <= 8192 and not self.fine and not self.has_bias
        self.fuse_qkv_attn = self.FUSE_QKV_ATTN and self.can_fuse_qkv_attn
        self._tickets = torch.zeros(max(self.nh, 64), dtype=torch.int32, device=dev)
        # 2 .. 8 sequences: the RMSNorms ride on per-row-tile sums of squares that o_proj / down_proj leave in their epilogues (ops.gemv_grouped_sums:
        # no pass over x for the statistic -- which a fused prologue repeats in every workgroup --, no rmsnorm launch); the first norm of block 0 (its x
        # comes from the embedding) keeps the fused prologue (2 .. 4 rows) / a launch of its own (5 .. 8)
        self._norm_sums = (self.NORM_SUMS and 2 <= self.B <= 8 and not self.fine and 2048 <= self.H <= 8192 and self.I >= 2048
                           and self.B <= ops.gemv_max_rows(self.H, plain=True))
        self.ss = torch.zeros(self.B, self.H // 16, dtype=torch.float32, device=dev) if self._norm_sums else None
        eligible = self.B == 1 and max_seq <= self.ENGINE_MAX_SEQ and self.H == self.nh * 128 and not self.fine and not self.has_bias
        if engine and not eligible:
            raise ValueError("the decode engine needs batch 1 and max_seq <= %d" % self.ENGINE_MAX_SEQ)
        self.engine = None
        if engine is None:
            engine = self.ENGINE_DEFAULT
        if eligible and engine:
            self.engine = ops.DecodeEngine(
                [dict({n: dict(qn=blk[n].qn, mn=blk[n].mn, bits=blk[n].bits, mode=blk[n].mode, N=blk[n].N) for n in ops.ENGINE_LINEARS},
                      ln1=blk["ln1"], ln2=blk["ln2"], kc=blk["kc"], vc=blk["vc"]) for blk in self.blocks],
                self.H, self.I, self.nh, self.nkv, max_seq, self.eps, self.x.view(-1), self.rope_cur)

    # model families whose decoder is the Llama block -- RMSNorm, rotary q / k, (grouped-query) softmax attention, SiLU-gated MLP, no extra norms --
    # and differs only in shapes, rope settings and projection biases: what the reference lists (README.md:90-92; amq/configs/{llama,mistral,qwen2}.json)
    HF_MODEL_TYPES = ("llama", "mistral", "qwen2")

    @classmethod
    def check_hf(cls, model, max_seq=None):
        """What ``from_hf`` needs of a swapped HF causal LM, checked without building anything: raises ValueError with the reason, returns
        (runner config, (inv_freq, attention_scaling) or None)."""
        from .checkpoint import runner_config
        hf = model.config.to_dict()
        mt = hf.get("model_type", "llama")
        if mt not in cls.HF_MODEL_TYPES:
            raise ValueError(f"from_hf: model_type '{mt}' is not one of {cls.HF_MODEL_TYPES} (a Llama-shaped decoder is required)")
        if hf.get("hidden_act", "silu") != "silu":
            raise ValueError("from_hf: a SiLU-gated MLP is required")
        attn0 = model.model.layers[0].self_attn
        if any(hasattr(attn0, n) for n in ("q_norm", "k_norm")):
            raise ValueError("from_hf: per-head q / k norms are not part of the runner's block")
        from .quant_linear import HIPQuantLinear
        if not isinstance(getattr(attn0, "q_proj", None), HIPQuantLinear):
            raise ValueError("from_hf: the decoder linears are not HIPQuantLinear modules: run prepare_for_inference(model, backend='hip') first")
        rp = hf.get("rope_parameters") or {}
        if hf.get("rope_theta") is None:
            hf["rope_theta"] = rp.get("rope_theta", 10000.0)
        rot = getattr(model.model, "rotary_emb", None)
        kind = getattr(rot, "rope_type", None) or rp.get("rope_type") or (hf.get("rope_scaling") or {}).get("rope_type", "default")
        rope = None
        if kind != "default":
            if kind not in ("llama3", "linear", "yarn") or rot is None or getattr(rot, "inv_freq", None) is None:
                raise ValueError(f"from_hf: rope type '{kind}' is not served (default, linear, llama3, yarn: the static ones)")
            rope = (rot.inv_freq.detach().to(torch.float32), float(getattr(rot, "attention_scaling", 1.0)))
            if rope[0].numel() != 64:
                raise ValueError("from_hf: head_dim must be 128")
        sw = hf.get("sliding_window")
        sliding = sw is not None and hf.get("use_sliding_window", True) is not False and \
            (not hf.get("layer_types") or any(t == "sliding_attention" for t in hf["layer_types"]))
        if sliding and max_seq is not None and max_seq > int(sw):
            raise ValueError(f"from_hf: max_seq {max_seq} exceeds the model's sliding window ({sw}): the runner attends the whole cache")
        cfg = runner_config(hf)
        return cfg, rope

    @classmethod
    def from_hf(cls, model, max_seq=256, batch=1, engine=None):
        """The hipGraph runner over a SWAPPED HF causal LM of the Llama family (``HF_MODEL_TYPES``: Llama 2 / 3.x, Mistral, Qwen2.5) -- what
        ``prepare_for_inference(model, backend="hip")`` (or the reference's deepcopy + setattr assembly of a mixed-precision model,
        amq_speed_benchmark.py:231-256) leaves behind.  The runner shares the modules' native weight buffers and biases, the embedding, lm_head and
        norm weights (no copies) and the rotary embedding's own frequencies (rope_scaling: Llama-3.1); it is to the swapped model what the
        reference's ``use_ft`` monkeypatch is to its HF model (kernel/monkeypatch/ftllama_modeling.py): the same weights behind a static-cache,
        fused token step.  Needs fp16 weights on one GPU, head_dim 128, SiLU."""
        from .quant_linear import HIPQuantLinear
        cfg, rope = cls.check_hf(model, max_seq)
        layers = model.model.layers
        pre, arch_linear = {}, {name: [] for name in cfg["linear"]}
        dev = None
        f16 = lambda t: t.detach() if t.dtype is torch.float16 else t.detach().to(torch.float16)
        for b, layer in enumerate(layers):
            for name in cfg["linear"]:
                parent, attr = name.split(".")
                m = getattr(getattr(layer, parent), attr)
                if not isinstance(m, HIPQuantLinear) or not m.qweight.is_cuda:
                    raise ValueError(f"model.layers.{b}.{name}: expected a HIPQuantLinear on the GPU (run prepare_for_inference first)")
                if m.is_bf16:
                    raise ValueError(f"model.layers.{b}.{name}: a bfloat16 module -- the decode runner is fp16 (the reference's kernels are, ft.py:62); "
                                     "bf16 models run through the modules' own forward")
                dev = dev or m.qweight.device
                pre[(b, name)] = _Lin(m.qweight, m.meta, m.bits, m.mode, m.outfeatures, m.infeatures, None if m.bias is None else f16(m.bias))
                arch_linear[name].append(m.bits)
        dense = {"embed": f16(model.model.embed_tokens.weight), "lm_head": f16(model.lm_head.weight), "norm": f16(model.model.norm.weight),
                 "ln1": [f16(l.input_layernorm.weight) for l in layers], "ln2": [f16(l.post_attention_layernorm.weight) for l in layers]}
        return cls(cfg, arch_linear, device=dev, max_seq=max_seq, dense=dense, batch=batch, engine=engine, prebuilt=pre, synthetic=False, rope=rope)

    # ----------------------------------------------------------------- sizes
    def linear_bytes_per_token(self):
        """algorithmic bytes of the quantized linears per decode token"""
        return sum(blk[name].nbytes() for blk in self.blocks for name in self.cfg["linear"])

    def total_bytes_per_token(self, context):
        kv = 2 * self.nb * self.nkv * 128 * 2 * context
        return self.linear_bytes_per_token() + self.lm_head.numel() * 2 + kv

    # ----------------------------------------------------------------- decode
    def _step(self):
        """one token: reads self.x (= embed[self.token], kept in step by set_token / the step's own tail) and self.pos
        (device), writes self.logits, self.token, self.pos and the next step's self.x"""
        H = self.H
        if self.engine is not None:
            self.engine.step()
            ops.gemv_f16w(self.x.reshape(-1), self.lm_head, gamma=self.norm, eps=self.eps, out=self.logits)
            ops.decode_tail(self.logits, self.embed, self.token, self.pos, self.x, table=self.rope_tab, cur=self.rope_cur, suppress=self.suppress)
            return
        have_sums = False                       # self.ss holds the sums of squares of self.x's rows (written by the launch that produced them)
        for blk in self.blocks:
            if self.fuse_qkv_attn:
                ops.gemv_qkv_attn(self.x, [blk["self_attn.q_proj"].seg(self.q.view(-1)), blk["self_attn.k_proj"].seg(self.k.view(-1)),
                                           blk["self_attn.v_proj"].seg(self.v.view(-1))], H, blk["ln1"], self.eps, blk["kc"], blk["vc"],
                                  self.att.view(-1), self.rope_cur, self.nh, self.nkv, self._tickets)
            else:
                qkv = [blk["self_attn.q_proj"].seg(self.q), blk["self_attn.k_proj"].seg(self.k), blk["self_attn.v_proj"].seg(self.v)]
                if self._norm_sums and have_sums:
                    ops.gemv_grouped_sums(self.x, qkv, H, gamma=blk["ln1"], eps=self.eps, sums_in=self.ss)
                elif self.B > self.NORM_FUSED_ROWS:
                    ops.gemv_grouped(ops.rmsnorm(self.x, blk["ln1"], self.eps, out=self.xn), qkv, H)
                else:
                    ops.gemv_grouped(self.x, qkv, H, prologue=ops.PRO_RMSNORM, gamma=blk["ln1"], eps=self.eps)
                ops.attn_decode(self.q, self.k, self.v, blk["kc"], blk["vc"], self.att, self.pos, self.nh, self.nkv, self.theta,
                                cur=self.rope_cur)
            if self._norm_sums:
                ops.gemv_grouped_sums(self.att, [blk["self_attn.o_proj"].seg(self.x, residual=self.x)], H, sums_out=self.ss)
                ops.gemv_grouped_sums(self.x, [blk["mlp.gate_proj"].seg(self.gate), blk["mlp.up_proj"].seg(self.up)], H,
                                      gamma=blk["ln2"], eps=self.eps, sums_in=self.ss)
            else:
                ops.gemv_grouped(self.att, [blk["self_attn.o_proj"].seg(self.x, residual=self.x)], H)
            if self._norm_sums:
                pass                            # (gate / up were launched above, behind o_proj's sums)
            elif self.B > self.NORM_FUSED_ROWS:
                ops.gemv_grouped(ops.rmsnorm(self.x, blk["ln2"], self.eps, out=self.xn), [blk["mlp.gate_proj"].seg(self.gate), blk["mlp.up_proj"].seg(self.up)], H)
            else:
                ops.gemv_grouped(self.x, [blk["mlp.gate_proj"].seg(self.gate), blk["mlp.up_proj"].seg(self.up)], H,
                                 prologue=ops.PRO_RMSNORM, gamma=blk["ln2"], eps=self.eps)
            if self._down_rows_fit:
                ops.gemv_grouped(self.gate, [blk["mlp.down_proj"].seg(self.x, residual=self.x)], self.I,
                                 prologue=ops.PRO_SILU_MUL, x2=self.up)
            elif self._down_rows_phased and self._norm_sums:
                ops.gemv_grouped_sums(ops.silu_mul(self.gate, self.up, out=self.gate), [blk["mlp.down_proj"].seg(self.x, residual=self.x)], self.I,
                                      sums_out=self.ss)
                have_sums = True
            elif self._down_rows_phased:
                ops.gemv_grouped(ops.silu_mul(self.gate, self.up, out=self.gate), [blk["mlp.down_proj"].seg(self.x, residual=self.x)], self.I)
            else:       # batch x intermediate size past the GEMV kernel's LDS stage: few-row MFMA kernel
                d = blk["mlp.down_proj"]
                ops.gemm(ops.silu_mul(self.gate, self.up, out=self.gate), d.qn, d.mn, d.bits, d.mode, d.N, d.K, bias=d.bias, residual=self.x, out=self.x)
                have_sums = False
        ops.gemv_f16w(self.x.reshape(-1) if self.B == 1 else self.x, self.lm_head, gamma=self.norm, eps=self.eps, out=self.logits)
        # argmax, pos += 1, x = embed[token], rope_cur = cos/sin row of the new position (per sequence; the position is shared)
        ops.decode_tail(self.logits, self.embed, self.token, self.pos, self.x, table=self.rope_tab, cur=self.rope_cur, suppress=self.suppress)

    def set_suppressed(self, ids=()):
        """token ids greedy decoding must not pick (at most 8; () = none): HF's generate(min_new_tokens = max_new_tokens) never emits an EOS id.
        Takes effect from the next step / prompt pass, captured or not."""
        ids = [int(i) for i in ids]
        if len(ids) > 8:
            raise ValueError("at most 8 suppressed token ids")
        self.suppress.copy_(torch.tensor(ids + [-1] * (8 - len(ids)), dtype=torch.int32))
        self._suppressed = tuple(ids)

    def _argmax(self, logits, dim, keepdim=False):
        """arg-max over the vocabulary (the last dimension) of a prompt pass with the suppressed ids left out, as the captured step's tail kernel
        does; no host synchronisation (unused slots are pointed at a spare element behind the vocabulary)"""
        assert dim in (-1, logits.dim() - 1)
        m = torch.zeros(self.vocab + 1, dtype=torch.float32, device=logits.device)
        m.index_fill_(0, torch.where(self.suppress >= 0, self.suppress, self.vocab).to(torch.int64), float("-inf"))
        return torch.argmax(logits.float() + m[:self.vocab], dim=dim, keepdim=keepdim)

    def set_pos(self, pos):
        """set the position of the next decode step (device state + its host mirror); follow with set_token()"""
        pos = int(pos)
        if not 0 <= pos <= self.max_seq:
            raise ValueError(f"position {pos} outside the KV cache (max_seq={self.max_seq})")
        self.pos.fill_(pos)
        self.host_pos = pos

    def check(self):
        """raise if any decode step ran with its device-side position outside the cache (synchronises)"""
        try:
            ops.check_step_state(self.step_err)
        except Exception:
            self._tickets.zero_()               # (the fused q/k/v + attention launch leaves a timed-out ticket as it is: include/amq_hip.h)
            raise
        if self.engine is not None:
            try:
                self.engine.check()
            except Exception:
                self.graph = None               # the engine re-zeroed its barrier words: the next decode_step re-captures
                raise

    def set_token(self, token):
        """make ``token`` (int or 1-element tensor) the input of the next decode step; also re-derives what the step
        reads besides the token (embedding row, cos/sin row of the current position) -- set_pos() first"""
        if isinstance(token, torch.Tensor) and token.is_cuda and token.dtype is torch.int64 and token.numel() in (1, self.B) and token.is_contiguous():
            # one launch instead of five framework ops (a caller that feeds every token itself pays this per token: hf_fast's forward)
            return ops.set_token(token, self.embed, self.token, self.pos, self.x, table=self.rope_tab, cur=self.rope_cur)
        if isinstance(token, torch.Tensor):
            self.token.copy_(token.reshape(-1).expand(self.B) if token.numel() == 1 else token.reshape(self.B))
        else:
            self.token.fill_(int(token))
        torch.index_select(self.embed, 0, self.token, out=self.x)
        torch.index_select(self.rope_tab.view(self.max_seq, 128), 0, self.pos.to(torch.int64).clamp_(0, self.max_seq - 1),
                           out=self.rope_cur.view(1, 128))

    def capture(self):
        """capture one token step into a hipGraph (replayed by decode_step)"""
        if self.graph is not None:
            return
        if self.host_pos >= self.max_seq:
            raise ValueError(f"cannot capture a decode step at position {self.host_pos}: the KV cache holds {self.max_seq} rows")
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        saved = (self.token.clone(), self.pos.clone())
        with torch.cuda.stream(side):
            self._step()                       # warm-up outside capture (allocator, lazy init)
            side.synchronize()
            self.pos.copy_(saved[1]); self.set_token(saved[0])
            g = torch.cuda.CUDAGraph()
            with _no_gc(), torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                self._step()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.pos.copy_(saved[1]); self.set_token(saved[0])
        self.graph = g

    def decode_step(self, use_graph=True):
        # the step appends cache row host_pos: refuse on the host (the kernels also guard the device-side position:
        # a step past the cache is skipped there and raises the sticky error word, see check())
        if self.host_pos >= self.max_seq:
            raise ValueError(f"decode step at position {self.host_pos} does not fit the KV cache (max_seq={self.max_seq})")
        if use_graph and self.graph is None:
            self.capture()
        self.host_pos += 1
        if use_graph:
            self.graph.replay()
        else:
            self._step()

    # ---------------------------------------------------------------- prefill
    def _rope(self, t, positions):
        # HF apply_rotary_pos_emb: cos/sin in fp32 -> fp16; rotate_half
        inv = 1.0 / (self.theta ** (torch.arange(0, 128, 2, device=self.dev, dtype=torch.float32) / 128.0)) if self.inv_freq is None \
            else self.inv_freq.to(self.dev, torch.float32)
        fr = positions.to(torch.float32)[:, None] * inv[None, :]
        emb = torch.cat([fr, fr], dim=-1)
        cos, sin = (emb.cos() * self.rope_scale).to(torch.float16)[:, None, :], (emb.sin() * self.rope_scale).to(torch.float16)[:, None, :]
        t1, t2 = t[..., :64], t[..., 64:]
        rot = torch.cat([-t2, t1], dim=-1)
        return t * cos + rot * sin

    def prefill(self, ids, use_graph=True, start_pos=0):
        """ids: int64 [S] prompt.  Fills the KV caches, leaves the next token in self.token and pos = start_pos + S.
        The ~25 framework launches per block make an eager prefill host-bound for short prompts (13 ms at S = 64,
        of which ~1 ms is GPU work); with ``use_graph`` the whole prefill of a given prompt LENGTH is captured once
        into a hipGraph and replayed for later prompts of that length.
        ``start_pos`` > 0 (the reference's patched forward takes the same argument, ftllama_modeling.py:76,98-104): the rows
        are appended behind ``start_pos`` cached positions -- a prompt fed in chunks, or the next turn of a conversation --
        and attend the whole cache; the graph cache is keyed by (length, start_pos)."""
        ids = self._ids_rows(ids)
        S = ids.shape[1]
        start_pos = int(start_pos)
        if start_pos < 0 or start_pos + S > self.max_seq:
            raise ValueError("prompt longer than the KV cache")
        if not use_graph:
            return self._prefill_rows(ids, start_pos)
        cache = self.__dict__.setdefault("_prefill_graphs", {})
        ent = cache.get((S, start_pos))
        if ent is None:
            static_ids = ids.to(self.dev).clone()
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._prefill_rows(static_ids, start_pos)      # warm-up outside capture (allocator, lazy init)
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with _no_gc(), torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    self._prefill_rows(static_ids, start_pos)
            torch.cuda.current_stream(self.dev).wait_stream(side)
            ent = cache[(S, start_pos)] = (g, static_ids, self.__dict__.get("logits_rows"))
        g, static_ids, rows = ent
        static_ids.copy_(ids)
        g.replay()
        if rows is not None:
            self.logits_rows = rows             # (this graph's own output buffer: valid until its next replay)
        self.host_pos = start_pos + S           # (the replay sets the device-side position; the host mirror is not part of it)
        return self.logits

    def _ids_rows(self, ids):
        """prompt ids as [batch, S] on the device (a 1-D prompt is the batch-1 form)"""
        ids = ids.to(self.dev)
        if ids.dim() == 1:
            ids = ids[None, :]
        if ids.dim() != 2 or ids.shape[0] != self.B:
            raise ValueError(f"expected {self.B} prompt(s) of equal length, got ids of shape {tuple(ids.shape)}")
        return ids

    def _prefill_rows(self, ids, start_pos):
        """the prompt pass of every sequence (each into its own slice of the caches), then the shared position / next tokens"""
        if self.B == 1:
            return self._prefill_eager(ids[0], start_pos)
        B, S = ids.shape
        last = self._rows_pass(ids, start_pos, cache=True)
        if self.all_logits:
            self.logits_rows = self._logits_of_rows(self.__dict__.pop("_rows_x"), B, S, self.logits)
        else:
            ops.gemv_f16w(last, self.lm_head, gamma=self.norm, eps=self.eps, out=self.logits)
        self.set_pos(start_pos + S)
        self.set_token(self._argmax(self.logits, 1))
        return self.logits

    def _rows_pass(self, ids, start_pos, cache):
        """ONE many-row pass over B prompts of S rows (ids [B, S]): the linears see all B * S rows at once (one pass over
        the weights), RoPE and the causal attention run as ONE launch each over all sequences.  ``cache``: the rotated keys /
        values are written into the runner's KV caches (rows start_pos .. start_pos + S - 1 of every sequence) and the
        attention reads them there (a batched decode runner's prompt); otherwise q / k are rotated in place and the
        attention reads the projection outputs (the harness' GeMM mode, no cache).  Returns the last rows [B, H]."""
        B, S = ids.shape
        H, nh, nkv = self.H, self.nh, self.nkv
        x = self.embed.index_select(0, ids.reshape(-1).to(self.dev))
        lin = self._rows_linear
        for blk in self.blocks:
            h = ops.rmsnorm(x, blk["ln1"], self.eps)
            q, k, v = lin(blk["self_attn.q_proj"], h), lin(blk["self_attn.k_proj"], h), lin(blk["self_attn.v_proj"], h)
            if cache:
                ops.rope_cache(q, k, v, blk["kc"], blk["vc"], self.rope_tab, start_pos, nh, nkv)
                a = ops.attn_prefill(q, blk["kc"], blk["vc"], torch.empty_like(q), S, nh, nkv, batch=B, pos0=start_pos, kv_cache=True)
            else:
                ops.rope_rows(q, k, self.rope_tab, S, nh, nkv)
                a = ops.attn_prefill(q, k, v, torch.empty_like(q), S, nh, nkv, batch=B)      # reads the projections in place
            x = lin(blk["self_attn.o_proj"], a, residual=x)
            h2 = ops.rmsnorm(x, blk["ln2"], self.eps)
            act = self._rows_up_gated(blk["mlp.up_proj"], h2, lin(blk["mlp.gate_proj"], h2))
            x = lin(blk["mlp.down_proj"], act, residual=x)
        if self.all_logits:
            self._rows_x = x                    # (picked up by _prefill_rows)
        return x.view(B, S, H)[:, S - 1].contiguous()

    def _logits_of_rows(self, x, B, S, last_logits):
        """logits of every prompt row, [B, S, vocab] fp16: final RMSNorm + ONE pass over the lm_head for all rows (fp16 MFMA GEMM); each sequence's
        last row is copied into ``last_logits`` (the runner's token choice), instead of a second pass over the lm_head for it"""
        if self.vocab % 16 == 0:
            rows = ops.gemm_f16w(ops.rmsnorm(x, self.norm, self.eps), self.lm_head).view(B, S, self.vocab)
        else:                                   # (the fp16 GEMM writes 16-column blocks; odd vocabularies -- test models -- go through the weight-streaming kernel, 8 rows a launch)
            rows = torch.empty(B * S, self.vocab, dtype=torch.float16, device=x.device)
            for r0 in range(0, B * S, 8):
                r1 = min(B * S, r0 + 8)
                if r1 - r0 == 1:
                    ops.gemv_f16w(x[r0], self.lm_head, gamma=self.norm, eps=self.eps, out=rows[r0])
                else:
                    ops.gemv_f16w(x[r0:r1], self.lm_head, gamma=self.norm, eps=self.eps, out=rows[r0:r1])
            rows = rows.view(B, S, self.vocab)
        last_logits.view(B, self.vocab).copy_(rows[:, S - 1])
        return rows

    # prompt rows (exclusive, inclusive) served by the fragment-ordered few-row kernels with q/k/v and gate/up as grouped
    # launches.  7B avg-3, ms per prompt pass, this path | the row-major / tiled kernels: 16 rows 2.91 | 2.55, 24 3.03 | 3.13,
    # 32 3.06 | 3.28, 64 3.19 | 3.72, 256 6.74 | 7.32, 384 10.86 | 11.30, 512 12.58 | 11.26
    FRAG_ROWS = (16, 384)
    FUSE_DOWN_NORM = True       # down_proj's split-K reduce also writes the next block's normed input (A/B: tools/prompt64_time.py)

    def _prefill_eager(self, ids, start_pos=0, b=None):
        """(b: sequence of a batched runner -- its slice of the caches and its logits row, position / token left to the caller)
        Many-row pass over the prompt: per block 2 RMSNorm + 7 GEMMs (residuals fused into the o_proj / down_proj
        epilogues) + one RoPE-and-cache-write launch + causal attention (library SDPA reading K/V straight from the
        cache) + one SiLU*up launch."""
        S = ids.numel()
        if start_pos + S > self.max_seq:
            raise ValueError("prompt longer than the KV cache")
        H, nh, nkv = self.H, self.nh, self.nkv
        x = self.embed.index_select(0, ids.to(self.dev))           # [S, H]; the residual stream, updated in place

        def lin(l, inp, residual=None):
            if S > 8:
                return ops.gemm(inp, l.qn, l.mn, l.bits, l.mode, l.N, l.K, bias=l.bias, residual=residual, out=residual)
            y = ops.linear(inp, l.qn, l.mn, l.bits, l.mode, l.N, l.K, bias=l.bias)
            return y if residual is None else residual.add_(y)

        def lin_xf(l, xf, residual=None, gate=None):
            return ops.gemm_xfrag(xf, S, l.qn, l.mn, l.bits, l.mode, l.N, l.K, bias=l.bias, residual=residual, gate=gate,
                                  out=residual if residual is not None else gate)

        def lin_xf_group(ls, xf):
            ys = [torch.empty(S, l.N, dtype=torch.float16, device=self.dev) for l in ls]
            ops.gemm_xfrag_grouped(xf, S, [l.seg(y) for l, y in zip(ls, ys)], ls[0].K)
            return ys

        # up to 256 rows the projections that read a normed / attention activation take it in fragment order (written
        # that way by the producing launch): 1.2-1.6x faster few-row GEMMs; down_proj (K = 11008: the
        # per-workgroup x stream is what bounds that kernel) and longer prompts stay on the tiled kernel
        frag = self.FRAG_ROWS[0] < S <= self.FRAG_ROWS[1] and not self.fine
        h_next = None                                # the next block's normed input, left behind by this block's down_proj (its split-K reduce carries the norm)
        for bi, blk in enumerate(self.blocks):
            if frag:
                h = h_next if h_next is not None else ops.rmsnorm_xfrag(x, blk["ln1"], self.eps)
                q, k, v = lin_xf_group([blk["self_attn." + n] for n in ("q_proj", "k_proj", "v_proj")], h)   # one launch
            else:
                h = ops.rmsnorm(x, blk["ln1"], self.eps)
                q, k, v = lin(blk["self_attn.q_proj"], h), lin(blk["self_attn.k_proj"], h), lin(blk["self_attn.v_proj"], h)
            kc, vc = self._cache_rows(blk, b)
            ops.rope_cache(q, k, v, kc[0], vc[0], self.rope_tab, start_pos, nh, nkv)
            if frag:
                a_xf = ops.attn_prefill(q, kc, vc, None, S, nh, nkv, batch=1, pos0=start_pos, kv_cache=True, out_xfrag=True)
                x = lin_xf(blk["self_attn.o_proj"], a_xf, residual=x)              # attention output handed over in fragment order
                h2 = ops.rmsnorm_xfrag(x, blk["ln2"], self.eps)
                g, u = lin_xf_group([blk["mlp.gate_proj"], blk["mlp.up_proj"]], h2)                            # one launch
                act = ops.silu_mul(g, u, out=g)
            else:
                x = lin(blk["self_attn.o_proj"], self._prefill_attention(q, blk, S, start_pos, b), residual=x)
                h2 = ops.rmsnorm(x, blk["ln2"], self.eps)
                g = lin(blk["mlp.gate_proj"], h2)
                if S > 8:
                    u = blk["mlp.up_proj"]                                         # silu(gate) * up in up_proj's epilogue
                    act = ops.gemm(h2, u.qn, u.mn, u.bits, u.mode, u.N, u.K, bias=u.bias, gate=g, out=g)
                else:
                    act = ops.silu_mul(g, lin(blk["mlp.up_proj"], h2), out=g)
            if frag and self.FUSE_DOWN_NORM and bi + 1 < len(self.blocks):
                l = blk["mlp.down_proj"]
                x, h_next = ops.gemm_res_norm_xfrag(act, l.qn, l.mn, l.bits, l.mode, l.N, l.K, self.blocks[bi + 1]["ln1"], self.eps,
                                                    bias=l.bias, residual=x, out=x)
            else:
                x = lin(blk["mlp.down_proj"], act, residual=x)
        return self._prefill_finish(x, S, start_pos, b)

    def _rows_linear(self, l, inp, residual=None):
        # y = inp . W^T (+ residual, in place) for many rows
        return ops.gemm(inp, l.qn, l.mn, l.bits, l.mode, l.N, l.K, bias=l.bias, residual=residual, out=residual)

    def _rows_up_gated(self, l, inp, gate):
        # silu(gate) * (inp . W^T), in place on gate: the LlamaMLP product formed in up_proj's epilogue
        return ops.gemm(inp, l.qn, l.mn, l.bits, l.mode, l.N, l.K, bias=l.bias, gate=gate, out=gate)

    def prefill_batch(self, ids):
        """ids: int64 [B, S].  The many-row pass over B prompts at once (B*S rows through every linear): what the reference
        harness times in GeMM mode with batch_size > 1 (amq/utils/speed.py:61-71; BASELINE.json configs[3] = 16 x 2048 on
        13B).  Returns the last-token logits [B, vocab].  The runner's KV cache is batch-1 (like the reference's FT path),
        so this pass does not write it and cannot be followed by decode steps."""
        B, S = ids.shape
        if S > self.max_seq:
            raise ValueError("prompt longer than the RoPE table")
        last = self._rows_pass(ids, 0, cache=False)
        logits = torch.empty(B, self.vocab, dtype=torch.float16, device=self.dev)
        for b0 in range(0, B, 8):                   # the lm_head is streamed once per 8 sequences
            rows = slice(b0, min(B, b0 + 8))
            if rows.stop - rows.start == 1:
                ops.gemv_f16w(last[b0], self.lm_head, gamma=self.norm, eps=self.eps, out=logits[b0])
            else:
                ops.gemv_f16w(last[rows], self.lm_head, gamma=self.norm, eps=self.eps, out=logits[rows])
        return logits

    def _cache_rows(self, blk, b=None):
        """the [1, n_kv_heads, max_seq, 128] cache views of sequence b (None: the batch-1 runner's whole caches)"""
        if b is None:
            return blk["kc"], blk["vc"]
        return blk["kc"][b:b + 1], blk["vc"][b:b + 1]

    def _prefill_attention(self, q, blk, S, start_pos=0, b=None):
        """causal attention of the prompt rows: q [S, nh*128] rotated, K / V = cache rows 0 .. start_pos + S - 1 -> [S, nh*128]"""
        kc, vc = self._cache_rows(blk, b)
        return ops.attn_prefill(q, kc, vc, torch.empty_like(q), S, self.nh, self.nkv, batch=1, pos0=start_pos, kv_cache=True)

    def _prefill_attention_sdpa(self, q, blk, S):
        """the same through the framework's SDPA (comparison point for tests / tools; not on the product path)"""
        nh, nkv = self.nh, self.nkv
        qh = q.view(S, nh, 128).transpose(0, 1)
        kh, vh = blk["kc"][0, :, :S], blk["vc"][0, :, :S]
        if nkv != nh:
            kh = kh.repeat_interleave(nh // nkv, dim=0)
            vh = vh.repeat_interleave(nh // nkv, dim=0)
        a = torch.nn.functional.scaled_dot_product_attention(qh[None], kh[None], vh[None], is_causal=True)[0]
        return a.transpose(0, 1).reshape(S, self.H).contiguous()

    def _prefill_finish(self, x, S, start_pos=0, b=None):
        last = x[S - 1].contiguous()
        if b is not None:                       # one sequence of a batch: its logits row; the caller sets position and tokens
            ops.gemv_f16w(last, self.lm_head, gamma=self.norm, eps=self.eps, out=self.logits[b])
            return self.logits[b]
        if self.all_logits:
            self.logits_rows = self._logits_of_rows(x, 1, S, self.logits)
        else:
            ops.gemv_f16w(last, self.lm_head, gamma=self.norm, eps=self.eps, out=self.logits)
        self.set_pos(start_pos + S)
        self.set_token(self._argmax(self.logits, 0, keepdim=True))
        return self.logits

    def _prefill_unfused(self, ids):
        """The same pass with framework ops for everything but the linears and RMSNorm (HF-style RoPE in fp32 -> fp16,
        separate cache copies, residual adds, SiLU and product): kept as the comparison point of the fused pass
        (tests/test_gpu_decode.py)."""
        S = ids.numel()
        if S > self.max_seq:
            raise ValueError("prompt longer than the KV cache")
        H, nh, nkv = self.H, self.nh, self.nkv
        x = self.embed.index_select(0, ids.to(self.dev))
        positions = torch.arange(S, device=self.dev)

        def lin(l, inp):
            return ops.linear(inp, l.qn, l.mn, l.bits, l.mode, l.N, l.K, bias=l.bias)

        for blk in self.blocks:
            h = ops.rmsnorm(x, blk["ln1"], self.eps)
            q = lin(blk["self_attn.q_proj"], h).view(S, nh, 128)
            k = lin(blk["self_attn.k_proj"], h).view(S, nkv, 128)
            v = lin(blk["self_attn.v_proj"], h).view(S, nkv, 128)
            q, k = self._rope(q, positions), self._rope(k, positions)
            blk["kc"][0, :, :S] = k.transpose(0, 1)
            blk["vc"][0, :, :S] = v.transpose(0, 1)
            x = x + lin(blk["self_attn.o_proj"], self._prefill_attention_sdpa(q.reshape(S, nh * 128).contiguous(), blk, S))
            h2 = ops.rmsnorm(x, blk["ln2"], self.eps)
            g, u = lin(blk["mlp.gate_proj"], h2), lin(blk["mlp.up_proj"], h2)
            x = x + lin(blk["mlp.down_proj"], torch.nn.functional.silu(g) * u)
        return self._prefill_finish(x, S)

    def reset(self):
        self.set_pos(0)
        self.set_token(0)

    def generate(self, ids, gen_len, use_graph=True):
        """greedy: prefill + gen_len tokens (min_new_tokens = max_new_tokens = gen_len,
        amq/utils/speed.py:34-39).  Returns the generated ids (device tensor)."""
        S = ids.shape[-1]
        if S + gen_len > self.max_seq:
            raise ValueError("sequence does not fit the KV cache")
        out = torch.empty(self.B, gen_len, dtype=torch.int64, device=self.dev)
        self.prefill(ids)
        out[:, 0] = self.token
        for i in range(1, gen_len):
            self.decode_step(use_graph)
            out[:, i] = self.token
        if self.engine is not None:
            self.check()                        # a barrier time-out of the one-launch-per-token engine must not pass as tokens
        return out[0] if self.B == 1 else out


class DenseLlama(QuantLlama):
    """fp16 baseline with the same runner interface (the reference's ``result['fp16']`` row,
    amq_speed_benchmark.py:171-197): plain library GEMMs (torch / hipBLASLt) for the seven
    linears, the same RMSNorm / attention / lm_head kernels around them."""

    def __init__(self, config, device="cuda:0", max_seq=256, seed=0, batch=1):
        if isinstance(config, str):
            config = MODEL_CONFIGS[config]
        if not 1 <= int(batch) <= 8:
            raise ValueError("batch must be 1..8")
        self.B = int(batch)
        self._dense_init(config, device, max_seq, seed)

    def _dense_init(self, config, device, max_seq, seed):
        self.cfg = config
        self.dev = torch.device(device)
        _prime_graph_state(self.dev)
        self.H, self.I = config["hidden_size"], config["intermediate_size"]
        self.nh, self.nkv = config["num_heads"], config["num_kv_heads"]
        self.kvd = self.nkv * 128
        self.nb, self.vocab, self.max_seq = config["n_block"], config["vocab_size"], max_seq
        self.eps = float(config.get("rms_norm_eps", EPS))
        self.theta = float(config.get("rope_theta", ROPE_THETA))
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        dev = self.dev
        f16 = dict(dtype=torch.float16, device=dev)
        self.blocks = []
        for _ in range(self.nb):
            blk = {}
            for name in config["linear"]:
                n, k = config["linear_shape"][name]
                blk[name] = (torch.randn(n, k, device=dev, generator=gen) * (0.5 / math.sqrt(k))).to(torch.float16)
            blk["ln1"] = (1.0 + 0.05 * torch.randn(self.H, device=dev, generator=gen)).to(torch.float16)
            blk["ln2"] = (1.0 + 0.05 * torch.randn(self.H, device=dev, generator=gen)).to(torch.float16)
            blk["kc"] = torch.zeros(self.B, self.nkv, max_seq, 128, **f16)
            blk["vc"] = torch.zeros(self.B, self.nkv, max_seq, 128, **f16)
            self.blocks.append(blk)
        self.embed = torch.randn(self.vocab, self.H, device=dev, generator=gen).to(torch.float16)
        self.lm_head = (torch.randn(self.vocab, self.H, device=dev, generator=gen) / math.sqrt(self.H)).to(torch.float16)
        self.norm = (1.0 + 0.05 * torch.randn(self.H, device=dev, generator=gen)).to(torch.float16)
        self.x = torch.zeros(self.B, self.H, **f16)
        self.att = torch.zeros(self.B, self.H, **f16)
        self.logits = torch.zeros(self.vocab, **f16) if self.B == 1 else torch.zeros(self.B, self.vocab, **f16)
        self.token = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.rope_tab = ops.rope_table(max_seq, self.theta, dev)
        # step state: cos/sin row of self.pos + the position itself in one block (set_token / the step's tail keep it)
        self.rope_cur, self.pos, self.step_err = ops.new_step_state(dev)
        self.rope_cur.copy_(self.rope_tab.view(max_seq, 128)[0])
        self.graph = None
        self.host_pos = 0
        self.suppress = torch.full((8,), -1, dtype=torch.int32, device=dev)
        self.engine = None           # (the one-launch-per-token engine serves the quantized runner only)

    def linear_bytes_per_token(self):
        return sum(blk[name].numel() * 2 for blk in self.blocks for name in self.cfg["linear"])

    def _step(self):
        F = torch.nn.functional
        x = self.x
        for blk in self.blocks:
            h = ops.rmsnorm(x, blk["ln1"], self.eps)
            q = F.linear(h, blk["self_attn.q_proj"])
            k = F.linear(h, blk["self_attn.k_proj"])
            v = F.linear(h, blk["self_attn.v_proj"])
            ops.attn_decode(q, k, v, blk["kc"], blk["vc"], self.att, self.pos, self.nh, self.nkv, self.theta, cur=self.rope_cur)
            x = x + F.linear(self.att, blk["self_attn.o_proj"])
            h2 = ops.rmsnorm(x, blk["ln2"], self.eps)
            x = x + F.linear(F.silu(F.linear(h2, blk["mlp.gate_proj"])) * F.linear(h2, blk["mlp.up_proj"]),
                             blk["mlp.down_proj"])
        ops.gemv_f16w(x.reshape(-1).contiguous() if self.B == 1 else x.contiguous(), self.lm_head, gamma=self.norm, eps=self.eps,
                      out=self.logits)
        ops.decode_tail(self.logits, self.embed, self.token, self.pos, self.x, table=self.rope_tab, cur=self.rope_cur, suppress=self.suppress)

    def _rows_linear(self, w, inp, residual=None):
        if residual is None:
            return torch.nn.functional.linear(inp, w)
        return torch.addmm(residual, inp, w.t(), out=residual)

    def _rows_up_gated(self, w, inp, gate):
        return ops.silu_mul(gate, torch.nn.functional.linear(inp, w), out=gate)

    def _prefill_eager(self, ids, start_pos=0, b=None):
        F = torch.nn.functional
        S = ids.numel()
        if start_pos + S > self.max_seq:
            raise ValueError("prompt longer than the KV cache")
        nh, nkv = self.nh, self.nkv
        x = self.embed.index_select(0, ids.to(self.dev))
        for blk in self.blocks:
            h = ops.rmsnorm(x, blk["ln1"], self.eps)
            q, k, v = F.linear(h, blk["self_attn.q_proj"]), F.linear(h, blk["self_attn.k_proj"]), F.linear(h, blk["self_attn.v_proj"])
            kc, vc = self._cache_rows(blk, b)
            ops.rope_cache(q, k, v, kc[0], vc[0], self.rope_tab, start_pos, nh, nkv)
            x = torch.addmm(x, self._prefill_attention(q, blk, S, start_pos, b), blk["self_attn.o_proj"].t())
            h2 = ops.rmsnorm(x, blk["ln2"], self.eps)
            g, u = F.linear(h2, blk["mlp.gate_proj"]), F.linear(h2, blk["mlp.up_proj"])
            x = torch.addmm(x, ops.silu_mul(g, u, out=g), blk["mlp.down_proj"].t())
        return self._prefill_finish(x, S, start_pos, b)


def get_memory_footprint(model, return_buffers=True):
    """bytes of weights + buffers (amq_speed_benchmark.py:88-95 counts parameters and buffers); takes a runner or, as the reference
    does, a ``torch.nn.Module`` (e.g. the swapped HF model: HIPQuantLinear keeps its weights in buffers)"""
    if isinstance(model, torch.nn.Module):
        mem = sum(p.nelement() * p.element_size() for p in model.parameters())
        return mem + (sum(b.nelement() * b.element_size() for b in model.buffers()) if return_buffers else 0)
    tot = model.embed.numel() * 2 + model.lm_head.numel() * 2 + model.norm.numel() * 2
    for blk in model.blocks:
        for k, v in blk.items():
            if isinstance(v, torch.Tensor):
                tot += v.numel() * v.element_size()
            elif hasattr(v, "nbytes"):
                tot += v.nbytes()
    return tot
