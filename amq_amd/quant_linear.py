"""HIPQuantLinear -- drop-in for the reference's two kernel-backed modules

    GPTQLinear      hqq/backends/autogptq.py:27-288   (2/3-bit, auto_gptq kernels)
    FT_QuantLinear  hqq/backends/ft.py:57-145         (4-bit, faster_transformer kernels)

Same surface: attributes ``bits, group_size, infeatures, outfeatures, bias,
name``; weights live in registered buffers so ``model.to(device)``,
``state_dict()/load_state_dict()`` and ``copy.deepcopy`` (amq_speed_benchmark.py:231)
work; ``forward(x)`` takes fp16 with any leading dims and returns
``x.shape[:-1] + (outfeatures,)`` with bias added.  One class serves all three
bit-widths: the weights are held in the MI355X-native AMQ-T16 layout
 and every forward goes through libamq_hip.so -- few rows take the
weight-streaming GEMV, many rows the tiled MFMA GEMM (the reference switches at
rows < 128, autogptq.py:163, and tokens < 8, ft.py:129).  There is no eager /
CPU fallback: without a GPU or the built library, forward raises.
"""
import weakref

import torch
import torch.nn as nn

from . import _ext, _lib, ops
from .hqq_format import GROUP, HQQWeights, from_hqq_layer, pack_rows

GROUP_MAX_ROWS = 8          # few-row forwards of grouped siblings run as ONE grouped GEMV launch (more rows: each member's GEMM)
# rows from which a fused MLP forms silu(gate) * up in its own launch instead of in down_proj's prologue: a fused prologue is redone by every
# workgroup of the launch, on both operands (profiles/r05_decode_batch.txt; QuantLlama.DOWN_FUSED_ROWS is the runner's copy of this rule)
DOWN_UNFUSED_FROM_ROWS = 2


class LinearGroup:
    """Sibling linears that read the SAME input -- q/k/v of an attention block, gate/up of an MLP -- as segments of ONE
    ``amq_gemv_grouped_f16`` launch (each with its own bit-width), without touching the parent module: HF's forward keeps calling
    ``self.q_proj(h)``, ``self.k_proj(h)``, ``self.v_proj(h)`` (modeling_llama.py; amq_speed_benchmark.py:231-256 only swaps the
    linears).  The first member called with a tensor launches all segments and keeps the siblings' outputs; the siblings' calls
    with that same tensor (same storage, version, shape) take theirs -- each output is handed out once.  The members stay
    ordinary modules with their own buffers, so state_dict keys, ``.to()`` and the reference's cache files are unchanged.
    What it buys (tools/module_walk_bench.py): 13 -> 10 launches per block for a module swap alone."""

    def __init__(self, members):
        self.members = list(members)
        if len({m.infeatures for m in self.members}) != 1:
            raise ValueError("grouped linears must share their input size")
        self._key = None
        self._xref = None
        self._outs = [None] * len(self.members)
        self._norm = None                       # a deferred RMSNorm waiting for this group's launch (HIPRMSNorm)
        self._h = None                          # (C++ launch handle, the qweight / meta tensors it was built over)
        for i, m in enumerate(self.members):
            m.__dict__["_group"] = (self, i)

    # -- the producer side of a fused RMSNorm (HIPRMSNorm): the norm hands its RAW input on and the group's launch forms the
    #    normalised row in its prologue (AMQ_PRO_RMSNORM)
    def accepts_norm(self, x):
        ms = self.members
        if x.shape[-1] != ms[0].infeatures:
            return False
        ext = _ext.get()
        if ext is not None:                     # (the handle was built over bias-free members on one device)
            return self._handle(ext) is not None and ms[0]._buffers["qweight"].device == x.device
        return not any(m.bias is not None for m in ms) and all(m.qweight.device == x.device for m in ms)

    def defer_norm(self, norm, x):
        _defer(self, norm, x)

    def __deepcopy__(self, memo):
        # (copy.deepcopy of the model: amq_speed_benchmark.py:231) the copy groups the COPIED members
        import copy
        new = LinearGroup.__new__(LinearGroup)
        memo[id(self)] = new
        new.members = [copy.deepcopy(m, memo) for m in self.members]
        new._key, new._outs, new._xref, new._norm, new._h = None, [None] * len(new.members), None, None, None
        for i, m in enumerate(new.members):
            m.__dict__["_group"] = (new, i)
        return new

    def _handle(self, ext):
        """the C++ launch handle of this group (``_amq_ext.Group``: weights checked once, held by the handle), rebuilt when a
        member's buffers were replaced (.to(), load_state_dict); None when the group cannot take the grouped launch"""
        ms = self.members
        h = self._h
        # (valid while the members still hold the buffer OBJECTS and the modes it was built over: load_state_dict copies INTO the same buffers and
        #  may change a member's mode -- meta would then be read as the wrong pair)
        if h is not None and all(m._buffers["qweight"] is q and m._buffers["meta"] is t for m, q, t in zip(ms, h[1], h[2])) \
                and h[3] == tuple(m.mode for m in ms):
            return h[0]
        if any(m.bias is not None for m in ms) or len({m.qweight.device for m in ms}) != 1:
            self._h = None
            return None
        qw, mt = [m.qweight for m in ms], [m.meta for m in ms]
        self._h = (ext.Group(qw, mt, [m.outfeatures for m in ms], [m.bits for m in ms], [m.mode for m in ms], ms[0].infeatures), qw, mt,
                   tuple(m.mode for m in ms))
        return self._h[0]

    def take(self, idx, x):
        """output of member ``idx`` for input ``x`` (fp16, GPU, <= GROUP_MAX_ROWS rows), launching the group if needed"""
        # "the same input": the same live tensor OBJECT (weak reference), same storage and shape, and -- where autograd tracks it
        # (inference tensors do not) -- the same version
        key = (x.data_ptr(), -1 if x.is_inference() else x._version, x.shape)
        same = self._key == key and self._xref is not None and self._xref() is x
        if not same or self._outs[idx] is None:
            ms = self.members
            ext = _ext.get()
            x2 = x if x.is_contiguous() else x.contiguous()
            if ext is not None:
                h = self._handle(ext)
                if h is None:
                    return None
                pro, gamma, eps = _claim_norm(self, x)
                outs = h.run(x2, pro, gamma, eps)
            else:
                if any(m.bias is not None for m in ms) or any(m.qweight.device != x.device for m in ms):
                    return None
                pro, gamma, eps = _claim_norm(self, x)
                outs = [torch.empty(x.shape[:-1] + (m.outfeatures,), dtype=torch.float16, device=x.device) for m in ms]
                ops.gemv_grouped(x2, [dict(qn=m.qweight, mn=m.meta, bits=m.bits, mode=m.mode, N=m.outfeatures, y=y.view(-1, m.outfeatures))
                                      for m, y in zip(ms, outs)], ms[0].infeatures, prologue=pro, gamma=gamma, eps=eps)
            self._outs = list(outs)
            self._key = key
            self._xref = weakref.ref(x)
        y = self._outs[idx]
        self._outs[idx] = None
        return y

    def __getstate__(self):
        # (pickling a whole model: torch.save(model)) launch handles, weak references and pending outputs are not state
        d = dict(self.__dict__)
        d.update(_key=None, _xref=None, _h=None, _norm=None, _outs=[None] * len(self.members))
        return d


def _defer(consumer, norm, x):
    """record that ``x`` is the RAW input of ``norm`` and that ``consumer``'s next launch has to normalise it"""
    p = consumer.__dict__.get("_norm")
    if p is not None:
        consumer.__dict__["_norm"] = None       # (reported once: a forward that died between the norm and its consumer must not brick the model)
        raise RuntimeError("a deferred RMSNorm output was never consumed by the grouped linears it was fused into "
                           "(the module that follows the norm does not read it through q/k/v or gate/up): "
                           "call prepare_for_inference(..., fuse_norms=False) for this model")
    consumer.__dict__["_norm"] = (norm, x.data_ptr(), x.numel())


def _claim_norm(consumer, x):
    """(prologue, gamma, eps) of ``consumer``'s launch over ``x``; a pending norm must be for exactly this tensor"""
    p = consumer.__dict__.get("_norm")
    if p is None:
        return 0, None, 0.0
    consumer.__dict__["_norm"] = None
    norm, ptr, numel = p
    if ptr != x.data_ptr() or numel != x.numel():
        raise RuntimeError("a deferred RMSNorm was followed by a forward over a different tensor than the norm's input")
    return ops.PRO_RMSNORM, norm.weight, float(norm.variance_epsilon)


class HIPRMSNorm(nn.Module):
    """LlamaRMSNorm (transformers modeling_llama.py; the reference swaps it for its FT kernel, kernel/monkeypatch/
    ftllama_modeling.py:39-46) fused into the launch that consumes it: for few rows (<= GROUP_MAX_ROWS, fp16, on the GPU) the
    forward returns its input UNCHANGED and tells its consumer -- the q/k/v LinearGroup for ``input_layernorm``, the
    HIPLlamaMLP for ``post_attention_layernorm`` -- to form weight * x / rms(x) in the prologue of its grouped GEMV
    (AMQ_PRO_RMSNORM, the arithmetic of amq_rmsnorm_f16).  Valid because in a Llama decoder layer the norm's output is read by
    those projections only; a norm whose deferred output is not consumed, or is followed by another tensor, raises.  Anything
    else (more rows, other dtypes) runs the wrapped module.  ``weight`` is the wrapped module's own Parameter, so state_dict
    keys are unchanged."""

    def __init__(self, inner, consumer, owner=None, slot=None):
        super().__init__()
        self.weight = inner.weight
        self.variance_epsilon = float(getattr(inner, "variance_epsilon", getattr(inner, "eps", 0.0)))
        self.__dict__["_inner"] = inner          # not registered: the state_dict keeps ``<norm>.weight`` only
        self.retarget(consumer, owner, slot)

    def retarget(self, consumer, owner=None, slot=None):
        """(re)bind the launch this norm is fused into.  ``owner`` (the decoder layer; kept in __dict__, not as a sub-module, so that
        deepcopy / pickling of the model carry it along) and ``slot`` ("qkv" | "mlp") let
        every forward verify that the consumer is STILL what follows the norm -- a sibling replaced after prepare_for_inference, or a
        second prepare_for_inference that regrouped q/k/v, leaves a stale group behind that would otherwise get the un-normalised x."""
        self.__dict__["_consumer"] = consumer
        self.__dict__["_owner"] = owner
        self.__dict__["_slot"] = slot

    def _consumer_is_current(self, c):
        owner = self.__dict__.get("_owner")
        if owner is None:
            return True                          # (bound by hand, no layer to check against)
        if self.__dict__.get("_slot") == "mlp":
            return getattr(owner, "mlp", None) is c
        attn = getattr(owner, "self_attn", None)
        ms = getattr(c, "members", None)
        if attn is None or ms is None or len(ms) != 3:
            return False
        for i, name in enumerate(("q_proj", "k_proj", "v_proj")):
            m = getattr(attn, name, None)
            g = m.__dict__.get("_group") if m is not None else None
            if m is not ms[i] or g is None or g[0] is not c:
                return False
        return True

    def forward(self, x):
        c = self.__dict__["_consumer"]
        K = self.weight.numel()
        if (x.dtype is torch.float16 and x.is_cuda and x.is_contiguous() and x.shape[-1] == K and 0 < x.numel() // K <= GROUP_MAX_ROWS
                and self.weight.dtype is torch.float16 and self.weight.device == x.device and self._consumer_is_current(c)
                and c.accepts_norm(x)):
            c.defer_norm(self, x)
            return x
        return self.__dict__["_inner"](x)

    def extra_repr(self):
        return f"{self.weight.numel()}, eps={self.variance_epsilon}, fused into {type(self.__dict__['_consumer']).__name__}"


class HIPQuantLinear(nn.Module):
    QUANT_TYPE = "hip-amq-t16"
    native_group = GROUP        # (instances set their own; the class default covers modules unpickled from before groups of 64 / 32 existed)

    def __init__(self, bits, group_size, infeatures, outfeatures, bias=False, name=None,
                 mode=ops.MODE_HQQ, weight_dtype=torch.float16):
        super().__init__()
        if bits not in [2, 3, 4]:
            raise NotImplementedError("Only 2,3,4 bits are supported.")     # autogptq.py:44-45
        group_size = group_size if group_size != -1 else infeatures
        if group_size not in ops.FINE_GROUPS and (group_size < GROUP or group_size % GROUP or infeatures % group_size):
            # 128 is what AMQ produces; coarser groups are read and their (scale, zero) replicated per 128 (native layout); 64 / 32 keep
            # 128 / group pairs per native tile row and run the GEMV kernel (<= 16 rows) or dequantize-once + the fp16 GEMM, unfused
            raise NotImplementedError("group_size must be 32, 64 or a multiple of 128 that divides infeatures.")
        if infeatures % 128 or outfeatures % 16:
            raise ValueError(f"need infeatures % 128 == 0 and outfeatures % 16 == 0 (got {infeatures}, {outfeatures})")
        # fp16 is what the reference's kernels take (ft.py:62); bfloat16 is this build's optional variant for models quantized with
        # compute_dtype = torch.bfloat16: HQQ arithmetic only, groups of 128 (and multiples), no sibling grouping / fusion (ops.linear_bf16)
        assert weight_dtype in (torch.float16, torch.bfloat16), "Only fp16 (and, for HQQ-format weights, bf16) is supported."
        if weight_dtype == torch.bfloat16 and (mode != ops.MODE_HQQ or group_size in ops.FINE_GROUPS):
            raise NotImplementedError("bfloat16 modules: HQQ dequant arithmetic and groups of 128 (or multiples) only")
        self.bits = bits
        self.group_size = group_size
        self.infeatures = infeatures
        self.outfeatures = outfeatures
        self.maxq = 2 ** bits - 1
        self.mode = mode
        self.name = name
        self.register_buffer("qweight", torch.zeros(infeatures * outfeatures * bits // 32, dtype=torch.int32))
        self.native_group = ops.native_group(group_size)           # granularity of the native meta: 128, or the group itself for 64 / 32
        self.register_buffer("meta", torch.zeros(infeatures // self.native_group * outfeatures * 2, dtype=weight_dtype))
        # dequant arithmetic travels with the weights (0: (q - z) * s two roundings, 1: fma(q, s, c))
        self.register_buffer("mode_flag", torch.tensor([mode], dtype=torch.int32))
        if bias is not None and bias is not False:
            self.register_buffer("bias", torch.zeros(outfeatures, dtype=weight_dtype))
            if isinstance(bias, torch.Tensor):
                self.bias.copy_(bias.detach().to(weight_dtype))
        else:
            self.bias = None

    @staticmethod
    def bf16_serves(group_size):
        """groups a bfloat16 module can hold (the bf16 kernels read one (scale, zero) pair per 128 columns)"""
        return group_size not in ops.FINE_GROUPS

    # ------------------------------------------------------------------ build
    def _set_native(self, qn, mn, mode):
        self.qweight = qn
        self.meta = mn
        self.mode = mode
        self.mode_flag = torch.tensor([mode], dtype=torch.int32, device=qn.device)
        if self.bias is not None:
            self.bias = self.bias.to(qn.device)

    @classmethod
    def from_hqq(cls, hqq, device=None, keep_bf16=True):
        """From an HQQLinear (reference object, duck-typed) or HQQWeights.  A layer quantized with compute_dtype = bfloat16 stays a bfloat16
        module (``keep_bf16``, and groups the bf16 kernels serve: 128 and multiples); otherwise its scale / zero go to fp16 -- what the reference's
        patch_hqq_to_gptq / patch_hqq_to_ft make of every layer (autogptq.py:301-306, ft.py:62) -- and the module takes the fp16 kernels."""
        h = hqq if isinstance(hqq, HQQWeights) else from_hqq_layer(hqq)
        if h.scale.dtype == torch.bfloat16 and not (keep_bf16 and cls.bf16_serves(h.group_size)):
            h = HQQWeights(h.W_q, h.scale.to(torch.float16), h.zero.to(torch.float16), h.nbits, h.shape, h.group_size,
                           None if h.bias is None else h.bias.to(torch.float16), h.name)
        n, k = h.shape
        dev = torch.device(device) if device is not None else h.W_q.device
        if dev.type != "cuda":
            raise RuntimeError("HIPQuantLinear.from_hqq needs a GPU device (repack runs as a HIP kernel)")
        mod = cls(h.nbits, h.group_size, k, n, bias=h.bias, name=h.name, weight_dtype=torch.bfloat16 if h.scale.dtype == torch.bfloat16 else torch.float16)
        qn, mn = ops.repack_from_hqq(h.W_q.to(dev).contiguous(), h.scale.to(dev).reshape(-1).contiguous(),
                                     h.zero.to(dev).reshape(-1).contiguous(), h.nbits, n, k, group=h.group_size)
        mod._set_native(qn, mn, ops.MODE_HQQ)
        return mod

    @classmethod
    def from_gptq_buffers(cls, qweight, scales, zeros, bits, bias=None, name=None):
        """From GPTQLinear buffers (autogptq.py:55-75): qweight int32 [K/32*bits, N],
        scales / zeros fp32 [K/G, N].  Keeps the reference kernels' arithmetic
        w = fma(q, s, -zeros) (auto_gptq_kernel.cu:206)."""
        n = qweight.shape[1]
        k = qweight.shape[0] * 32 // bits
        group = k // scales.shape[0]                       # scales [K / group, N]
        mod = cls(bits, group, k, n, bias=bias, name=name, mode=ops.MODE_FMA)
        qn, mn = ops.repack_from_gptq(qweight.contiguous(), scales.contiguous(), zeros.contiguous(), bits, n, k, group=group)
        mod._set_native(qn, mn, ops.fma_mode_for(mn, bits))        # MODE_FMA1 where the scales allow the one-op unpack (same weights, faster GEMV)
        return mod

    @classmethod
    def from_ft_buffers(cls, qweight, scales, scaled_zeros, bias=None, name=None):
        """From FT_QuantLinear buffers (ft.py:75-88): qweight int16 [N/4, K],
        scales / scaled_zeros fp16 [K/G, N].  w = fma(q, s, scaled_zeros) (gemv_cuda.cu:151)."""
        k = qweight.shape[1]
        n = qweight.shape[0] * 4
        group = k // scales.shape[0]
        mod = cls(4, group, k, n, bias=bias, name=name, mode=ops.MODE_FMA)
        qn, mn = ops.repack_from_awq(qweight.contiguous(), scales.contiguous(), scaled_zeros.contiguous(), n, k, group=group)
        mod._set_native(qn, mn, ops.fma_mode_for(mn, 4))
        return mod

    def pack(self, W, scales, zeros):
        """Signature of GPTQLinear.pack / FT_QuantLinear.pack (autogptq.py:111,
        ft.py:103): W = dequantized weight [N,K], scales / zeros = HQQ meta
        reshaped to [N, K/G].  The integers are recovered exactly as the
        reference does, ``round((W + z*s) / s)`` (autogptq.py:120), then packed
        on the GPU; the raw fp16 (scale, zero) are kept, so the module
        dequantizes like HQQ itself ((q - z) * s), not like the kernels' fma."""
        if W.device.type != "cuda":
            raise RuntimeError("HIPQuantLinear.pack needs GPU tensors")
        n, k = self.outfeatures, self.infeatures
        scale_zeros = zeros * scales
        s_rep = torch.repeat_interleave(scales, self.group_size, dim=1)
        sz_rep = torch.repeat_interleave(scale_zeros, self.group_size, dim=1)
        intweight = torch.round((W + sz_rep) / s_rep).to(torch.int32).clamp_(0, self.maxq)
        W_q = pack_rows(intweight.reshape(-1, self.group_size), self.bits)
        qn, mn = ops.repack_from_hqq(W_q.contiguous(), scales.to(torch.float16).reshape(-1).contiguous(),
                                     zeros.to(torch.float16).reshape(-1).contiguous(), self.bits, n, k, group=self.group_size)
        self._set_native(qn, mn, ops.MODE_HQQ)

    def post_init(self):
        pass

    def to_kernel_arithmetic(self):
        """Switch the module from HQQ's dequant arithmetic, w = fp16(fp16(q - z) * s) (Quantizer.dequantize, quantize.py:198: what ``backend='hip'``
        keeps), to the arithmetic of the reference's GPTQ / FT kernels, w = fp16(fma(q, s, -fp16(z * s))) -- what ``patch_hqq_to_gptq`` stores
        (``scale_zeros = zeros * scales``, autogptq.py:112-114) and ``vecquant*matmul`` / ``gemv_4bit`` compute (auto_gptq_kernel.cu:206,
        gemv_cuda.cu:151): one rounding per weight instead of two.  Distance from ``W_deq``: the rounding of c = fp16(z * s) carries over to every weight of
        the group, |dw| <= ~ulp(z * s) -- relative to |z * s|, so several ulps of a small weight (z ~ 7.5, q - z = 0.5: ~7) -- which IS the reference's
        ``backend='gptq'`` arithmetic, bit for bit (tests/test_gpu_hf.py).  The GEMV kernel then unpacks a weight pair
        with one packed op where the layer's scales allow it (``AMQ_MODE_FMA1``): ~5-8 % more decode tokens/s.  Idempotent; returns self."""
        if self.mode != ops.MODE_HQQ or self.is_bf16:       # (bf16 modules: the reference's kernels, whose arithmetic this is, are fp16-only)
            return self
        mt = self.meta.clone().view(-1, 2)                   # a NEW buffer: launch handles built over the old one notice and rebuild
        mt[:, 1] = -(mt[:, 1] * mt[:, 0])                    # fp16 product, as the reference forms it
        self._set_native(self.qweight, mt.reshape(-1), ops.fma_mode_for(mt.reshape(-1), self.bits))
        self.__dict__.pop("_ptrs_ok", None)
        grp = self.__dict__.get("_group")
        if grp is not None:
            grp[0]._h = None
        return self

    # ---------------------------------------------------------------- forward
    def _mode(self):
        return self.mode

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self.mode = int(self.mode_flag.item())

    def _buffer_ptrs(self):
        """Device pointers of the module's own buffers, validated ONCE per set of buffers: the decode loop of the reference
        harness is a walk over 225 module forwards per token (amq_speed_benchmark.py:231-256), so what a forward costs on
        the host is what the drop-in path costs.  Re-validated whenever the buffers moved (``.to()``, deepcopy,
        ``load_state_dict``): the cache is the tuple of data pointers itself (plain ints: deepcopy / pickle safe)."""
        key = (self.qweight.data_ptr(), self.meta.data_ptr(), None if self.bias is None else self.bias.data_ptr())
        if self.__dict__.get("_ptrs_ok") != key:
            ops._check_shape(self.bits, self.outfeatures, self.infeatures)
            ops._check_native(self.qweight, self.meta, self.bits, self.outfeatures, self.infeatures, fine=True)
            if self.bias is not None:
                ops._need(self.bias, torch.float16, "bias", self.outfeatures)
            self.__dict__["_ptrs_ok"] = key
        return key

    @property
    def is_bf16(self):
        return self.meta.dtype == torch.bfloat16     # (the buffer's dtype: survives state_dict / deepcopy / unpickling of older modules)

    def forward(self, x):
        x_dtype = x.dtype
        if self.is_bf16:                             # the optional bfloat16 variant: one call, any row count, no grouping
            y = ops.linear_bf16(x if x_dtype == torch.bfloat16 else x.to(torch.bfloat16), self.qweight, self.meta, self.bits,
                                self.outfeatures, self.infeatures, bias=self.bias)
            return y if x_dtype == torch.bfloat16 else y.to(x_dtype)
        if x_dtype != torch.float16:
            # the reference casts (with a warning) too: autogptq.py:166-169
            x = x.to(torch.float16)
        K, N = self.infeatures, self.outfeatures
        if x.shape[-1] != K:
            raise ValueError(f"x: last dim {x.shape[-1]} != K={K}")
        M = x.numel() // K
        if M > 8 or M == 0 or not x.is_cuda:        # many rows: the GEMM route (workspace handling lives in ops.gemm)
            out = ops.linear(x, self.qweight, self.meta, self.bits, self.mode, N, K, bias=self.bias)
            return out if x_dtype == torch.float16 else out.to(x_dtype)
        if x.device != self.qweight.device:
            raise ValueError(f"x is on {x.device} but the module's weights are on {self.qweight.device}")
        hold = self.__dict__.get("_residual")
        if hold is not None and hold[0] is not None:  # (fused decoder layer, patching.fuse_llama_layers) y = residual + fp16(x . W^T) in the epilogue
            y = self._forward_residual(x, hold[0])
            if y is not None:
                hold[0] = None                        # consumed: the caller does not add it again
                return y
        grp = self.__dict__.get("_group")
        # q/k/v, gate/up: one grouped launch for all siblings (LinearGroup) -- fp16 callers only: a cast input is a fresh tensor per sibling
        # call, the group would never recognise it and every sibling would launch the whole group
        if grp is not None and M <= GROUP_MAX_ROWS and x_dtype == torch.float16:
            y = grp[0].take(grp[1], x)
            if y is not None:
                return y if x_dtype == torch.float16 else y.to(x_dtype)
        ext = _ext.get()
        if ext is not None:                          # one C++ call: checks, output allocation, stream query, amq_linear_f16
            y = ext.linear(x if x.is_contiguous() else x.contiguous(), self.qweight, self.meta, self.bias, self.bits, self.mode, N, K)
            return y if x_dtype == torch.float16 else y.to(x_dtype)
        # few rows (decode) through ctypes: one call, no per-call re-validation of the module's own buffers
        qp, mp, bp = self._buffer_ptrs()
        x2 = x if x.is_contiguous() else x.contiguous()
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float16, device=x.device)
        # the stream of x's device (not of whatever device is current): the pointers above belong to that device
        rc = _lib.load().amq_linear_f16(self.bits, self.mode, x2.data_ptr(), qp, mp, bp, y.data_ptr(), M, N, K, self.native_group,
                                        _lib.stream_of(x.device))
        if rc != 0:
            _lib.check(rc)
        return y if x_dtype == torch.float16 else y.to(x_dtype)

    def _forward_residual(self, x, residual):
        """``residual + self(x)`` as ONE launch (the residual added in the GEMV's epilogue: the same two fp16 roundings as the
        separate add), or None when this call cannot take it (then the caller adds).  fp16 x on the GPU, <= 8 rows, no bias."""
        ext = _ext.get()
        if (ext is None or self.bias is not None or x.dtype is not torch.float16 or residual.dtype is not torch.float16
                or residual.shape[:-1] != x.shape[:-1] or residual.shape[-1] != self.outfeatures or residual.device != x.device
                or not residual.is_contiguous()):
            return None
        h = _LIN_HANDLES.get(self)
        if h is None or h[1] is not self._buffers["qweight"] or h[2] is not self._buffers["meta"] or h[3] != self.mode:
            h = _LIN_HANDLES[self] = (ext.Group([self.qweight], [self.meta], [self.outfeatures], [self.bits], [self.mode], self.infeatures),
                                      self.qweight, self.meta, self.mode)
        return h[0].run(x if x.is_contiguous() else x.contiguous(), 0, None, 0.0, residual)[0]

    def dequantize(self):
        """W[N,K] exactly as Quantizer.dequantize would give it (MODE_HQQ): fp16, or bfloat16 for a bfloat16 module."""
        if self.is_bf16:
            return ops.dequantize_bf16(self.qweight, self.meta, self.bits, self.outfeatures, self.infeatures)
        return ops.dequantize(self.qweight, self.meta, self.bits, self.mode, self.outfeatures, self.infeatures)

    def extra_repr(self):
        return (f"in_features={self.infeatures}, out_features={self.outfeatures}, bits={self.bits}, "
                f"group_size={self.group_size}, bias={self.bias is not None}, mode={self.mode}")


# HIPLlamaMLP / HIPQuantLinear -> C++ launch handles (kept OUTSIDE the modules: deepcopy / pickling of a model never meets a handle)
_MLP_HANDLES = weakref.WeakKeyDictionary()
_LIN_HANDLES = weakref.WeakKeyDictionary()


class HIPLlamaMLP(nn.Module):
    """LlamaMLP over three HIPQuantLinear children -- ``down_proj(act_fn(gate_proj(x)) * up_proj(x))`` with act_fn = SiLU
    (transformers modeling_llama.py) -- as TWO launches for few rows: gate / up as segments of one grouped GEMV, then down_proj
    with the SiLU * mul product formed in its prologue (``AMQ_PRO_SILU_MUL``: the same fp16 expression as the separate
    element-wise kernel, so the result is bit-identical to the unfused walk).  The reference's FT path replaces whole Llama
    sub-modules in the same way (kernel/monkeypatch/ftllama_modeling.py:39-46, 127-155).  The children keep their names, so
    state_dict keys (``mlp.gate_proj.qweight`` ...) and the cache files do not change.  More than GROUP_MAX_ROWS rows, biases
    or non-fp16 inputs take the plain composition."""

    def __init__(self, gate_proj, up_proj, down_proj):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = gate_proj, up_proj, down_proj

    def accepts_norm(self, x):
        """a fused post_attention_layernorm (HIPRMSNorm) may hand its raw input on: exactly the fast path's condition"""
        g_, u_, d_ = self.gate_proj, self.up_proj, self.down_proj
        return (x.shape[-1] == g_.infeatures and g_.bias is None and u_.bias is None and d_.bias is None
                and x.device == g_.qweight.device == u_.qweight.device)

    def defer_norm(self, norm, x):
        _defer(self, norm, x)

    def forward(self, x, residual=None):
        """``residual`` (optional, fused decoder layers): returns residual + mlp(x), the add formed in down_proj's epilogue"""
        g_, u_, d_ = self.gate_proj, self.up_proj, self.down_proj
        K = g_.infeatures
        if residual is not None and not (residual.dtype is torch.float16 and residual.is_contiguous() and residual.device == x.device
                                         and residual.shape[:-1] == x.shape[:-1] and residual.shape[-1] == d_.outfeatures):
            return residual + self.forward(x)
        if (x.dtype is torch.float16 and x.is_cuda and x.shape[-1] == K and 0 < x.numel() // K <= GROUP_MAX_ROWS
                and g_.bias is None and u_.bias is None and d_.bias is None and x.device == g_.qweight.device):
            x2 = x if x.is_contiguous() else x.contiguous()
            ext = _ext.get()
            pro, gamma, eps = _claim_norm(self, x)
            if ext is not None:
                h = _MLP_HANDLES.get(self)       # (C++ launch handles: weights checked once; rebuilt when a buffer was replaced)
                if h is None or any(m._buffers["qweight"] is not q or m._buffers["meta"] is not t for m, q, t in zip((g_, u_, d_), h[2], h[3])) \
                        or h[4] != (g_.mode, u_.mode, d_.mode):
                    qw, mt = [g_.qweight, u_.qweight, d_.qweight], [g_.meta, u_.meta, d_.meta]
                    h = _MLP_HANDLES[self] = (ext.Group(qw[:2], mt[:2], [g_.outfeatures, u_.outfeatures], [g_.bits, u_.bits], [g_.mode, u_.mode], K),
                                              ext.Group(qw[2:], mt[2:], [d_.outfeatures], [d_.bits], [d_.mode], d_.infeatures), qw, mt,
                                              (g_.mode, u_.mode, d_.mode))
                g, u = h[0].run(x2, pro, gamma, eps)
                if x2.numel() // K >= DOWN_UNFUSED_FROM_ROWS:      # (the same values: silu_mul_kernel's expression is the prologue's)
                    return h[1].run(ext.silu_mul(g, u), 0, None, 0.0, residual)[0]
                return h[1].run(g, 2, u, 0.0, residual)[0]
            I = g_.outfeatures
            g = torch.empty(x.shape[:-1] + (I,), dtype=torch.float16, device=x.device)
            u = torch.empty_like(g)
            ops.gemv_grouped(x2, [dict(qn=g_.qweight, mn=g_.meta, bits=g_.bits, mode=g_.mode, N=I, y=g.view(-1, I)),
                                  dict(qn=u_.qweight, mn=u_.meta, bits=u_.bits, mode=u_.mode, N=I, y=u.view(-1, I))], K,
                             prologue=pro, gamma=gamma, eps=eps)
            y = torch.empty(x.shape[:-1] + (d_.outfeatures,), dtype=torch.float16, device=x.device)
            dseg = [dict(qn=d_.qweight, mn=d_.meta, bits=d_.bits, mode=d_.mode, N=d_.outfeatures, y=y.view(-1, d_.outfeatures),
                         residual=None if residual is None else residual.view(-1, d_.outfeatures))]
            if x2.numel() // K >= DOWN_UNFUSED_FROM_ROWS:
                ops.gemv_grouped(ops.silu_mul(g.view(-1, I), u.view(-1, I), out=g.view(-1, I)), dseg, I)
            else:
                ops.gemv_grouped(g.view(-1, I), dseg, I, prologue=ops.PRO_SILU_MUL, x2=u.view(-1, I))
            return y
        if self.__dict__.get("_norm") is not None:
            raise RuntimeError("a deferred RMSNorm reached HIPLlamaMLP's unfused path")
        out = d_(torch.nn.functional.silu(g_(x)) * u_(x))
        return out if residual is None else residual + out
