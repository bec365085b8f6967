"""Tensor-level wrappers over the C ABI (one function per entry point).

All tensors must live on a HIP device; inputs are validated on the host before
a kernel is launched (shape/dtype/contiguity), because an out-of-bounds access
in a hand-written kernel can take the GPU down.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (MODE_HQQ, MODE_FMA, MODE_FMA1, PRO_NONE, PRO_RMSNORM, PRO_SILU_MUL, Segment, GemvOpts, EngineBlock, EngineLinear,  # noqa: F401
                   GEMM_AUTO, GEMM_TILED, GEMM_SKINNY, GEMM_RING, GEMM_RING128, GEMM_WS, GEMM_DEQ, MATH_DEFAULT, MATH_EXACT, MATH_LINEAR, MATH_GROUPSCALE)

GROUP = 128


def _need(t, dtype, name, numel=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a tensor on the GPU")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name}: expected {numel} elements, got {t.numel()}")


FINE_GROUPS = (64, 32)      # groups finer than the native tile: 128 / group (scale, zero) pairs per (row, tile) in the native meta


def native_group(group):
    """granularity of the NATIVE meta for a source-format group size: 128 for 128 and its multiples (pairs replicated), else the group"""
    return int(group) if int(group) in FINE_GROUPS else GROUP


def native_sizes(bits, N, K, group=GROUP):
    lib = _lib.load()
    return int(lib.amq_native_qweight_bytes(bits, N, K)), int(lib.amq_native_meta_bytes(N, K, native_group(group)))


def alloc_native(bits, N, K, device, group=GROUP, meta_dtype=torch.float16):
    qb, mb = native_sizes(bits, N, K, group)
    return (torch.empty(qb // 4, dtype=torch.int32, device=device),
            torch.empty(mb // 2, dtype=meta_dtype, device=device))


def _check_shape(bits, N, K):
    if bits not in (2, 3, 4):
        raise ValueError(f"bits must be 2, 3 or 4 (got {bits})")
    if N % 16 or K % 128 or N <= 0 or K <= 0:
        raise ValueError(f"need N % 16 == 0 and K % 128 == 0 (got N={N}, K={K})")


def _check_group(group, K):
    """source-format group sizes the repack kernels read: 128 or a multiple of it that divides K (each group's (scale, zero) is
    replicated into the native layout's per-128 pairs), or 64 / 32 (128 / group pairs per native tile row)"""
    group = int(group)
    if group in FINE_GROUPS:
        return group
    if group < GROUP or group % GROUP or K % group:
        raise ValueError(f"group size must be 32, 64 or a multiple of {GROUP} that divides K={K} (got {group})")
    return group


def repack_from_hqq(W_q, scale, zero, bits, N, K, group=GROUP):
    """HQQLinear.W_q + meta['scale'|'zero'] -> (qweight_native, meta_native), MODE_HQQ."""
    _check_shape(bits, N, K)
    group = _check_group(group, K)
    lib = _lib.load()
    R = N * K // group
    if bits == 3:
        _need(W_q, torch.int32, "W_q", ((R + 9) // 10) * group)
    else:
        _need(W_q, torch.uint8, "W_q", R * group * bits // 8)
    # compute_dtype = bfloat16 models (quantize.py:516): the repack copies 16-bit patterns, the native meta is then a bfloat16 tensor and only the
    # *_bf16 entry points take it (the tensor's dtype is the tag the C ABI does not carry)
    meta_dtype = torch.bfloat16 if isinstance(scale, torch.Tensor) and scale.dtype == torch.bfloat16 else torch.float16
    if meta_dtype == torch.bfloat16 and group in FINE_GROUPS:
        raise ValueError("bfloat16 meta: groups of 128 (and multiples) only")
    _need(scale, meta_dtype, "scale", R)
    _need(zero, meta_dtype, "zero", R)
    qn, mn = alloc_native(bits, N, K, W_q.device, group, meta_dtype)
    _lib.check(lib.amq_repack_from_hqq(bits, _lib.ptr(W_q), _lib.ptr(scale), _lib.ptr(zero), N, K, group,
                                       _lib.ptr(qn), _lib.ptr(mn), _lib.current_stream()))
    return qn, mn


def repack_from_gptq(qweight, scales, zeros, bits, N, K, group=GROUP):
    """GPTQLinear buffers -> native, MODE_FMA."""
    _check_shape(bits, N, K)
    group = _check_group(group, K)
    lib = _lib.load()
    _need(qweight, torch.int32, "qweight", K // 32 * bits * N)
    _need(scales, torch.float32, "scales", K // group * N)
    _need(zeros, torch.float32, "zeros", K // group * N)
    qn, mn = alloc_native(bits, N, K, qweight.device, group)
    _lib.check(lib.amq_repack_from_gptq(bits, _lib.ptr(qweight), _lib.ptr(scales), _lib.ptr(zeros), N, K, group,
                                        _lib.ptr(qn), _lib.ptr(mn), _lib.current_stream()))
    return qn, mn


def repack_from_awq(qweight, scales, scaled_zeros, N, K, group=GROUP):
    """FT_QuantLinear buffers (4 bit) -> native, MODE_FMA."""
    _check_shape(4, N, K)
    group = _check_group(group, K)
    if N % 4 or K % 64:
        raise ValueError("AWQ pack needs N % 4 == 0 and K % 64 == 0")
    lib = _lib.load()
    _need(qweight, torch.int16, "qweight", N // 4 * K)
    _need(scales, torch.float16, "scales", K // group * N)
    _need(scaled_zeros, torch.float16, "scaled_zeros", K // group * N)
    qn, mn = alloc_native(4, N, K, qweight.device, group)
    _lib.check(lib.amq_repack_from_awq(_lib.ptr(qweight), _lib.ptr(scales), _lib.ptr(scaled_zeros), N, K, group,
                                       _lib.ptr(qn), _lib.ptr(mn), _lib.current_stream()))
    return qn, mn


def fma_mode_for(mn, bits):
    """AMQ_MODE_FMA1 for a native meta buffer whose scales allow the GEMV kernel's one-op unpack (|scale| <= amq_fma1_scale_bound(bits); the same
    weights bit for bit, ~5 % faster decode launches), else AMQ_MODE_FMA.  One device -> host read: call at load time, not per forward."""
    bound = float(_lib.load().amq_fma1_scale_bound(int(bits)))
    smax = float(mn.view(-1, 2)[:, 0].abs().max().item()) if mn.numel() else 0.0
    return MODE_FMA1 if smax <= bound else MODE_FMA


def _check_native(qn, mn, bits, N, K, fine=False):
    """validates a native (payload, meta) pair and returns its group granularity: 128, or -- where the caller serves them (``fine``) --
    64 / 32, recognised by the meta tensor's size (128 / group pairs per tile row)"""
    qb, mb = native_sizes(bits, N, K)
    _need(qn, torch.int32, "qweight_native", qb // 4)
    if isinstance(mn, torch.Tensor) and mn.numel() != mb // 2 and mn.numel() in (mb, 2 * mb):
        group = GROUP * (mb // 2) // mn.numel()
        if not fine:
            raise ValueError(f"meta_native holds groups of {group}: this entry point serves groups of 128 (and multiples); "
                             f"groups of 64 / 32 run through gemv / gemv_grouped (<= 16 rows), gemm and dequantize")
        _need(mn, torch.float16, "meta_native", mn.numel())
        return group
    _need(mn, torch.float16, "meta_native", mb // 2)
    return GROUP


def dequantize(qn, mn, bits, mode, N, K, out=None):
    _check_shape(bits, N, K)
    group = _check_native(qn, mn, bits, N, K, fine=True)
    if out is None:
        out = torch.empty(N, K, dtype=torch.float16, device=qn.device)
    _need(out, torch.float16, "out", N * K)
    _lib.check(_lib.load().amq_dequantize_f16(bits, mode, _lib.ptr(qn), _lib.ptr(mn), N, K, group,
                                              _lib.ptr(out), _lib.current_stream()))
    return out


def dequantize_hqq(W_q, scale, zero, bits, N, K, group=GROUP):
    _check_shape(bits, N, K)
    group = _check_group(group, K)
    R = N * K // group
    if bits == 3:
        _need(W_q, torch.int32, "W_q", ((R + 9) // 10) * group)
    else:
        _need(W_q, torch.uint8, "W_q", R * group * bits // 8)
    if isinstance(scale, torch.Tensor) and scale.dtype == torch.bfloat16:      # a compute_dtype = bfloat16 layer: dequantized in bf16, as the reference does
        _need(scale, torch.bfloat16, "scale", R)
        _need(zero, torch.bfloat16, "zero", R)
        out = torch.empty(N, K, dtype=torch.bfloat16, device=W_q.device)
        _lib.check(_lib.load().amq_dequantize_hqq_bf16(bits, _lib.ptr(W_q), _lib.ptr(scale), _lib.ptr(zero), N, K, group,
                                                       _lib.ptr(out), _lib.current_stream()))
        return out
    _need(scale, torch.float16, "scale", R)
    _need(zero, torch.float16, "zero", R)
    out = torch.empty(N, K, dtype=torch.float16, device=W_q.device)
    _lib.check(_lib.load().amq_dequantize_hqq_f16(bits, _lib.ptr(W_q), _lib.ptr(scale), _lib.ptr(zero), N, K, group,
                                                  _lib.ptr(out), _lib.current_stream()))
    return out


def _prep_x(x, K):
    if x.dtype != torch.float16:
        raise ValueError(f"x: expected float16, got {x.dtype}")
    if x.shape[-1] != K:
        raise ValueError(f"x: last dim {x.shape[-1]} != K={K}")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    return x2


def gemv(x, qn, mn, bits, mode, N, K, bias=None, out=None, opts=None):
    """y = x . W^T for few rows (weight-streaming kernels).  ``opts``: a :class:`GemvOpts` (per-call A/B / math options)."""
    if opts is None:
        opts = DEFAULT_GEMV_OPTS
    if opts is not None:
        x2 = _prep_x(x, K)
        y = out if out is not None else torch.empty(x2.shape[0], N, dtype=torch.float16, device=x.device)
        gemv_grouped(x2, [dict(qn=qn, mn=mn, bits=bits, mode=mode, N=N, y=y, bias=bias)], K, opts=opts)
        return y.reshape(*x.shape[:-1], N)
    _check_shape(bits, N, K)
    group = _check_native(qn, mn, bits, N, K, fine=True)
    x2 = _prep_x(x, K)
    M = x2.shape[0]
    if bias is not None:
        _need(bias, torch.float16, "bias", N)
    y = out if out is not None else torch.empty(M, N, dtype=torch.float16, device=x.device)
    _need(y, torch.float16, "y", M * N)
    _lib.check(_lib.load().amq_gemv_f16(bits, mode, _lib.ptr(x2), _lib.ptr(qn), _lib.ptr(mn), _lib.ptr(bias),
                                        _lib.ptr(y), M, N, K, group, 0, 0, _lib.current_stream()))
    return y.reshape(*x.shape[:-1], N)


class _ScratchPool:
    """Grow-only device scratch, one buffer per (device, stream).  A buffer that has been handed out is NEVER freed:
    a captured hipGraph (QuantLlama.prefill keeps one per prompt length) has the raw pointer baked in, so when a larger
    request arrives the old block is retired into ``_keep`` instead of being dropped -- replaying an older graph then
    still writes memory this pool owns.  Keyed by stream so that launches on different streams never share partials."""

    def __init__(self, dtype):
        self.dtype = dtype
        self._cur = {}
        self._keep = []

    def get(self, device, numel):
        key = (device.index if device.index is not None else torch.cuda.current_device(),
               torch.cuda.current_stream(device).cuda_stream)
        t = self._cur.get(key)
        if t is None or t.numel() < numel:
            if t is not None:
                self._keep.append(t)
            t = self._cur[key] = torch.empty(numel, dtype=self.dtype, device=device)
        return t


_SPLITK_WS = _ScratchPool(torch.float32)


def _splitk_workspace(device, nbytes):
    """scratch for split-K partials (see _ScratchPool)"""
    return _SPLITK_WS.get(device, (nbytes + 3) // 4)


# OPT-IN comparison leg for many rows: from LIB_GEMM_ROWS rows on (0 = never, the default) `gemm` runs the bit-exact dequantize
# kernel into a scratch and hands the plain fp16 GEMM to the library (torch.matmul -> hipBLASLt) -- what the reference itself
# does from 128 rows on (GPTQLinear.forward: torch unpack + matmul, hqq/backends/autogptq.py:245-283).  The product path is
# hand-written at every size: the fused kernels below ~6k rows, and from there the same split as the reference's with BOTH halves
# hand-written (GEMM_DEQ: dequantize kernel + amq_gemm_f16.hip; profiles/r04_gemm_f16pp.txt: 1.33-1.45 PFLOP/s at BASELINE configs[3]
# sizes against 1.27-1.32 for the fused ring kernel and 1.42-1.52 for dequantize + hipBLASLt, faster than the library up to 8192 rows);
# bench.py reports the library route next to it.
LIB_GEMM_ROWS = 0
_DEQ_SCRATCH = _ScratchPool(torch.float16)


def _dequant_scratch(device, numel):
    """fp16 scratch for one dequantized weight matrix (see _ScratchPool)"""
    return _DEQ_SCRATCH.get(device, numel)[:numel]


def _route_workspace(lib, device, route, M, N, K, group=GROUP):
    """(tensor or None, bytes) for ``amq_gemm_route_f16`` / ``amq_gemm_gated_f16``: split-K partials for few rows, the dequantized
    fp16 weights for the dequantize-once route (GEMM_DEQ, GEMM_AUTO on MFMA-bound launches, and launches of more than 256 rows over groups of 64 / 32)."""
    need = lib.amq_gemm_route_workspace_bytes_g(route, M, N, K, group)
    if not need:
        return None, 0
    if need == N * K * 2 and (route == GEMM_DEQ or route == GEMM_AUTO):
        return _dequant_scratch(device, N * K), need
    ws = _splitk_workspace(device, need)
    return ws, ws.numel() * 4


def gemm_route_name(M, N=13824, K=5120):
    """what :func:`gemm` runs for an M-row launch under the current settings (for result files)"""
    if LIB_GEMM_ROWS and M >= LIB_GEMM_ROWS:
        return "amq::dequant_kernel + library GEMM (torch.matmul -> hipBLASLt)"
    if _lib.load().amq_gemm_route_workspace_bytes(GEMM_AUTO, M, N, K) == N * K * 2:
        return ("amq::dequant_native_kernel + amq::gemm_f16_pp_kernel (dequantize once into a scratch, then a hand-written fp16 MFMA GEMM: "
                "256x256 tiles, two wave groups in ping-pong, persistent tiles, no unpack in the K loop)")
    return ("amq::gemm_ring_kernel / amq::gemm_ws_kernel / amq::gemm_kernel (fused unpack + MFMA, hand-written; 256x256 ring tiles when they "
            "fill the chip, 256x128 wave-specialised tiles where those fill it better)")


def gemm(x, qn, mn, bits, mode, N, K, bias=None, out=None, residual=None, route=GEMM_AUTO, gate=None):
    """y = x . W^T for any number of rows: few-row kernel, tiled MFMA kernel (split-K when few rows would leave the chip
    idle), or -- from LIB_GEMM_ROWS rows -- dequantize kernel + library GEMM.
    ``residual`` (fp16 [M, N], may be ``out``) is added in the epilogue: y = residual + fp16(x . W^T (+ bias))
    (the library path rounds the sum once).  ``gate`` (fp16 [M, N] contiguous, may be ``out``; not with ``residual``):
    y = fp16(silu(gate)) * fp16(x . W^T (+ bias)), i.e. :func:`silu_mul` of the two projections with this one as ``up``.
    ``route`` != GEMM_AUTO forces one hand-written kernel family (tests, tools)."""
    _check_shape(bits, N, K)
    group = _check_native(qn, mn, bits, N, K, fine=True)
    x2 = _prep_x(x, K)
    M = x2.shape[0]
    if bias is not None:
        _need(bias, torch.float16, "bias", N)
    if residual is not None:
        _need(residual, torch.float16, "residual", M * N)
    if gate is not None:
        if residual is not None:
            raise ValueError("gate and residual are exclusive")
        _need(gate, torch.float16, "gate", M * N)
    y = out if out is not None else torch.empty(M, N, dtype=torch.float16, device=x.device)
    _need(y, torch.float16, "y", M * N)
    lib = _lib.load()
    if gate is not None:
        ws, ws_bytes = _route_workspace(lib, x.device, route, M, N, K, group)
        if y.data_ptr() == gate.data_ptr() and not lib.amq_gemm_gated_fused_g(route, M, N, K, 1 if ws_bytes else 0, group):
            # the tiled kernel cannot apply the gate itself: in place on the gate needs the projection somewhere else first
            up = gemm(x, qn, mn, bits, mode, N, K, bias=bias, route=route)
            silu_mul(gate, up.view(-1), out=y)
            return y.reshape(*x.shape[:-1], N)
        _lib.check(lib.amq_gemm_gated_f16(route, bits, mode, _lib.ptr(x2), _lib.ptr(qn), _lib.ptr(mn), _lib.ptr(bias),
                                          _lib.ptr(gate), _lib.ptr(y), M, N, K, group, 0, _lib.ptr(ws),
                                          ws_bytes, _lib.current_stream()))
        return y.reshape(*x.shape[:-1], N)
    if route == GEMM_AUTO and LIB_GEMM_ROWS and M >= LIB_GEMM_ROWS:
        w = dequantize(qn, mn, bits, mode, N, K, out=_dequant_scratch(x.device, N * K).view(N, K))
        y2 = y.view(M, N)
        if residual is not None:
            r2 = residual.view(M, N)
            if r2.data_ptr() == y2.data_ptr():
                y2.addmm_(x2, w.t())
            else:
                torch.addmm(r2, x2, w.t(), out=y2)
            if bias is not None:
                y2.add_(bias)
        elif bias is not None:
            torch.addmm(bias, x2, w.t(), out=y2)
        else:
            torch.matmul(x2, w.t(), out=y2)
        return y.reshape(*x.shape[:-1], N)
    ws, ws_bytes = _route_workspace(lib, x.device, route, M, N, K, group)
    _lib.check(lib.amq_gemm_route_f16(route, bits, mode, _lib.ptr(x2), _lib.ptr(qn), _lib.ptr(mn), _lib.ptr(bias),
                                      _lib.ptr(residual), _lib.ptr(y), M, N, K, group, 0, 0, _lib.ptr(ws),
                                      ws_bytes, _lib.current_stream()))
    return y.reshape(*x.shape[:-1], N)


def gemm_res_norm_xfrag(x, qn, mn, bits, mode, N, K, gamma, eps, bias=None, residual=None, out=None, xf_out=None):
    """:func:`gemm` (``residual`` fused, dense result rows) followed by :func:`rmsnorm_xfrag` of the result, as one call: returns (y, xf).  Where the
    GEMM runs split-K -- a short prompt's down_proj -- the sum over the splits and the norm are one launch (include/amq_hip.h:
    amq_gemm_res_norm_xfrag_f16); same bits as the two calls."""
    _check_shape(bits, N, K)
    group = _check_native(qn, mn, bits, N, K, fine=True)
    x2 = _prep_x(x, K)
    M = x2.shape[0]
    if bias is not None:
        _need(bias, torch.float16, "bias", N)
    if residual is not None:
        _need(residual, torch.float16, "residual", M * N)
    _need(gamma, torch.float16, "gamma", N)
    y = out if out is not None else torch.empty(M, N, dtype=torch.float16, device=x.device)
    _need(y, torch.float16, "y", M * N)
    lib = _lib.load()
    nbytes = lib.amq_xfrag_bytes(M, N)
    if nbytes == 0:
        raise ValueError(f"N = {N}: the normed rows are handed on in fragment order (N % 128 == 0)")
    xf = xf_out if xf_out is not None else torch.empty(nbytes // 2, dtype=torch.float16, device=x.device)
    _need(xf, torch.float16, "xf", nbytes // 2)
    ws, ws_bytes = _route_workspace(lib, x.device, GEMM_AUTO, M, N, K, group)
    _lib.check(lib.amq_gemm_res_norm_xfrag_f16(bits, mode, _lib.ptr(x2), _lib.ptr(qn), _lib.ptr(mn), _lib.ptr(bias), _lib.ptr(residual),
                                               _lib.ptr(y), M, N, K, group, 0, _lib.ptr(ws), ws_bytes, _lib.ptr(gamma), float(eps),
                                               _lib.ptr(xf), _lib.current_stream()))
    return y.reshape(*x.shape[:-1], N), xf


def gemm_f16w(x, w, bias=None, out=None, residual=None, gate=None):
    """y = x . w^T for DENSE fp16 weights w [N, K] (hand-written ping-pong MFMA kernel, amq_gemm_f16.hip): what GEMM_DEQ runs behind
    :func:`dequantize`; ``residual`` / ``gate`` as in :func:`gemm`."""
    _need(w, torch.float16, "w")
    N, K = w.shape
    x2 = _prep_x(x, K)
    M = x2.shape[0]
    y = out if out is not None else torch.empty(M, N, dtype=torch.float16, device=x.device)
    _need(y, torch.float16, "y", M * N)
    for t, nm, n in ((bias, "bias", N), (residual, "residual", M * N), (gate, "gate", M * N)):
        if t is not None:
            _need(t, torch.float16, nm, n)
    _lib.check(_lib.load().amq_gemm_f16w_f16(_lib.ptr(x2), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(gate),
                                             _lib.ptr(y), M, N, K, 0, 0, _lib.current_stream()))
    return y.reshape(*x.shape[:-1], N)


def xfrag(src, M, K, stride_m=None, stride_kt=128, out=None):
    """Fragment-ordered copy of M x K activations (see include/amq_hip.h: amq_xfrag_f16).  ``src`` is any fp16 tensor whose
    element (m, k) sits at m*stride_m + (k // 128)*stride_kt + k % 128 (default: row-major [M, K])."""
    if not isinstance(src, torch.Tensor) or not src.is_cuda or src.dtype != torch.float16:
        raise ValueError("src: expected an fp16 tensor on the GPU")
    lib = _lib.load()
    nbytes = lib.amq_xfrag_bytes(M, K)
    xf = out if out is not None else torch.empty(nbytes // 2, dtype=torch.float16, device=src.device)
    _need(xf, torch.float16, "xf", nbytes // 2)
    _lib.check(lib.amq_xfrag_f16(_lib.ptr(src), _lib.ptr(xf), M, K, K if stride_m is None else int(stride_m), int(stride_kt),
                                 _lib.current_stream()))
    return xf


def rmsnorm_xfrag(x, gamma, eps, out=None):
    """:func:`rmsnorm` of x [M, K] with the result in fragment order (input of :func:`gemm_xfrag`)."""
    M, K = x.shape
    _need(x, torch.float16, "x")
    _need(gamma, torch.float16, "gamma", K)
    lib = _lib.load()
    nbytes = lib.amq_xfrag_bytes(M, K)
    xf = out if out is not None else torch.empty(nbytes // 2, dtype=torch.float16, device=x.device)
    _need(xf, torch.float16, "xf", nbytes // 2)
    _lib.check(lib.amq_rmsnorm_xfrag_f16(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(xf), M, K, float(eps), _lib.current_stream()))
    return xf


def gemm_xfrag(xf, M, qn, mn, bits, mode, N, K, bias=None, out=None, residual=None, gate=None):
    """y[M, N] = x . W^T with x given in fragment order (:func:`xfrag`); ``residual`` as in :func:`gemm`;
    ``gate`` (fp16 [M, N], may be ``out``): y = fp16(silu(gate)) * fp16(x . W^T (+ bias)), i.e. :func:`silu_mul` fused."""
    _check_shape(bits, N, K)
    _check_native(qn, mn, bits, N, K)
    _need(xf, torch.float16, "xf", _lib.load().amq_xfrag_bytes(M, K) // 2)
    if bias is not None:
        _need(bias, torch.float16, "bias", N)
    if residual is not None:
        _need(residual, torch.float16, "residual", M * N)
    if gate is not None:
        _need(gate, torch.float16, "gate", M * N)
    y = out if out is not None else torch.empty(M, N, dtype=torch.float16, device=xf.device)
    _need(y, torch.float16, "y", M * N)
    _lib.check(_lib.load().amq_gemm_xfrag_f16(bits, mode, _lib.ptr(xf), _lib.ptr(qn), _lib.ptr(mn), _lib.ptr(bias),
                                              _lib.ptr(gate), _lib.ptr(residual), _lib.ptr(y), M, N, K, GROUP, 0,
                                              _lib.current_stream()))
    return y


def gemm_xfrag_grouped(xf, M, segments, K, form=0, blocks_per_wg=0):
    """One few-row launch for several linears over the same fragment-ordered x (q/k/v, gate/up of a prompt pass).
    segments: list of dicts {qn, mn, bits, mode, N, y, bias=None, residual=None} (y / residual: fp16 [M, N] contiguous).
    Per segment the result is :func:`gemm_xfrag`'s, bit for bit -- whatever the kernel form (``form``: _lib.FEWROW_AUTO / _TILE / _STREAM,
    ``blocks_per_wg`` with _STREAM: tests and A/B tools)."""
    if not 1 <= len(segments) <= _lib.MAX_SEGMENTS:
        raise ValueError(f"1..{_lib.MAX_SEGMENTS} segments")
    lib = _lib.load()
    _need(xf, torch.float16, "xf", lib.amq_xfrag_bytes(M, K) // 2)
    arr = (Segment * len(segments))()
    for i, s in enumerate(segments):
        _check_shape(s["bits"], s["N"], K)
        _check_native(s["qn"], s["mn"], s["bits"], s["N"], K)
        _need(s["y"], torch.float16, "y", M * s["N"])
        if s.get("bias") is not None:
            _need(s["bias"], torch.float16, "bias", s["N"])
        if s.get("residual") is not None:
            _need(s["residual"], torch.float16, "residual", M * s["N"])
        arr[i] = Segment(_lib.ptr(s["qn"]), _lib.ptr(s["mn"]), _lib.ptr(s.get("bias")), _lib.ptr(s.get("residual")),
                         _lib.ptr(s["y"]), s["N"], s["bits"], s["mode"], 0)
    _lib.check(lib.amq_gemm_xfrag_grouped_form_f16(arr, len(segments), _lib.ptr(xf), M, K, GROUP, int(form), int(blocks_per_wg), _lib.current_stream()))


def linear(x, qn, mn, bits, mode, N, K, bias=None):
    """Reference-style dispatch: few rows -> gemv family, otherwise gemm."""
    _check_shape(bits, N, K)
    group = _check_native(qn, mn, bits, N, K, fine=True)
    x2 = _prep_x(x, K)
    M = x2.shape[0]
    if bias is not None:
        _need(bias, torch.float16, "bias", N)
    if group != GROUP and M > 0:    # groups of 64 / 32: the GEMV kernel as far as it reaches (16 rows), then gemm (few-row kernel up to 256 rows, dequantize once + the fp16 GEMM beyond)
        if M <= gemv_max_rows(K):
            return gemv(x, qn, mn, bits, mode, N, K, bias=bias)
        return gemm(x, qn, mn, bits, mode, N, K, bias=bias)
    if M > 8:                       # tiled MFMA GEMM (split-K for few rows); amq_linear_f16 makes the same cut at 8 rows
        return gemm(x, qn, mn, bits, mode, N, K, bias=bias)
    y = torch.empty(M, N, dtype=torch.float16, device=x.device)
    if M == 0:
        return y.reshape(*x.shape[:-1], N)
    _lib.check(_lib.load().amq_linear_f16(bits, mode, _lib.ptr(x2), _lib.ptr(qn), _lib.ptr(mn), _lib.ptr(bias),
                                          _lib.ptr(y), M, N, K, GROUP, _lib.current_stream()))
    return y.reshape(*x.shape[:-1], N)


# ---- bfloat16 variants (include/amq_hip.h "bfloat16 variants"; csrc/amq_bf16.hip): models quantized with compute_dtype = torch.bfloat16
_DEQ_SCRATCH_BF16 = _ScratchPool(torch.bfloat16)


def _check_native_bf16(qn, mn, bits, N, K):
    qb, mb = native_sizes(bits, N, K)
    _need(qn, torch.int32, "qweight_native", qb // 4)
    if isinstance(mn, torch.Tensor) and mn.dtype == torch.float16:
        raise ValueError("meta_native is float16: this buffer was repacked from an fp16 model -- use the fp16 entry points")
    _need(mn, torch.bfloat16, "meta_native", mb // 2)


def dequantize_bf16(qn, mn, bits, N, K, out=None):
    """native (bf16 meta) -> W[N, K] bfloat16, bit-identical to Quantizer.dequantize under compute_dtype = bfloat16"""
    _check_shape(bits, N, K)
    _check_native_bf16(qn, mn, bits, N, K)
    if out is None:
        out = torch.empty(N, K, dtype=torch.bfloat16, device=qn.device)
    _need(out, torch.bfloat16, "out", N * K)
    _lib.check(_lib.load().amq_dequantize_bf16(bits, _lib.ptr(qn), _lib.ptr(mn), N, K, GROUP, _lib.ptr(out), _lib.current_stream()))
    return out


def linear_bf16(x, qn, mn, bits, N, K, bias=None, residual=None, out=None):
    """y = x . W^T (+ bias) (+ residual) in bfloat16: up to 16 rows the weight-streaming kernel, beyond dequantize once + the bf16 MFMA GEMM"""
    _check_shape(bits, N, K)
    _check_native_bf16(qn, mn, bits, N, K)
    if x.dtype != torch.bfloat16:
        raise ValueError(f"x: expected bfloat16, got {x.dtype}")
    if x.shape[-1] != K:
        raise ValueError(f"x: last dim {x.shape[-1]} != K={K}")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    if bias is not None:
        _need(bias, torch.bfloat16, "bias", N)
    if residual is not None:
        _need(residual, torch.bfloat16, "residual", M * N)
    y = out if out is not None else torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _need(y, torch.bfloat16, "y", M * N)
    if M == 0:
        return y.reshape(*x.shape[:-1], N)
    lib = _lib.load()
    need = int(lib.amq_gemm_bf16_workspace_bytes(M, N, K))
    ws = _DEQ_SCRATCH_BF16.get(x.device, need // 2) if need else None
    _lib.check(lib.amq_gemm_bf16(bits, _lib.ptr(x2), _lib.ptr(qn), _lib.ptr(mn), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(y),
                                 M, N, K, GROUP, 0, 0, _lib.ptr(ws), need, _lib.current_stream()))
    return y.reshape(*x.shape[:-1], N)


# Default launch options of the grouped GEMV (None = the library's defaults: exact math, auto geometry).  Only tools/
# set this (explicitly, from their own command lines) to run whole-model A/B experiments; no environment variable does.
DEFAULT_GEMV_OPTS = None


def default_gemv_math():
    """the arithmetic an ``opts=None`` GEMV launch runs in this build of the library (MATH_EXACT or MATH_GROUPSCALE)"""
    return int(_lib.load().amq_default_gemv_math())


def gemv_max_rows(K, plain=False, norm=True):
    """largest number of x rows the weight-streaming GEMV stages in LDS for this K (amq_query); ``plain``: with default options over groups of
    128 (launches of 2 .. 8 rows then run kernels with a smaller cross-wave sum buffer: one row more at K = 11008); ``norm=False``: and without
    an RMSNorm prologue (x may then be staged in two K phases: 8 rows at K = 11008, the 7B down_proj)"""
    out = (ctypes.c_int * 6)()
    _lib.load().amq_query(int(K), out, 6)
    return int((out[4] if norm else out[5]) if plain else out[0])


def gemv_grouped(x, segments, K, prologue=PRO_NONE, x2=None, gamma=None, eps=0.0, opts=None):
    """One launch for several linears sharing x.

    segments: list of dicts {qn, mn, bits, mode, N, y, bias=None, residual=None}
    (y / residual: fp16 [M, N] contiguous)."""
    xx = _prep_x(x, K)
    M = xx.shape[0]
    if not 1 <= len(segments) <= _lib.MAX_SEGMENTS:
        raise ValueError(f"1..{_lib.MAX_SEGMENTS} segments")
    arr = (Segment * len(segments))()
    group = None
    for i, s in enumerate(segments):
        _check_shape(s["bits"], s["N"], K)
        g = _check_native(s["qn"], s["mn"], s["bits"], s["N"], K, fine=True)
        if group is not None and g != group:
            raise ValueError(f"segments of one launch must share their group size (got {group} and {g})")
        group = g
        _need(s["y"], torch.float16, "y", M * s["N"])
        if s.get("bias") is not None:
            _need(s["bias"], torch.float16, "bias", s["N"])
        if s.get("residual") is not None:
            _need(s["residual"], torch.float16, "residual", M * s["N"])
        arr[i] = Segment(_lib.ptr(s["qn"]), _lib.ptr(s["mn"]), _lib.ptr(s.get("bias")), _lib.ptr(s.get("residual")),
                         _lib.ptr(s["y"]), s["N"], s["bits"], s["mode"], 0)
    if prologue == PRO_RMSNORM:
        _need(gamma, torch.float16, "gamma", K)
    if prologue == PRO_SILU_MUL:
        x2 = _prep_x(x2, K)
        if x2.shape[0] != M:
            raise ValueError("x2 rows != x rows")
    if opts is None:
        opts = DEFAULT_GEMV_OPTS
    _lib.check(_lib.load().amq_gemv_grouped_f16(arr, len(segments), _lib.ptr(xx), _lib.ptr(x2), _lib.ptr(gamma),
                                                ctypes.c_float(eps), prologue, M, K, group, 0,
                                                ctypes.byref(opts) if opts is not None else None, _lib.current_stream()))


# ---------------------------------------------------------------- decode-step surroundings
def gemv_grouped_sums(x, segments, K, gamma=None, eps=0.0, sums_in=None, sums_out=None):
    """5 .. 8 rows: :func:`gemv_grouped` whose RMSNorm takes the rows' sums of squares as per-16-column partials (``sums_in`` fp32 [M, K / 16], with
    ``gamma``) from the launch that produced x, and / or that leaves such partials of ITS output rows in ``sums_out`` (fp32 [M, N / 16]; one segment):
    include/amq_hip.h amq_gemv_grouped_sums_f16.  No pass over x for the statistic, no separate rmsnorm launch."""
    xx = _prep_x(x, K)
    M = xx.shape[0]
    if not 1 <= len(segments) <= _lib.MAX_SEGMENTS:
        raise ValueError(f"1..{_lib.MAX_SEGMENTS} segments")
    arr = (Segment * len(segments))()
    for i, s in enumerate(segments):
        _check_shape(s["bits"], s["N"], K)
        _check_native(s["qn"], s["mn"], s["bits"], s["N"], K)
        _need(s["y"], torch.float16, "y", M * s["N"])
        if s.get("bias") is not None:
            _need(s["bias"], torch.float16, "bias", s["N"])
        if s.get("residual") is not None:
            _need(s["residual"], torch.float16, "residual", M * s["N"])
        arr[i] = Segment(_lib.ptr(s["qn"]), _lib.ptr(s["mn"]), _lib.ptr(s.get("bias")), _lib.ptr(s.get("residual")),
                         _lib.ptr(s["y"]), s["N"], s["bits"], s["mode"], 0)
    if sums_in is not None:
        _need(gamma, torch.float16, "gamma", K)
        _need(sums_in, torch.float32, "sums_in", M * (K // 16))
    if sums_out is not None:
        _need(sums_out, torch.float32, "sums_out", M * (segments[0]["N"] // 16))
    _lib.check(_lib.load().amq_gemv_grouped_sums_f16(arr, len(segments), _lib.ptr(xx), _lib.ptr(gamma) if sums_in is not None else None,
                                                     ctypes.c_float(eps), _lib.ptr(sums_in), _lib.ptr(sums_out), M, K, GROUP, _lib.current_stream()))


def rmsnorm(x, gamma, eps, out=None):
    K = x.shape[-1]
    x2 = _prep_x(x, K)
    _need(gamma, torch.float16, "gamma", K)
    y = out if out is not None else torch.empty_like(x2)
    _need(y, torch.float16, "y", x2.numel())
    _lib.check(_lib.load().amq_rmsnorm_f16(_lib.ptr(x2), _lib.ptr(gamma), _lib.ptr(y), x2.shape[0], K,
                                           ctypes.c_float(eps), _lib.current_stream()))
    return y.reshape(x.shape)


def gemv_f16w(x, W, bias=None, gamma=None, eps=0.0, out=None):
    """y[N] = (RMSNorm'ed if gamma) x[K] . W[N,K]^T, fp16 weights (lm_head).  x [M, K] with 2 <= M <= 8 (batched decode):
    y [M, N], W streamed once for all rows."""
    N, K = W.shape
    _need(W, torch.float16, "W", N * K)
    M = x.shape[0] if x.dim() == 2 else 1
    _need(x, torch.float16, "x", M * K)
    if gamma is not None:
        _need(gamma, torch.float16, "gamma", K)
    if bias is not None:
        _need(bias, torch.float16, "bias", N)
    shape = (M, N) if x.dim() == 2 else (N,)
    y = out if out is not None else torch.empty(shape, dtype=torch.float16, device=x.device)
    _need(y, torch.float16, "y", M * N)
    if M == 1:
        _lib.check(_lib.load().amq_gemv_f16w(_lib.ptr(x), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(gamma),
                                             ctypes.c_float(eps), N, K, _lib.current_stream()))
    else:
        if M > 8:
            raise ValueError("at most 8 rows")
        _lib.check(_lib.load().amq_gemv_f16w_rows(_lib.ptr(x), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(gamma),
                                                  ctypes.c_float(eps), M, N, K, _lib.current_stream()))
    return y


def decode_tail(logits, embed, token, pos, x, table=None, cur=None, suppress=None):
    """token = argmax(logits), pos += 1, x = embed[token] (and cur = table[pos], the next step's cos/sin row) -- one
    launch (graph-capturable).  Batched decode: logits [B, vocab], token [B], x [B, hidden]; pos / cur advance once.
    suppress: int32 [8] device tensor of token ids never chosen (-1 = unused slot): HF's min_new_tokens treatment of the EOS ids."""
    vocab, hidden = embed.shape
    B = token.numel()
    _need(logits, torch.float16, "logits", B * vocab)
    _need(embed, torch.float16, "embed", vocab * hidden)
    _need(token, torch.int64, "token", B)
    _need(pos, torch.int32, "pos", 1)
    _need(x, torch.float16, "x", B * hidden)
    if cur is not None:
        _need(cur, torch.float16, "rope_cur", 128)
        _need(table, torch.float16, "rope table")
    tab = _lib.ptr(table) if cur is not None else None
    rows = table.numel() // 128 if cur is not None else 0
    if suppress is not None:
        _need(suppress, torch.int32, "suppress", 8)
        _lib.check(_lib.load().amq_decode_tail_suppress_f16(_lib.ptr(logits), vocab, _lib.ptr(embed), hidden, _lib.ptr(token), _lib.ptr(pos),
                                                            _lib.ptr(x), tab, _lib.ptr(cur), rows, B, _lib.ptr(suppress), _lib.current_stream()))
        return
    if B == 1:
        _lib.check(_lib.load().amq_decode_tail_f16(_lib.ptr(logits), vocab, _lib.ptr(embed), hidden, _lib.ptr(token), _lib.ptr(pos),
                                                   _lib.ptr(x), tab, _lib.ptr(cur), rows, _lib.current_stream()))
    else:
        _lib.check(_lib.load().amq_decode_tail_batch_f16(_lib.ptr(logits), vocab, _lib.ptr(embed), hidden, _lib.ptr(token),
                                                         _lib.ptr(pos), _lib.ptr(x), tab, _lib.ptr(cur), rows, B,
                                                         _lib.current_stream()))


def set_token(token_in, embed, token, pos, x, table=None, cur=None):
    """token = token_in (int64 CUDA tensor: one id, or one per sequence), x = embed[token], cur = table[pos] -- one launch (amq_set_token_f16); pos unchanged"""
    vocab, hidden = embed.shape
    B = token.numel()
    n_in = token_in.numel()
    _need(token_in, torch.int64, "token_in", n_in)
    _need(embed, torch.float16, "embed", vocab * hidden)
    _need(token, torch.int64, "token", B)
    _need(pos, torch.int32, "pos", 1)
    _need(x, torch.float16, "x", B * hidden)
    if cur is not None:
        _need(cur, torch.float16, "rope_cur", 128)
        _need(table, torch.float16, "rope table")
    _lib.check(_lib.load().amq_set_token_f16(_lib.ptr(token_in), n_in, _lib.ptr(embed), vocab, hidden, _lib.ptr(token), _lib.ptr(pos), _lib.ptr(x),
                                             _lib.ptr(table) if cur is not None else None, _lib.ptr(cur), table.numel() // 128 if cur is not None else 0, B,
                                             _lib.current_stream()))


def rope_cache(q, k, v, kcache, vcache, table, pos0, n_heads, n_kv_heads):
    """Prefill glue: rotate q [S, n_heads*128] in place, rotate k [S, n_kv_heads*128] into kcache[h, pos0+s], copy v into
    vcache (both [n_kv_heads, max_seq, 128]); ``table`` from :func:`rope_table`.
    Caches [B, n_kv_heads, max_seq, 128] (4-D): q / k / v hold B sequences of S = rows / B rows each, one launch for all."""
    if kcache.dim() == 4:
        B = kcache.shape[0]
        rows = q.shape[0]
        if rows % B or kcache.shape[1] != n_kv_heads or kcache.shape[3] != 128 or vcache.shape != kcache.shape:
            raise ValueError("caches must be [B, n_kv_heads, max_seq, 128] and q rows a multiple of B")
        S = rows // B
        _need(q, torch.float16, "q", rows * n_heads * 128)
        _need(k, torch.float16, "k", rows * n_kv_heads * 128)
        _need(v, torch.float16, "v", rows * n_kv_heads * 128)
        _need(kcache, torch.float16, "kcache")
        _need(vcache, torch.float16, "vcache")
        _need(table, torch.float16, "rope table")
        _lib.check(_lib.load().amq_rope_cache_batch_f16(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(kcache), _lib.ptr(vcache),
                                                        _lib.ptr(table), table.numel() // 128, int(pos0), S, B, n_heads, n_kv_heads,
                                                        128, kcache.shape[2], _lib.current_stream()))
        return
    S = q.shape[0]
    _need(q, torch.float16, "q", S * n_heads * 128)
    _need(k, torch.float16, "k", S * n_kv_heads * 128)
    _need(v, torch.float16, "v", S * n_kv_heads * 128)
    if kcache.dim() != 3 or kcache.shape[0] != n_kv_heads or kcache.shape[2] != 128 or vcache.shape != kcache.shape:
        raise ValueError("caches must be [n_kv_heads, max_seq, 128]")
    _need(kcache, torch.float16, "kcache")
    _need(vcache, torch.float16, "vcache")
    _need(table, torch.float16, "rope table")
    _lib.check(_lib.load().amq_rope_cache_f16(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(kcache), _lib.ptr(vcache),
                                              _lib.ptr(table), table.numel() // 128, int(pos0), S, n_heads, n_kv_heads, 128,
                                              kcache.shape[1], _lib.current_stream()))


def attn_prefill(q, k, v, out, S, n_heads, n_kv_heads, batch=1, pos0=0, kv_cache=False, out_xfrag=False):
    """Causal attention over a prompt (amq_attn_prefill_f16).  q / out: fp16 [batch*S, n_heads*128] (q rotated).
    kv_cache=False: k / v are projection outputs [batch*S, n_kv_heads*128] (rotated keys); kv_cache=True: k / v are cache
    tensors [batch, n_kv_heads, max_seq, 128] whose rows 0 .. pos0+S-1 are valid.
    out_xfrag (batch 1): ``out`` is an :func:`xfrag` buffer of the result (None: allocated) -- what :func:`gemm_xfrag` reads."""
    H = n_heads * 128
    _need(q, torch.float16, "q", batch * S * H)
    if out_xfrag:
        if batch != 1:
            raise ValueError("out_xfrag needs batch == 1")
        nb = _lib.load().amq_xfrag_bytes(S, H)
        if out is None:
            out = torch.empty(nb // 2, dtype=torch.float16, device=q.device)
        _need(out, torch.float16, "out_xf", nb // 2)
    else:
        _need(out, torch.float16, "out", batch * S * H)
    if kv_cache:
        if k.dim() != 4 or k.shape[0] != batch or k.shape[1] != n_kv_heads or k.shape[3] != 128 or v.shape != k.shape:
            raise ValueError("caches must be [batch, n_kv_heads, max_seq, 128]")
        max_seq = k.shape[2]
        if pos0 + S > max_seq:
            raise ValueError(f"keys 0..{pos0 + S - 1} do not fit the cache (max_seq={max_seq})")
        _need(k, torch.float16, "kcache")
        _need(v, torch.float16, "vcache")
        kr, kb, kh = 128, n_kv_heads * max_seq * 128, max_seq * 128
    else:
        if pos0 != 0:
            raise ValueError("pos0 needs a cache")
        _need(k, torch.float16, "k", batch * S * n_kv_heads * 128)
        _need(v, torch.float16, "v", batch * S * n_kv_heads * 128)
        kr, kb, kh = n_kv_heads * 128, S * n_kv_heads * 128, 128
    if out_xfrag:
        _lib.check(_lib.load().amq_attn_prefill_xfrag_f16(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(out), S, int(pos0), n_heads,
                                                          n_kv_heads, 128, H, kr, kh, kr, kh, _lib.current_stream()))
        return out
    _lib.check(_lib.load().amq_attn_prefill_f16(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(out), batch, S, int(pos0), n_heads,
                                                n_kv_heads, 128, H, S * H, kr, kb, kh, kr, kb, kh, H, S * H, _lib.current_stream()))
    return out


def rope_rows(q, k, table, seq_len, n_heads, n_kv_heads, pos0=0):
    """RoPE in place on q [rows, n_heads*128] and k [rows, n_kv_heads*128], rows = batch * seq_len (no cache)."""
    rows = q.shape[0]
    _need(q, torch.float16, "q", rows * n_heads * 128)
    _need(k, torch.float16, "k", rows * n_kv_heads * 128)
    _need(table, torch.float16, "rope table")
    _lib.check(_lib.load().amq_rope_rows_f16(_lib.ptr(q), _lib.ptr(k), _lib.ptr(table), table.numel() // 128, int(pos0), rows,
                                             int(seq_len), n_heads, n_kv_heads, 128, _lib.current_stream()))


def silu_mul(gate, up, out=None):
    """fp16(silu(gate)) * up (LlamaMLP activation), the same expression as the GEMV SiLU prologue."""
    _need(gate, torch.float16, "gate")
    _need(up, torch.float16, "up", gate.numel())
    y = out if out is not None else torch.empty_like(gate)
    _need(y, torch.float16, "out", gate.numel())
    _lib.check(_lib.load().amq_silu_mul_f16(_lib.ptr(gate), _lib.ptr(up), _lib.ptr(y), gate.numel(), _lib.current_stream()))
    return y


def new_step_state(device):
    """-> (rope_cur fp16 [128], pos int32 [1], err int32 [1]): three views of one 264-byte block, the layout
    amq_attn_decode_cur_f16 reads.  ``err`` is the sticky error word the attention kernel raises when the device-side
    position is outside the cache (see :func:`check_step_state`)."""
    block = torch.zeros(66, dtype=torch.int32, device=device)
    return block[:64].view(torch.float16), block[64:65], block[65:66]


def check_step_state(err):
    """raise if a decode step ran with its device-side position outside the KV cache (synchronises)"""
    if int(err.item()) != 0:
        raise _lib.AmqError("a decode step ran with its position outside the KV cache (step skipped on the device)")


def gemv_qkv_attn(x, segments, K, gamma, eps, kcache, vcache, out, cur, n_heads, n_kv_heads, tickets):
    """q / k / v GEMV (RMSNorm prologue) + decode attention of one block as ONE launch (include/amq_hip_ab.h: amq_gemv_qkv_attn_f16; an A/B route in libamq_hip_ab.so).
    segments: three dicts {qn, mn, bits, mode, N, y} (q, k, v; y fp16 [N]); caches [1, n_kv_heads, max_seq, 128]; ``cur``: the
    fp16 [128] view of a step-state block; ``tickets``: int32 [>= n_heads], zero (left zero)."""
    xx = _prep_x(x, K)
    if xx.shape[0] != 1 or len(segments) != 3:
        raise ValueError("one row, three segments (q, k, v)")
    arr = (Segment * 3)()
    for i, s in enumerate(segments):
        _check_shape(s["bits"], s["N"], K)
        _check_native(s["qn"], s["mn"], s["bits"], s["N"], K)
        _need(s["y"], torch.float16, "y", s["N"])
        arr[i] = Segment(_lib.ptr(s["qn"]), _lib.ptr(s["mn"]), None, None, _lib.ptr(s["y"]), s["N"], s["bits"], s["mode"], 0)
    max_seq = kcache.shape[2]
    _need(gamma, torch.float16, "gamma", K)
    _need(kcache, torch.float16, "kcache", n_kv_heads * max_seq * 128)
    _need(vcache, torch.float16, "vcache", n_kv_heads * max_seq * 128)
    _need(out, torch.float16, "out", n_heads * 128)
    _need(cur, torch.float16, "rope_cur", 128)
    if tickets.dtype != torch.int32 or tickets.numel() < n_heads or not tickets.is_cuda:
        raise ValueError("tickets: int32 [n_heads] on the GPU")
    _lib.check_ab(_lib.load_ab().amq_gemv_qkv_attn_f16(arr, _lib.ptr(xx), _lib.ptr(gamma), ctypes.c_float(eps), K, GROUP, _lib.ptr(kcache),
                                                 _lib.ptr(vcache), _lib.ptr(out), _lib.ptr(cur), n_heads, n_kv_heads, 128, max_seq,
                                                 _lib.ptr(tickets), _lib.current_stream()))


ENGINE_LINEARS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj",
                  "mlp.down_proj")


class DecodeEngine:
    """One decode token of a whole model as ONE persistent launch (include/amq_hip_ab.h: amq_decode_engine_f16, an A/B route in libamq_hip_ab.so; counterpart of
    the per-token loop of amq/kernel/monkeypatch/ftllama_modeling.py:167-230).

    blocks: list of dicts, one per decoder block: the seven linears under ENGINE_LINEARS as dicts {qn, mn, bits, mode, N}
    (native buffers), "ln1" / "ln2" fp16 [hidden], "kc" / "vc" fp16 [1, n_kv_heads, max_seq, 128] (batch 1).
    x: fp16 [hidden] residual stream (in / out); step_cur: the fp16 [128] view of a step-state block (new_step_state()).
    The device table, scratch and barrier words are owned by this object; step() only enqueues (graph-capturable)."""

    def __init__(self, blocks, hidden, inter, n_heads, n_kv_heads, max_seq, eps, x, step_cur, grid=0):
        lib = _lib.load_ab()
        dev = x.device
        nb = len(blocks)
        _need(x, torch.float16, "x", hidden)
        _need(step_cur, torch.float16, "rope_cur", 128)
        arr = (EngineBlock * nb)()
        keep = []
        for b, blk in enumerate(blocks):
            for i, name in enumerate(ENGINE_LINEARS):
                l = blk[name]
                K = inter if name == "mlp.down_proj" else hidden
                _check_shape(l["bits"], l["N"], K)
                _check_native(l["qn"], l["mn"], l["bits"], l["N"], K)
                if l["qn"].device != dev:
                    raise ValueError("all engine buffers must live on x's device")
                arr[b].lin[i] = EngineLinear(_lib.ptr(l["qn"]), _lib.ptr(l["mn"]), l["N"], l["bits"], l["mode"], 0)
                keep += [l["qn"], l["mn"]]
            _need(blk["ln1"], torch.float16, "ln1", hidden)
            _need(blk["ln2"], torch.float16, "ln2", hidden)
            for c in ("kc", "vc"):
                _need(blk[c], torch.float16, c, n_kv_heads * max_seq * 128)
            arr[b].ln1, arr[b].ln2 = _lib.ptr(blk["ln1"]), _lib.ptr(blk["ln2"])
            arr[b].kcache, arr[b].vcache = _lib.ptr(blk["kc"]), _lib.ptr(blk["vc"])
            keep += [blk["ln1"], blk["ln2"], blk["kc"], blk["vc"]]
        nbytes = int(lib.amq_decode_engine_image_bytes(nb))
        host = (ctypes.c_ubyte * nbytes)()
        _lib.check_ab(lib.amq_decode_engine_image(arr, nb, hidden, inter, n_heads, n_kv_heads, 128, GROUP, host))
        self.image = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
        self.scratch = torch.zeros(int(lib.amq_decode_engine_scratch_bytes(hidden, inter, n_kv_heads)), dtype=torch.uint8, device=dev)
        self.sync = torch.zeros(int(lib.amq_decode_engine_sync_bytes()) // 4, dtype=torch.int32, device=dev)
        self._keep = keep
        self.x, self.cur = x, step_cur
        self.args = (nb, hidden, inter, n_heads, n_kv_heads, 128, int(max_seq), ctypes.c_float(eps))
        self.grid = int(grid)
        self._grid_used = None

    def step(self):
        if self._grid_used != self.grid:         # the barrier words count arrivals of ONE grid size across launches
            self.sync.zero_()
            self._grid_used = self.grid
        _lib.check_ab(_lib.load_ab().amq_decode_engine_f16(_lib.ptr(self.image), *self.args, _lib.ptr(self.x), _lib.ptr(self.scratch),
                                                     self.scratch.numel(), _lib.ptr(self.cur), _lib.ptr(self.sync),
                                                     self.sync.numel() * 4, self.grid, _lib.current_stream()))

    def check(self):
        """raise if a barrier poll of an earlier step ran into its bound (synchronises)"""
        word = int(self.sync[(2 + 2 * 32 - 1) * 64].item())
        if word != 0:
            who = int(self.sync[(2 + 2 * 32 - 1) * 64 + 1].item())
            # recover here, not only in step(): a runner replays a captured hipGraph and never comes through step(), and the sticky error
            # word / the un-advanced epoch base would fail every later replay.  The caller drops its graph (QuantLlama.check).
            self.sync.zero_()
            self._grid_used = None
            raise _lib.AmqError("decode engine: a device-wide barrier timed out (barrier %d of the launch, first seen by workgroup %d)"
                                % (word & 0xFFFF, who))

    def vector(self, name):
        """fp16 view of one hand-off vector of the scratch buffer ('q', 'k', 'v', 'att', 'gate', 'up'): tests only"""
        nb, H, I, nh, nkv = self.args[0], self.args[1], self.args[2], self.args[3], self.args[4]
        sizes = [("q", H), ("k", nkv * 128), ("v", nkv * 128), ("att", H), ("gate", I), ("up", I)]
        off = 0
        for n, sz in sizes:
            if n == name:
                return self.scratch[off:off + 2 * sz].view(torch.float16)
            off += (2 * sz + 255) // 256 * 256
        raise KeyError(name)


def rope_table(max_seq, rope_theta, device, inv_freq=None, scale=1.0):
    """fp16 [max_seq, 64, 2] (cos, sin) rows.  ``inv_freq`` (fp32 [64]: HF's ``rotary_emb.inv_freq`` after a static rope_scaling -- Llama-3.1's
    "llama3" factors, arch.rope_inv_freq) and ``scale`` (its attention_scaling) replace the plain ``rope_theta`` frequencies."""
    tab = torch.empty(max_seq, 64, 2, dtype=torch.float16, device=device)
    if inv_freq is None:
        _lib.check(_lib.load().amq_rope_table_f16(_lib.ptr(tab), max_seq, ctypes.c_float(rope_theta), _lib.current_stream()))
    else:
        f = inv_freq.detach().to(device=device, dtype=torch.float32).contiguous()
        if f.numel() != 64:
            raise ValueError("inv_freq must hold 64 frequencies (head_dim 128)")
        _lib.check(_lib.load().amq_rope_table_freqs_f16(_lib.ptr(tab), max_seq, _lib.ptr(f), ctypes.c_float(scale), _lib.current_stream()))
    return tab


class _ZeroedPool:
    """Zeroed int32 ticket arrays (split decode attention), one per (device, stream), grow-only like :class:`_ScratchPool`:
    every launch leaves its tickets zero, so one array serves all launches of a stream."""

    def __init__(self):
        self._cur = {}
        self._keep = []

    def get(self, device, n):
        key = (device.index if device.index is not None else torch.cuda.current_device(),
               torch.cuda.current_stream(device).cuda_stream)
        t = self._cur.get(key)
        if t is None or t.numel() < n:
            if t is not None:
                self._keep.append(t)
            t = self._cur[key] = torch.zeros(max(n, 1024), dtype=torch.int32, device=device)
        return t


_ATTN_WS = _ScratchPool(torch.float32)
_ATTN_TICKETS = _ZeroedPool()
ATTN_SPLIT_FROM = 512      # caches longer than this use the split kernel (n_splits = 0 / auto)
ATTN_CHUNK = 272           # keys per workgroup aimed at for a full cache (the kernel holds up to 384 in registers; its smallest chunk is 256)
ATTN_CUS = 256             # workgroups per round of the chip (MI355X: 256 CUs)
ATTN_PREFETCH_KEYS = 384   # keys of a chunk the split kernel requests ahead into registers (amq_decode.hip); longer chunks take its remainder loop


ATTN_GQA_KEYS = 0          # keys per workgroup of the grouped-query kernel (0: by cache length, attn_decode_splits; a multiple of 128: A/B tools)


def attn_decode_splits(max_seq, n_heads=32, batch=1, n_kv_heads=None):
    """workgroups per head the auto policy gives a cache of ``max_seq`` rows (1: the single-workgroup kernel): about ATTN_CHUNK keys each, and --
    where the heads divide the CU count -- a whole number of rounds of the chip (7B at 2048 keys: 8 x 32 = 256 workgroups of 264 keys run the launch
    in 11.9 us, 6 x 32 of 352 keys in 11.8; at 4000 keys 16 x 32 in 18.1 against 11 x 32 in 19.6: profiles/r05_attn_decode_long.txt).
    Grouped-query models (``n_kv_heads`` given, 2 .. 16 query heads per kv head): workgroups per KV head of the grouped kernel (amq_attn_prefill.hip:
    attn_decode_gqa_kernel), chunks of 64 / 128 / 256 keys -- the longest that still gives the chip a round of workgroups."""
    if max_seq <= ATTN_SPLIT_FROM:
        return 1
    g = n_heads // n_kv_heads if n_kv_heads else 1
    if 2 <= g <= 16:
        # a round of the chip (workgroups = kv heads x sequences x splits), chunks of whole 128-key stages: 128 keys up to 2048 cached keys, at least 256
        # beyond (more, smaller chunks cost more in partial results than they gain: profiles/r06_attn_gqa.txt); ATTN_GQA_KEYS: A/B tools
        want = max(1, ATTN_CUS // (n_kv_heads * batch))
        if want > 32 and 32 * n_kv_heads * batch >= ATTN_CUS // 2:
            want = 32           # the combine launch takes 32 chunks in one round trip: 28 / 4 heads at 16384 keys 20.2 us with 64 chunks, 17.4 with 32
        per = ATTN_GQA_KEYS or max(128 if max_seq <= 2048 else 256, -(-max_seq // want))
        per = -(-per // 128) * 128
        return max(2, -(-max_seq // per))
    s = max(1, round(max_seq / ATTN_CHUNK))
    wg = max(1, n_heads * batch)
    if ATTN_CUS % wg == 0:
        step = ATTN_CUS // wg              # splits per round of the chip
        s = max(step, round(s / step) * step) if s >= step else s
        # rounding DOWN to a whole round may leave chunks beyond the kernel's 384-key register prefetch (32 heads at ~3000 - 3128 keys: 11 -> 8
        # splits of 416 keys, the slower remainder loop): go up a round instead
        while s >= step and (-(-max_seq // s) + 31) // 32 * 32 > ATTN_PREFETCH_KEYS:
            s += step
    return s


def attn_decode(q, k, v, kcache, vcache, out, pos, n_heads, n_kv_heads, rope_theta=10000.0, table=None, cur=None, n_splits=0):
    """One new token per sequence.  q [B, n_heads*128], k/v [B, n_kv_heads*128],
    caches [B, n_kv_heads, max_seq, 128]; ``pos`` is an int or a device int32 tensor.  ``cur``: fp16 [128] cos/sin row
    of the current position (maintained by decode_tail) -- needs ``pos`` as a device tensor.
    ``n_splits``: workgroups per head (include/amq_hip.h: amq_attn_decode_split_f16); 0 = by cache length, 1 = the
    single-workgroup kernel."""
    B = kcache.shape[0]
    max_seq = kcache.shape[2]
    if n_splits == 0:
        n_splits = attn_decode_splits(max_seq, n_heads, B, n_kv_heads)
    _need(q, torch.float16, "q", B * n_heads * 128)
    _need(k, torch.float16, "k", B * n_kv_heads * 128)
    _need(v, torch.float16, "v", B * n_kv_heads * 128)
    _need(kcache, torch.float16, "kcache", B * n_kv_heads * max_seq * 128)
    _need(vcache, torch.float16, "vcache", B * n_kv_heads * max_seq * 128)
    _need(out, torch.float16, "out", B * n_heads * 128)
    if isinstance(pos, torch.Tensor):
        _need(pos, torch.int32, "pos", 1)
        pos_dev, pos_i = ctypes.cast(pos.data_ptr(), ctypes.c_void_p), 0
    else:
        pos_dev, pos_i = None, int(pos)
        if not 0 <= pos_i < max_seq:
            raise ValueError(f"pos {pos_i} outside the cache (max_seq={max_seq})")
    if cur is not None:
        # step-state block: cos/sin row (256 bytes) immediately followed by the int32 position and the error word (new_step_state())
        _need(cur, torch.float16, "rope_cur", 128)
        if pos_dev is None or pos.data_ptr() != cur.data_ptr() + 256:
            raise ValueError("cur / pos must be the two views of one step-state block (ops.new_step_state)")
    if table is not None:
        _need(table, torch.float16, "rope table", max_seq * 128)
    if n_splits > 1:
        lib = _lib.load()
        wsb = lib.amq_attn_decode_split_workspace_bytes(B, n_heads, n_splits)
        ws = _ATTN_WS.get(q.device, wsb // 4)
        tk = _ATTN_TICKETS.get(q.device, B * n_heads)
        _lib.check(lib.amq_attn_decode_split_f16(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(kcache), _lib.ptr(vcache),
                                                 _lib.ptr(out), _lib.ptr(cur), pos_dev, pos_i, B, n_heads, n_kv_heads, 128,
                                                 max_seq, ctypes.c_float(rope_theta), _lib.ptr(table), n_splits,
                                                 _lib.ptr(ws), wsb, _lib.ptr(tk), _lib.current_stream()))
        return out
    if cur is not None:
        _lib.check(_lib.load().amq_attn_decode_cur_f16(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(kcache), _lib.ptr(vcache),
                                                       _lib.ptr(out), _lib.ptr(cur), B, n_heads, n_kv_heads, 128, max_seq,
                                                       _lib.current_stream()))
        return out
    _lib.check(_lib.load().amq_attn_decode_f16(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(kcache), _lib.ptr(vcache),
                                               _lib.ptr(out), pos_dev, pos_i, B, n_heads, n_kv_heads, 128, max_seq,
                                               ctypes.c_float(rope_theta), _lib.ptr(table), _lib.current_stream()))
    return out


# ---------------------------------------------------------------- the device of a launch
# Every launch goes to the CURRENT device's stream (``_lib.current_stream()``), kernel attributes and CU counts are the current device's too.  A
# process that keeps its weights on cuda:1 while cuda:0 is current (HF device_map, no set_device) must therefore have cuda:1 made current around
# the call -- the library's own guard cannot help when the stream it is handed is the null stream.  Done here, once, for every
# public entry point that takes tensors: the first CUDA tensor argument names the device; one comparison when it already is the current one.
def _on_tensor_device(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        for v in args:
            if isinstance(v, torch.Tensor) and v.is_cuda:
                if v.device.index != torch.cuda.current_device():
                    with torch.cuda.device(v.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapped


for _name in ("repack_from_hqq", "repack_from_gptq", "repack_from_awq", "dequantize", "dequantize_hqq", "dequantize_bf16", "linear_bf16", "gemv", "gemm", "gemm_f16w", "xfrag",
              "rmsnorm_xfrag", "gemm_xfrag", "gemm_xfrag_grouped", "linear", "gemv_grouped", "rmsnorm", "gemv_f16w", "decode_tail", "rope_cache",
              "attn_prefill", "rope_rows", "silu_mul", "gemv_qkv_attn", "attn_decode"):
    globals()[_name] = _on_tensor_device(globals()[_name])
del _name

