"""ctypes binding of libamq_hip.so (C ABI: include/amq_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call
fails, an exception is raised.  PyTorch is used only for device memory and the
current HIP stream.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product always loads THIS file.  A/B build variants (`make variant`) are loaded by tools/ only, through
# ``use_library(path)`` called explicitly before the first ``load()`` -- no environment variable changes what the product runs.
LIB_PATH = os.path.join(_HERE, "libamq_hip.so")
# the A/B routes of the decode step (include/amq_hip_ab.h: one launch per token, q/k/v + attention in one launch): built, bit-identical,
# slower -- their own library (`make -C amq_amd/csrc ab`), loaded only by ops.DecodeEngine / ops.gemv_qkv_attn
AB_LIB_PATH = os.path.join(_HERE, "libamq_hip_ab.so")

AMQ_OK = 0
MODE_HQQ, MODE_FMA, MODE_FMA1 = 0, 1, 2
PRO_NONE, PRO_RMSNORM, PRO_SILU_MUL = 0, 1, 2
MAX_SEGMENTS = 4
MATH_DEFAULT, MATH_LINEAR, MATH_GROUPSCALE, MATH_EXACT = 0, 1, 2, 3
FEWROW_AUTO, FEWROW_TILE, FEWROW_STREAM = 0, 1, 2      # kernel form of the grouped few-row launch (amq_gemm_xfrag_grouped_form_f16)
ABI_VERSION = 521            # include/amq_hip.h AMQ_VERSION these bindings mirror (checked at load)
GEMM_AUTO, GEMM_TILED, GEMM_SKINNY, GEMM_RING, GEMM_RING128, GEMM_WS, GEMM_DEQ = 0, 1, 2, 3, 4, 5, 6

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t


class Segment(ctypes.Structure):
    """mirror of `amq_segment` (include/amq_hip.h)"""
    _fields_ = [("qweight_native", _vp), ("meta_native", _vp), ("bias", _vp), ("residual", _vp),
                ("y", _vp), ("N", _i), ("bits", _i), ("mode", _i), ("y_stride", _i)]


class EngineLinear(ctypes.Structure):
    """mirror of `amq_engine_linear` (include/amq_hip.h)"""
    _fields_ = [("qweight_native", _vp), ("meta_native", _vp), ("N", _i), ("bits", _i), ("mode", _i), ("reserved", _i)]


class EngineBlock(ctypes.Structure):
    """mirror of `amq_engine_block`: lin = q, k, v, o, gate, up, down"""
    _fields_ = [("lin", EngineLinear * 7), ("ln1", _vp), ("ln2", _vp), ("kcache", _vp), ("vcache", _vp)]


class GemvOpts(ctypes.Structure):
    """mirror of `amq_gemv_opts` (include/amq_hip.h): per-call launch options, all zero = defaults"""
    _fields_ = [("math", _i), ("waves", _i), ("depth", _i), ("rpt", _i), ("dot", _i)]


# name -> (restype, argtypes); must list every symbol include/amq_hip.h declares
SIGNATURES = {
    "amq_version": (_i, []),
    "amq_default_gemv_math": (_i, []),
    "amq_last_error": (ctypes.c_char_p, []),
    "amq_query": (_i, [_i, ctypes.POINTER(_i), _i]),
    "amq_native_qweight_bytes": (_sz, [_i, _i, _i]),
    "amq_fma1_scale_bound": (ctypes.c_float, [_i]),
    "amq_native_meta_bytes": (_sz, [_i, _i, _i]),
    "amq_repack_from_hqq": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "amq_repack_from_gptq": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "amq_repack_from_awq": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "amq_dequantize_f16": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "amq_dequantize_hqq_f16": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "amq_dequantize_bf16": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "amq_dequantize_hqq_bf16": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "amq_gemv_bf16": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "amq_gemm_bf16_workspace_bytes": (_sz, [_i, _i, _i]),
    "amq_gemm_bf16": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "amq_gemv_f16": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "amq_gemm_f16": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "amq_linear_f16": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "amq_gemm_splitk_workspace_bytes": (_sz, [_i, _i, _i]),
    "amq_gemm_splitk_f16": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "amq_compat_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "amq_vecquantmatmul_faster_old": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "amq_gemv_4bit": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "amq_gemm_4bit": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "amq_rmsnorm_f16": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "amq_gemv_f16w": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "amq_attn_decode_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "amq_rope_table_f16": (_i, [_vp, _i, _f, _vp]),
    "amq_rope_table_freqs_f16": (_i, [_vp, _i, _vp, _f, _vp]),
    "amq_decode_tail_f16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "amq_attn_decode_cur_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "amq_gemm_res_f16": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "amq_xfrag_bytes": (_sz, [_i, _i]),
    "amq_xfrag_f16": (_i, [_vp, _vp, _i, _i, ctypes.c_longlong, ctypes.c_longlong, _vp]),
    "amq_rmsnorm_xfrag_f16": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "amq_gemm_xfrag_f16": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "amq_gemm_gated_fused": (_i, [_i, _i, _i, _i, _i]),
    "amq_gemm_gated_fused_g": (_i, [_i, _i, _i, _i, _i, _i]),
    "amq_gemm_gated_f16": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "amq_attn_decode_split_workspace_bytes": (_sz, [_i, _i, _i]),
    "amq_attn_decode_split_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _sz, _vp, _vp]),
    "amq_gemm_xfrag_grouped_f16": (_i, [ctypes.POINTER(Segment), _i, _vp, _i, _i, _i, _vp]),
    "amq_gemm_xfrag_grouped_form_f16": (_i, [ctypes.POINTER(Segment), _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "amq_attn_prefill_xfrag_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [ctypes.c_longlong] * 5 + [_vp]),
    "amq_gemv_f16w_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp]),
    "amq_decode_tail_batch_f16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "amq_decode_tail_suppress_f16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "amq_set_token_f16": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "amq_attn_prefill_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i] + [ctypes.c_longlong] * 10 + [_vp]),
    "amq_rope_cache_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "amq_rope_cache_batch_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "amq_rope_rows_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "amq_silu_mul_f16": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "amq_gemv_grouped_f16": (_i, [ctypes.POINTER(Segment), _i, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, ctypes.POINTER(GemvOpts), _vp]),
    "amq_gemv_grouped_sums_f16": (_i, [ctypes.POINTER(Segment), _i, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp]),
    "amq_gemm_f16w_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "amq_gemm_route_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "amq_gemm_route_workspace_bytes_g": (_sz, [_i, _i, _i, _i, _i]),
    "amq_gemm_res_norm_xfrag_f16": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _f, _vp, _vp]),
    "amq_gemm_route_f16": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
}

# name -> (restype, argtypes) of include/amq_hip_ab.h
AB_SIGNATURES = {
    "amq_gemv_qkv_attn_f16": (_i, [ctypes.POINTER(Segment), _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "amq_decode_engine_image_bytes": (_sz, [_i]),
    "amq_decode_engine_scratch_bytes": (_sz, [_i, _i, _i]),
    "amq_decode_engine_sync_bytes": (_sz, []),
    "amq_decode_engine_image": (_i, [ctypes.POINTER(EngineBlock), _i, _i, _i, _i, _i, _i, _i, _vp]),
    "amq_decode_engine_f16": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _sz, _vp, _vp, _sz, _i, _vp]),
}

_lib = None
_ab_lib = None


class AmqError(RuntimeError):
    pass


def load():
    """Load libamq_hip.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmqError(
            f"{LIB_PATH} not found: the HIP extension is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C amq_amd/csrc`). "
            "There is no CPU fallback.")
    # torch bundles its own libamdhip64.so.7; load it FIRST so that this
    # library binds to the same HIP runtime instance (two runtimes in one
    # process do not share a device context -> "no ROCm-capable device").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.amq_version() != ABI_VERSION:
        raise AmqError(f"{LIB_PATH} reports ABI version {lib.amq_version()}, these bindings were written against {ABI_VERSION} "
                       "(include/amq_hip.h AMQ_VERSION): rebuild with `make -C amq_amd/csrc`")
    _lib = lib
    return lib


def load_ab():
    """Load libamq_hip_ab.so (the A/B routes of include/amq_hip_ab.h) once; raise if it has not been built."""
    global _ab_lib
    if _ab_lib is not None:
        return _ab_lib
    if not os.path.exists(AB_LIB_PATH):
        raise AmqError(f"{AB_LIB_PATH} not found: the A/B routes (decode engine, fused q/k/v + attention) live in their own library -- "
                       "`make -C amq_amd/csrc ab` (or __graft_entry__.build()).  They are slower than the product step and not needed for it.")
    import torch  # noqa: F401
    lib = ctypes.CDLL(AB_LIB_PATH)
    for name, (res, args) in AB_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.amq_last_error.restype = ctypes.c_char_p
    _ab_lib = lib
    return lib


def check_ab(rc):
    if rc != AMQ_OK:
        raise AmqError(f"libamq_hip_ab error {rc}: {load_ab().amq_last_error().decode('utf-8', 'replace')}")


def use_library(path):
    """tools/ only: load an A/B build variant instead of the product library.  Must be called before the first load()."""
    global LIB_PATH
    global AB_LIB_PATH
    if _lib is not None or _ab_lib is not None:
        raise AmqError("use_library() must be called before the library is first loaded")
    LIB_PATH = os.path.abspath(path)
    AB_LIB_PATH = LIB_PATH          # (a variant that carries the A/B routes -- `make abvariant` -- serves them too; others fail loudly on load_ab)


SAFE_LIB_PATH = os.path.join(_HERE, "libamq_hip_safe.so")      # the twin built with -DAMQ_WAITS_CONSERVATIVE (`make -C amq_amd/csrc safe`)


def open_twin(path=None):
    """tests/ and tools/ only: a SECOND build of the library (default: the conservative-waits twin) opened beside the product one, with the same
    signatures.  Nothing in the product path calls this; route the ops wrappers through it for a block with :func:`routed_to`."""
    path = SAFE_LIB_PATH if path is None else os.path.abspath(path)
    if not os.path.exists(path):
        raise AmqError(f"{path} not found: `make -C amq_amd/csrc safe` (or __graft_entry__.build())")
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.amq_version() != ABI_VERSION:
        raise AmqError(f"{path} reports ABI version {lib.amq_version()}, expected {ABI_VERSION}")
    return lib


class routed_to:
    """tests/ and tools/ only: ``with routed_to(twin): ops.gemm(...)`` -- every ``load()`` inside the block hands out ``twin`` (the ctypes
    wrappers of ops.py fetch the library per call; the C++ fast path of the modules, _amq_ext, is not affected)."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        global _lib
        load()                      # (the product library is loaded first, as in any product run)
        self.prev, _lib = _lib, self.lib
        return self.lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def check(rc):
    if rc != AMQ_OK:
        msg = load().amq_last_error().decode("utf-8", "replace")
        raise AmqError(f"libamq_hip error {rc}: {msg}")


def current_stream():
    """raw handle of torch's current HIP stream on the current device"""
    import torch
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
    except AttributeError:                      # (older / newer torch without the private fast path)
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def stream_of(device):
    """raw handle of torch's current HIP stream on ``device`` (a torch.device with an index, or an int)"""
    import torch
    idx = device if isinstance(device, int) else (device.index if device.index is not None else torch.cuda.current_device())
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx))
    except AttributeError:
        return ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream)


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())
