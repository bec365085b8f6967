"""CPU-only: the C-ABI library loads and exports every symbol include/amq_hip.h
declares; argument validation works without a GPU (no compute calls)."""
import ctypes
import os
import re

import pytest

from amq_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="amq_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(amq_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in amq_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes SIGNATURES out of sync with amq_hip.h"
    # the A/B routes (include/amq_hip_ab.h) live in their own library: the product library does NOT export them
    ab = _declared_symbols("amq_hip_ab.h")
    assert sorted(_lib.AB_SIGNATURES) == ab and len(ab) == 6
    for s in ab:
        assert not hasattr(lib, s), f"{s} is an A/B route and must not be exported by the product library"
    if os.path.exists(_lib.AB_LIB_PATH):              # optional since round 6 (`make -C amq_amd/csrc ab`): not part of the default build
        ablib = ctypes.CDLL(_lib.AB_LIB_PATH)
        for s in ab + syms:
            assert hasattr(ablib, s), f"{s} not exported by libamq_hip_ab.so"
    # the conservative-waits twin (`make safe`, built by __graft_entry__.build()): same exports as the product library
    assert os.path.exists(_lib.SAFE_LIB_PATH), "run __graft_entry__.build() first (make safe)"
    safe = ctypes.CDLL(_lib.SAFE_LIB_PATH)
    for s in syms:
        assert hasattr(safe, s), f"{s} not exported by libamq_hip_safe.so"


def test_version_sizes_and_validation():
    lib = _lib.load()
    assert lib.amq_version() == 521 == _lib.ABI_VERSION
    # native sizes: N*K*bits/8 payload, 4 B of (scale, zero) per (row, group)
    for bits in (2, 3, 4):
        assert lib.amq_native_qweight_bytes(bits, 4096, 4096) == 4096 * 4096 * bits // 8
    assert lib.amq_native_meta_bytes(4096, 4096, 128) == 4096 * 4096 // 128 * 4
    # validation happens before any HIP call
    one = ctypes.c_void_p(16)
    rc = lib.amq_gemv_f16(5, 0, one, one, one, None, one, 1, 4096, 4096, 128, 0, 0, None)
    assert rc == -1 and b"bits" in lib.amq_last_error()
    rc = lib.amq_gemv_f16(4, 0, one, one, one, None, one, 1, 4096, 4100, 128, 0, 0, None)
    assert rc == -2
    rc = lib.amq_gemv_f16(4, 0, one, one, one, None, one, 1, 4096, 4096, 96, 0, 0, None)
    assert rc == -2 and b"group" in lib.amq_last_error()
    rc = lib.amq_gemv_f16(4, 0, one, one, one, None, one, 17, 4096, 4096, 128, 0, 0, None)
    assert rc == -2 and b"amq_gemm_f16" in lib.amq_last_error()
    # per-call options are validated too (host struct; the library holds no option state)
    seg = (_lib.Segment * 1)(_lib.Segment(16, 16, None, None, 16, 4096, 4, 0, 0))
    for bad in (_lib.GemvOpts(math=4), _lib.GemvOpts(math=-1), _lib.GemvOpts(waves=5), _lib.GemvOpts(depth=3), _lib.GemvOpts(rpt=65)):
        rc = lib.amq_gemv_grouped_f16(seg, 1, one, None, None, 0.0, 0, 1, 4096, 128, 0, ctypes.byref(bad), None)
        assert rc == -1 and b"opts." in lib.amq_last_error()
    assert not hasattr(lib, "amq_set_option")
    assert lib.amq_gemm_route_f16(9, 4, 0, one, one, one, None, None, one, 40, 4096, 4096, 128, 0, 0, None, 0, None) == -1
    assert lib.amq_gemm_route_f16(2, 4, 0, one, one, one, None, None, one, 65, 4096, 4096, 128, 0, 0, None, 0, None) == -2
    assert lib.amq_gemm_route_workspace_bytes(1, 64, 4096, 4096) > 0 and lib.amq_gemm_route_workspace_bytes(2, 64, 4096, 4096) == 0
    rc = lib.amq_gemm_f16(4, 7, one, one, one, None, one, 40, 4096, 4096, 128, 0, 0, None)
    assert rc == -1
    out = (ctypes.c_int * 4)()
    assert lib.amq_query(4096, out, 4) == 4 and out[0] >= 8 and out[2] == 16 and out[3] == 128
    # the ring kernel addresses its LDS-DMA sources as scalar base + 32-bit lane offset: a launch whose x (or packed W) spans
    # >= 4 GiB is not its launch (it goes to the tiled kernel, where a gate is an element-wise launch behind it) -- host logic
    assert lib.amq_gemm_gated_fused(_lib.GEMM_RING if hasattr(_lib, "GEMM_RING") else 3, 32768, 13824, 5120, 0) == 1
    assert lib.amq_gemm_gated_fused(3, 300000, 4096, 8192, 0) == 0          # x: 300000 x 8192 x 2 B = 4.9 GB
    assert lib.amq_gemm_gated_fused(3, 200000, 4096, 8192, 0) == 1          # 3.3 GB


def test_dequantize_once_route_host_side():
    """route 6 (AMQ_GEMM_DEQ: dequantize once into the caller's workspace + the fp16 ping-pong GEMM) and the dense entry: workspace
    sizes, AUTO's crossover and argument validation -- host logic, no GPU"""
    lib = _lib.load()
    N, K = 13824, 5120
    assert lib.amq_gemm_route_workspace_bytes(_lib.GEMM_DEQ, 64, N, K) == N * K * 2            # forced: always the fp16 weights
    assert lib.amq_gemm_route_workspace_bytes(_lib.GEMM_AUTO, 32768, N, K) == N * K * 2        # BASELINE configs[3]: MFMA-bound
    assert lib.amq_gemm_route_workspace_bytes(_lib.GEMM_AUTO, 6144, N, K) == N * K * 2
    assert lib.amq_gemm_route_workspace_bytes(_lib.GEMM_AUTO, 4096, N, K) == 0                 # fused kernels below the crossover
    assert lib.amq_gemm_route_workspace_bytes(_lib.GEMM_AUTO, 8192, 1024, 1024) == 0           # < one round of 256 x 256 tiles
    assert lib.amq_gemm_gated_fused(_lib.GEMM_DEQ, 8192, N, K, 1) == 1 and lib.amq_gemm_gated_fused(_lib.GEMM_DEQ, 8192, N, K, 0) == 0
    one = ctypes.c_void_p(256)
    # route 6 needs its workspace, of the right size
    assert lib.amq_gemm_route_f16(_lib.GEMM_DEQ, 4, 0, one, one, one, None, None, one, 512, N, K, 128, 0, 0, None, 0, None) == -1
    assert b"workspace" in lib.amq_last_error()
    assert lib.amq_gemm_route_f16(_lib.GEMM_DEQ, 4, 0, one, one, one, None, None, one, 512, N, K, 128, 0, 0, one, 1024, None) == -1
    assert lib.amq_gemm_route_f16(7, 4, 0, one, one, one, None, None, one, 512, N, K, 128, 0, 0, None, 0, None) == -1   # unknown route
    # dense entry
    f = lib.amq_gemm_f16w_f16
    assert f(None, one, None, None, None, one, 512, N, K, 0, 0, None) == -1
    assert f(one, one, None, one, one, one, 512, N, K, 0, 0, None) == -1 and b"exclusive" in lib.amq_last_error()
    assert f(one, one, None, None, None, one, 512, N, K - 64, 0, 0, None) == -2                 # K % 128
    assert f(one, one, None, None, None, one, 512, N + 8, K, 0, 0, None) == -2                  # N % 16
    assert f(one, one, None, None, None, one, 512, N, K, K + 4, 0, None) == -2                  # x rows not 16-byte aligned
    assert f(one, one, None, None, None, one, 1 << 20, N, 8192, 0, 0, None) == -2               # x spans >= 4 GiB


def test_finer_groups_host_side():
    """groups of 64 / 32: native meta sizes, the workspace query, and which entry points take them (host logic, no GPU)"""
    lib = _lib.load()
    N, K = 4096, 4096
    base = lib.amq_native_meta_bytes(N, K, 128)
    assert lib.amq_native_meta_bytes(N, K, 256) == base and lib.amq_native_meta_bytes(N, K, 64) == 2 * base and lib.amq_native_meta_bytes(N, K, 32) == 4 * base
    for g in (64, 32):
        for m in (1, 17, 256):
            assert lib.amq_gemm_route_workspace_bytes_g(_lib.GEMM_AUTO, m, N, K, g) == 0               # the (pair-aware) few-row kernel: no workspace
            assert lib.amq_gemm_route_workspace_bytes_g(_lib.GEMM_DEQ, m, N, K, g) == N * K * 2
        assert lib.amq_gemm_route_workspace_bytes_g(_lib.GEMM_AUTO, 300, N, K, g) != N * K * 2         # the tiled kernel (split-K partials, or nothing)
        for m in (2048, 4096, 32768):
            assert lib.amq_gemm_route_workspace_bytes_g(_lib.GEMM_AUTO, m, N, K, g) == N * K * 2       # launches that fill 256 x 256 tiles: dequantize-once
        assert lib.amq_gemm_route_workspace_bytes_g(_lib.GEMM_RING, 4096, N, K, g) == 0
    assert lib.amq_gemm_route_workspace_bytes_g(_lib.GEMM_AUTO, 64, N, K, 128) == lib.amq_gemm_route_workspace_bytes(_lib.GEMM_AUTO, 64, N, K)
    # the gate is applied by the few-row kernel and by the dequantize-once route, not by the tiled kernel between them
    assert lib.amq_gemm_gated_fused_g(_lib.GEMM_AUTO, 64, N, K, 0, 64) == 1 and lib.amq_gemm_gated_fused_g(_lib.GEMM_AUTO, 512, N, K, 1, 64) == 0
    assert lib.amq_gemm_gated_fused_g(_lib.GEMM_AUTO, 4096, N, K, 1, 64) == 1 and lib.amq_gemm_gated_fused_g(_lib.GEMM_AUTO, 4096, N, K, 0, 64) == 0
    assert lib.amq_gemm_gated_fused_g(_lib.GEMM_AUTO, 64, N, K, 0, 128) == lib.amq_gemm_gated_fused(_lib.GEMM_AUTO, 64, N, K, 0)
    one = ctypes.c_void_p(256)
    # the plain GEMM entry has no workspace: refused with the way out in the message
    # route calls: the ring / wave-specialised kernels read one pair per tile; the dequantize-once route needs its workspace
    assert lib.amq_gemm_route_f16(_lib.GEMM_RING, 4, 0, one, one, one, None, None, one, 4096, N, K, 32, 0, 0, one, N * K * 2, None) == -1
    assert b"dequantize-once" in lib.amq_last_error()
    assert lib.amq_gemm_route_f16(_lib.GEMM_DEQ, 4, 0, one, one, one, None, None, one, 300, N, K, 64, 0, 0, None, 0, None) == -1
    assert b"workspace" in lib.amq_last_error()
    assert lib.amq_gemm_route_f16(_lib.GEMM_AUTO, 4, 0, one, one, one, None, None, one, 4096, N, K, 64, 0, 0, one, 1024, None) == -1  # too small
    assert lib.amq_gemm_route_f16(_lib.GEMM_SKINNY, 4, 0, one, one, one, None, None, one, 300, N, K, 64, 0, 0, None, 0, None) == -2  # past its rows
    # GEMV options other than the default form
    seg = (_lib.Segment * 1)(_lib.Segment(256, 256, None, None, 256, N, 4, 0, 0))
    for bad in (_lib.GemvOpts(math=1), _lib.GemvOpts(dot=1), _lib.GemvOpts(depth=4)):
        assert lib.amq_gemv_grouped_f16(seg, 1, one, None, None, 0.0, 0, 1, K, 64, 0, ctypes.byref(bad), None) == -1
        assert b"default form" in lib.amq_last_error()
    assert lib.amq_gemm_xfrag_f16(4, 0, one, one, one, None, None, None, one, 17, N, K, 64, 0, None) == -2


@pytest.mark.skipif(not os.path.exists(_lib.AB_LIB_PATH), reason="libamq_hip_ab.so not built (make -C amq_amd/csrc ab): A/B routes only")
def test_decode_engine_host_side():
    """the one-launch-per-token engine (an A/B route, libamq_hip_ab.so): sizes, the host-side table builder and argument validation
    (no GPU needed)"""
    lib = _lib.load_ab()
    assert lib.amq_decode_engine_sync_bytes() % 256 == 0 and lib.amq_decode_engine_sync_bytes() >= 66 * 256
    assert lib.amq_decode_engine_image_bytes(32) == 32 * lib.amq_decode_engine_image_bytes(1) > 0
    H, I, nh, nkv = 4096, 11008, 32, 32
    r256 = lambda halves: (2 * halves + 255) // 256 * 256
    assert lib.amq_decode_engine_scratch_bytes(H, I, nkv) == 2 * r256(H) + 2 * r256(nkv * 128) + 2 * r256(I)
    blocks = (_lib.EngineBlock * 2)()
    Ns = [H, nkv * 128, nkv * 128, H, I, I, H]
    for b in range(2):
        for i in range(7):
            blocks[b].lin[i] = _lib.EngineLinear(4096 * (1 + i), 8192 * (1 + i), Ns[i], 2 + (i + b) % 3, 0, 0)
        blocks[b].ln1, blocks[b].ln2, blocks[b].kcache, blocks[b].vcache = 16, 32, 48, 64
    image = (ctypes.c_ubyte * lib.amq_decode_engine_image_bytes(2))()
    assert lib.amq_decode_engine_image(blocks, 2, H, I, nh, nkv, 128, 128, image) == 0
    assert any(image)                                                       # the table was written
    blocks[1].lin[4].N = I - 16
    assert lib.amq_decode_engine_image(blocks, 2, H, I, nh, nkv, 128, 128, image) == -2 and b"linear 4" in lib.amq_last_error()
    blocks[1].lin[4].N = I
    blocks[0].lin[2].bits = 5
    assert lib.amq_decode_engine_image(blocks, 2, H, I, nh, nkv, 128, 128, image) == -1
    blocks[0].lin[2].bits = 3
    assert lib.amq_decode_engine_image(blocks, 2, H, I, nh, nkv, 64, 128, image) == -2          # head_dim
    assert lib.amq_decode_engine_image(blocks, 2, H + 128, I, nh, nkv, 128, 128, image) == -2   # hidden != heads * 128
    one = ctypes.c_void_p(16)
    args = dict(sc=lib.amq_decode_engine_scratch_bytes(H, I, nkv), sy=lib.amq_decode_engine_sync_bytes())
    assert lib.amq_decode_engine_f16(None, 2, H, I, nh, nkv, 128, 256, 1e-5, one, one, args["sc"], one, one, args["sy"], 0, None) == -1
    assert lib.amq_decode_engine_f16(one, 2, H, I, nh, nkv, 128, 256, 1e-5, one, one, args["sc"] - 1, one, one, args["sy"], 0, None) == -1
    assert b"scratch" in lib.amq_last_error()
    assert lib.amq_decode_engine_f16(one, 2, H, I, nh, nkv, 128, 256, 1e-5, one, one, args["sc"], one, one, 64, 0, None) == -1
    assert lib.amq_decode_engine_f16(one, 2, H, I, nh, 3, 128, 256, 1e-5, one, one, args["sc"], one, one, args["sy"], 0, None) == -2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.AmqError, match="no CPU fallback"):
        _lib.load()
