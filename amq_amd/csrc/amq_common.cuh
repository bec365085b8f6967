// amq_common.cuh -- native weight layout ("AMQ-T16") and the in-register
// unpack / dequantize primitives shared by every kernel.  gfx950 only.
//
// Native layout (chosen for a 64-lane wavefront and the 16x16x32 MFMA operand
// map; none of the reference formats is used on the device hot path):
//
//   tile  = 16 output rows (n) x 128 input columns (k) = one quantization
//           group per row.  Tiles are stored row-tile major:
//               tile_index = (n/16) * (K/128) + (k/128)
//           so everything a workgroup needs for 16 output rows is ONE
//           contiguous byte range, streamed front to back.
//   lane  l of the wave that owns a tile holds row  r = l & 15  and the four
//           k-octets  k = 32*t + 8*(l>>4) + j   (t = 0..3, j = 0..7)
//           -- exactly the B-operand lane map of v_mfma_f32_16x16x32_f16
//           (lane l: B[k = 8*(l>>4) + j][col = l & 15]) for MFMA step t, and a
//           16-byte-contiguous slice of x for the dot-product path.
//   bytes per lane and tile: 4*BITS (16 / 12 / 8 for 4 / 3 / 2 bit); the 64
//           lanes' payloads are contiguous, so one wave-load instruction reads
//           1024 / 768 / 512 contiguous bytes.
//   the 32 weights of a lane form 16 "pairs" P = 4*t + p (p = 0..3) =
//           (w[t][2p], w[t][2p+1]); a pair lives at the SAME bit offset of the
//           low and the high 16-bit half of a dword, so one v_and_or_b32 turns
//           it into a packed fp16x2 (magic-number trick, 0x6400 = 1024.0):
//     4-bit: dword t, slot p at bits 4p          (4 pairs / dword)
//     2-bit: dword P/8, slot P%8 at bits 2*(P%8) (8 pairs / dword)
//     3-bit: dword P/5, slot P%5 at bits 3*(P%5) for P < 15 (5 pairs/dword);
//            pair 15 is spread over bit 15 (low weight) and bit 31 (high
//            weight) of dwords 0,1,2 (value bit 0,1,2).
//   meta:   __half2 per (row, group): [N/16][K/128][16]
//            MODE_HQQ: (scale, zero)  w = fp16(fp16(q - zero) * scale)
//                      == Quantizer.dequantize, hqq/core/quantize.py:198
//            MODE_FMA: (scale, c)     w = fp16(fma(q, scale, c))
//                      == the reference CUDA kernels' dequant
//                      (auto_gptq_kernel.cu:206, gemv_cuda.cu:151)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amq {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u3 __attribute__((ext_vector_type(3)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

enum { MODE_HQQ = 0, MODE_FMA = 1 };
enum { TILE_N = 16, TILE_K = 128, GROUP = 128 };

__device__ __forceinline__ h2 as_h2(uint32_t u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ uint32_t as_u32(h2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ h2 bcast(_Float16 v) { return (h2){v, v}; }

// (u & MASK) | 0x64006400 -> packed halves 1024 + q*2^SHIFT ; bring back to q.
// All steps are exact in fp16 (q*2^SHIFT <= 960 fits the 10-bit mantissa).
template <int SHIFT>
__device__ __forceinline__ h2 field_to_h2(uint32_t u, uint32_t fieldmask) {
    const uint32_t m = (fieldmask << SHIFT) | ((fieldmask << SHIFT) << 16);
    h2 v = as_h2((u & m) | 0x64006400u);
    if (SHIFT == 0) {
        return v - bcast((_Float16)1024.0f);
    } else {
        const _Float16 inv = (_Float16)(1.0f / (float)(1 << SHIFT));
        const _Float16 off = (_Float16)(-1024.0f / (float)(1 << SHIFT));
        return __builtin_elementwise_fma(v, bcast(inv), bcast(off));
    }
}

template <int MODE>
__device__ __forceinline__ h2 apply_meta(h2 q, h2 s2, h2 z2) {
    if (MODE == MODE_HQQ) {
        h2 d = q - z2;          // fp16 rounding #1   (W_r - zero)
        return d * s2;          // fp16 rounding #2   (* scale)
    } else {
        return __builtin_elementwise_fma(q, s2, z2);   // one fused rounding
    }
}

// Unpack + dequantize one lane's 32 weights of a tile.
// out[4*t + p] = (w[t][2p], w[t][2p+1]) as packed fp16.
template <int BITS, int MODE>
__device__ __forceinline__ void dequant_lane(const uint32_t* w, h2 meta, h2* out) {
    const h2 s2 = bcast(meta.x);
    const h2 z2 = bcast(meta.y);
    if (BITS == 4) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t u = w[t];
            const uint32_t v = u >> 8;
            out[4 * t + 0] = apply_meta<MODE>(field_to_h2<0>(u, 0xFu), s2, z2);
            out[4 * t + 1] = apply_meta<MODE>(field_to_h2<4>(u, 0xFu), s2, z2);
            out[4 * t + 2] = apply_meta<MODE>(field_to_h2<0>(v, 0xFu), s2, z2);
            out[4 * t + 3] = apply_meta<MODE>(field_to_h2<4>(v, 0xFu), s2, z2);
        }
    } else if (BITS == 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const uint32_t u = w[d];
            const uint32_t v = u >> 10;
            out[8 * d + 0] = apply_meta<MODE>(field_to_h2<0>(u, 0x3u), s2, z2);
            out[8 * d + 1] = apply_meta<MODE>(field_to_h2<2>(u, 0x3u), s2, z2);
            out[8 * d + 2] = apply_meta<MODE>(field_to_h2<4>(u, 0x3u), s2, z2);
            out[8 * d + 3] = apply_meta<MODE>(field_to_h2<6>(u, 0x3u), s2, z2);
            out[8 * d + 4] = apply_meta<MODE>(field_to_h2<8>(u, 0x3u), s2, z2);
            out[8 * d + 5] = apply_meta<MODE>(field_to_h2<0>(v, 0x3u), s2, z2);
            out[8 * d + 6] = apply_meta<MODE>(field_to_h2<2>(v, 0x3u), s2, z2);
            out[8 * d + 7] = apply_meta<MODE>(field_to_h2<4>(v, 0x3u), s2, z2);
        }
    } else {  // BITS == 3
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t u = w[d];
            const uint32_t v = u >> 9;
            out[5 * d + 0] = apply_meta<MODE>(field_to_h2<0>(u, 0x7u), s2, z2);
            out[5 * d + 1] = apply_meta<MODE>(field_to_h2<3>(u, 0x7u), s2, z2);
            out[5 * d + 2] = apply_meta<MODE>(field_to_h2<6>(u, 0x7u), s2, z2);
            out[5 * d + 3] = apply_meta<MODE>(field_to_h2<0>(v, 0x7u), s2, z2);
            out[5 * d + 4] = apply_meta<MODE>(field_to_h2<3>(v, 0x7u), s2, z2);
        }
        // pair 15: value bit b of (low, high) weight = bit (15, 31) of dword b
        const uint32_t e = ((w[0] >> 15) & 0x00010001u) | ((w[1] >> 14) & 0x00020002u) |
                           ((w[2] >> 13) & 0x00040004u);
        out[15] = apply_meta<MODE>(field_to_h2<0>(e, 0x7u), s2, z2);
    }
}

// Integer-only view of the same map (used by the repack / reference-format
// kernels): where does weight (t, j) of a lane live?
__host__ __device__ __forceinline__ void native_slot(int bits, int t, int j, int* dword, int* shift) {
    const int P = 4 * t + (j >> 1);
    const int hi = (j & 1) * 16;
    if (bits == 4) { *dword = t; *shift = 4 * (j >> 1) + hi; }
    else if (bits == 2) { *dword = P >> 3; *shift = 2 * (P & 7) + hi; }
    else { *dword = P / 5; *shift = 3 * (P % 5) + hi; }   // 3-bit, P < 15 (P == 15 handled by caller)
}

__host__ __device__ __forceinline__ size_t native_qweight_bytes(int bits, int N, int K) {
    return (size_t)(N / TILE_N) * (size_t)(K / TILE_K) * 64u * 4u * (size_t)bits;
}
__host__ __device__ __forceinline__ size_t native_meta_bytes(int N, int K) {
    return (size_t)(N / TILE_N) * (size_t)(K / TILE_K) * TILE_N * 4u;
}

// ---- wave-load of one lane's tile payload (BITS dwords), non-temporal: the
// weights are read exactly once per token, keep them from displacing x / KV.
template <int BITS>
struct LanePayload { uint32_t w[BITS]; };

template <int BITS>
__device__ __forceinline__ LanePayload<BITS> load_payload(const uint32_t* tile_base, int lane) {
    LanePayload<BITS> r;
    if (BITS == 4) {
        u4 v = __builtin_nontemporal_load((const u4*)tile_base + lane);
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    } else if (BITS == 2) {
        u2 v = __builtin_nontemporal_load((const u2*)tile_base + lane);
        r.w[0] = v.x; r.w[1] = v.y;
    } else {
        const uint32_t* p = tile_base + 3 * lane;
        r.w[0] = __builtin_nontemporal_load(p);
        r.w[1] = __builtin_nontemporal_load(p + 1);
        r.w[2] = __builtin_nontemporal_load(p + 2);
    }
    return r;
}

}  // namespace amq
