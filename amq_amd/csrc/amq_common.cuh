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

// MODE_FMA1: MODE_FMA for buffers whose scales are small enough for the GEMV's one-op unpack (fma1_scale_bound) -- same results; every other kernel
// treats it as MODE_FMA (all mode tests are "== MODE_HQQ ? ... : fma form")
enum { MODE_HQQ = 0, MODE_FMA = 1, MODE_FMA1 = 2 };
enum { TILE_N = 16, TILE_K = 128, GROUP = 128 };

// ---- hand-counted waits --------------------------------------------------------------------------------------------------------------
// LDS-DMA transfers (global_load_lds / buffer_load ... lds, issued from inline asm) are invisible to the compiler's wait insertion, and the
// software pipelines built on them leave a COUNTED number of younger transfers in flight.  Every s_waitcnt this build writes itself goes
// through these helpers (tests/test_waits_cpu.py refuses a bare one anywhere in csrc/):
//   * the site's name and what its count is derived from travel as an assembler comment into the device assembly the Makefile keeps under
//     csrc/asm/; tools/check_waits.py walks the control-flow graph of every kernel there and checks, per site, that the number of
//     vector-memory instructions the compiler actually emitted between the named points is the number the source's count assumes
//     (`from=<site>:<ops>` pairs; `entry` = kernel start).  A load the compiler merged, split or sank shows as a CPU-side failure.
//   * -DAMQ_WAITS_CONSERVATIVE (make safe -> libamq_hip_safe.so) turns every one of them into a full drain: the twin library the GPU suite
//     compares the product with bit for bit (tests/test_gpu_waits.py).
// AMQ_WAIT_VM(name, n, "from=<site>:%1 ...", "n"(ops) ...)   s_waitcnt vmcnt(n); n and the spec's counts are compile-time integers (asm operands)
// AMQ_WAIT_VM_LGKM0(...)                                      ... and lgkmcnt(0)
// AMQ_MARK(name)                                              a comment on an EXISTING asm statement (no instruction of its own) for `from=`
#ifdef AMQ_WAITS_CONSERVATIVE
#define AMQ_WAIT_VM(name, n, spec, ...) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0) ; AMQ_WAIT id=" name " conservative" ::: "memory")
#define AMQ_WAIT_VM_LGKM0(name, n, spec, ...) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0) ; AMQ_WAIT id=" name " conservative" ::: "memory")
#else
#define AMQ_WAIT_VM(name, n, spec, ...) asm volatile("s_waitcnt vmcnt(%0) ; AMQ_WAIT id=" name " n=%0 " spec :: "n"(n), ##__VA_ARGS__ : "memory")
#define AMQ_WAIT_VM_LGKM0(name, n, spec, ...) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0) ; AMQ_WAIT id=" name " n=%0 " spec :: "n"(n), ##__VA_ARGS__ : "memory")
#endif
#define AMQ_WAIT_LGKM0(name) asm volatile("s_waitcnt lgkmcnt(0) ; AMQ_WAIT id=" name " lgkm" ::: "memory")
#define AMQ_MARK(name) " ; AMQ_MARK id=" name

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

// ---------------------------------------------------------------------------
// Fast-path unpack ("scaled-subnormal" form) used inside the matmul kernels.
//
// Measured on gfx950 (tools/ubench/valu_rate.hip): every VOP3/VOP3P op
// (v_pk_*_f16, v_and_or_b32, v_perm_b32, v_dot2*) issues at 4 cycles per wave,
// plain VOP2 (v_and_b32, v_lshrrev_b32) at 2.  The magic-number unpack above
// costs and+or+sub+sub+mul = 18 cycles / pair; at 2-4 bits per weight that is
// MORE VALU time than the HBM time of the weights.  This form needs
// shift+and+fma+mul = 12:
//
//   t = u << / >> c       bring the pair's field to the TOP of the fp16
//                         mantissa (bit SH of each half)
//   (t & mask)            exponent bits are zero, so the half IS the
//                         subnormal  q * 2^(SH-24)
//   d = fma(sub, 2^B, -z * 2^E)   =  RN16((q - z) * 2^E),   E = SH + B - 24
//   w = d * (s * 2^-E)            =  RN16(RN16(q - z) * s)
//
// Scaling by a power of two commutes with fp16 rounding, so w is bit-identical
// to the reference's two-rounding dequant (quantize.py:198) as long as
// z * 2^E and (q - z) * 2^E are normal fp16 numbers: E = -3 / -5 / -5 for
// 4 / 3 / 2 bit, i.e. for |z|, |q - z| >= 2^-11 / 2^-9 / 2^-9; below that the
// value involved is < 1e-3 of a quantization step and may differ by
// <= 2^-24 * 2^-E (tests bound it).  Requires |s| * 2^-E < 65504.
// MODE_FMA: w = fma(sub * 2^B, s * 2^-E, c)   (q * 2^E is exact).
template <int BITS> struct SdCfg;
template <> struct SdCfg<4> { static constexpr int E = -3; };
template <> struct SdCfg<3> { static constexpr int E = -5; };   // (two field positions per shifted copy: 3 shifts per 5 pairs)
template <> struct SdCfg<2> { static constexpr int E = -5; };

struct SdMeta { h2 zc, sc; };

template <int BITS, int MODE>
__device__ __forceinline__ SdMeta sd_meta(h2 meta) {
    constexpr int E = SdCfg<BITS>::E;
    SdMeta m;
    m.sc = bcast(meta.x) * bcast((_Float16)(float)(1 << (-E)));
    if (MODE == MODE_HQQ) m.zc = bcast(meta.y) * bcast((_Float16)(-1.0f / (float)(1 << (-E))));
    else m.zc = bcast(meta.y);
    return m;
}

// t: word already shifted so that the pair's field starts at bit SH of each half
template <int BITS, int MODE, int SH>
__device__ __forceinline__ h2 sd_pair(uint32_t t, const SdMeta& m) {
    constexpr uint32_t fm = (1u << BITS) - 1u;
    constexpr uint32_t msk = (fm << SH) | ((fm << SH) << 16);
    constexpr int B = SdCfg<BITS>::E + 24 - SH;                  // multiplier exponent, <= 15
    static_assert(B <= 15 && B >= 0 && SH + BITS <= 10, "field must sit in the mantissa");
    const h2 sub = as_h2(t & msk);                               // q * 2^(SH-24), exact subnormal
    const _Float16 mul = (_Float16)(float)(1 << B);
    if (MODE == MODE_HQQ) {
        h2 d = __builtin_elementwise_fma(sub, bcast(mul), m.zc); // RN16((q - z) * 2^E)
#ifdef AMQ_ABL_NOMUL               /* timing-only ablation: the second rounding's multiply dropped (one VOP3P per pair instead of two; wrong weights) */
        return d;
#else
        return d * m.sc;
#endif
    } else {
        h2 qs = sub * bcast(mul);                                // q * 2^E, exact
        return __builtin_elementwise_fma(qs, m.sc, m.zc);
    }
}

template <int BITS, int MODE>
__device__ __forceinline__ void dequant_lane_sd(const uint32_t* w, h2 meta, h2* out) {
    const SdMeta m = sd_meta<BITS, MODE>(meta);
    if (BITS == 4) {            // fields at bits 0,4,8,12 of each half -> bits 6..9
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t u = w[t];
            out[4 * t + 0] = sd_pair<4, MODE, 6>(u << 6, m);
            out[4 * t + 1] = sd_pair<4, MODE, 6>(u << 2, m);
            out[4 * t + 2] = sd_pair<4, MODE, 6>(u >> 2, m);
            out[4 * t + 3] = sd_pair<4, MODE, 6>(u >> 6, m);
        }
    } else if (BITS == 2) {     // fields at bits 2i: three shifted copies put all of them at 4, 6 or 8
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const uint32_t u = w[d];
            const uint32_t a = u << 4, b = u >> 2, c = u >> 8;
            out[8 * d + 0] = sd_pair<2, MODE, 4>(a, m);
            out[8 * d + 1] = sd_pair<2, MODE, 6>(a, m);
            out[8 * d + 2] = sd_pair<2, MODE, 8>(a, m);
            out[8 * d + 3] = sd_pair<2, MODE, 4>(b, m);
            out[8 * d + 4] = sd_pair<2, MODE, 6>(b, m);
            out[8 * d + 5] = sd_pair<2, MODE, 8>(b, m);
            out[8 * d + 6] = sd_pair<2, MODE, 4>(c, m);
            out[8 * d + 7] = sd_pair<2, MODE, 6>(c, m);
        }
    } else {                    // fields at bits 3i: three shifted copies put them at bit 4 or 7 (two positions per copy)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t u = w[d];
            const uint32_t a = u << 4, b = u >> 2, c = u >> 8;
            out[5 * d + 0] = sd_pair<3, MODE, 4>(a, m);
            out[5 * d + 1] = sd_pair<3, MODE, 7>(a, m);
            out[5 * d + 2] = sd_pair<3, MODE, 4>(b, m);
            out[5 * d + 3] = sd_pair<3, MODE, 7>(b, m);
            out[5 * d + 4] = sd_pair<3, MODE, 4>(c, m);
        }
        // pair 15: value bit b of the (low, high) weight = bit (15, 31) of dword b -> bits 7+b
        const uint32_t e = ((w[0] >> 8) & 0x00800080u) | ((w[1] >> 7) & 0x01000100u) |
                           ((w[2] >> 6) & 0x02000200u);
        out[15] = sd_pair<3, MODE, 7>(e, m);
    }
}

// MODE_FMA in ONE packed op per pair: w = fma(q 2^(SH-24), s 2^(24-SH), c) = RN16(q s + c), the same single rounding as the two-op form above
// (q 2^E exact, then fma with s 2^-E) -- bit-identical wherever s 2^(24-SH) is finite in fp16: with the fields at bit 4 / 6 / 7 / 8 of the halves
// that is |s| <= 65504 / 2^20 (2-, 3-bit: a field at bit 4) or 65504 / 2^18 (4-bit: bit 6).  The bound is checked per LAYER at load time
// (fma1_scale_bound; MODE_FMA1 in the buffers' mode): an in-kernel fallback would cost every body of the kernel its register budget.
__host__ __device__ __forceinline__ float fma1_scale_bound(int bits) { return bits == 4 ? 65504.0f / 262144.0f : 65504.0f / 1048576.0f; }
template <int BITS, int SH>
__device__ __forceinline__ h2 fma1_pair(uint32_t t, h2 s_sh, h2 c2) {
    constexpr uint32_t fm = (1u << BITS) - 1u;
    constexpr uint32_t msk = (fm << SH) | ((fm << SH) << 16);
    return __builtin_elementwise_fma(as_h2(t & msk), s_sh, c2);
}
template <int BITS>
__device__ __forceinline__ void dequant_lane_fma1(const uint32_t* w, h2 meta, h2* out) {
    const h2 s2 = bcast(meta.x), c2 = bcast(meta.y);
    if (BITS == 4) {            // fields brought to bit 6 of each half: x 2^18
        const h2 k6 = s2 * bcast((_Float16)512.0f) * bcast((_Float16)512.0f);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t u = w[t];
            out[4 * t + 0] = fma1_pair<4, 6>(u << 6, k6, c2);
            out[4 * t + 1] = fma1_pair<4, 6>(u << 2, k6, c2);
            out[4 * t + 2] = fma1_pair<4, 6>(u >> 2, k6, c2);
            out[4 * t + 3] = fma1_pair<4, 6>(u >> 6, k6, c2);
        }
    } else if (BITS == 2) {     // fields at bit 4 / 6 / 8: x 2^20 / 2^18 / 2^16
        const h2 k8 = s2 * bcast((_Float16)256.0f) * bcast((_Float16)256.0f);
        const h2 k6 = k8 * bcast((_Float16)4.0f), k4 = k8 * bcast((_Float16)16.0f);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const uint32_t u = w[d];
            const uint32_t a = u << 4, b = u >> 2, c = u >> 8;
            out[8 * d + 0] = fma1_pair<2, 4>(a, k4, c2);
            out[8 * d + 1] = fma1_pair<2, 6>(a, k6, c2);
            out[8 * d + 2] = fma1_pair<2, 8>(a, k8, c2);
            out[8 * d + 3] = fma1_pair<2, 4>(b, k4, c2);
            out[8 * d + 4] = fma1_pair<2, 6>(b, k6, c2);
            out[8 * d + 5] = fma1_pair<2, 8>(b, k8, c2);
            out[8 * d + 6] = fma1_pair<2, 4>(c, k4, c2);
            out[8 * d + 7] = fma1_pair<2, 6>(c, k6, c2);
        }
    } else {                    // fields at bit 4 / 7: x 2^20 / 2^17
        const h2 k7 = s2 * bcast((_Float16)512.0f) * bcast((_Float16)256.0f);
        const h2 k4 = k7 * bcast((_Float16)8.0f);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t u = w[d];
            const uint32_t a = u << 4, b = u >> 2, c = u >> 8;
            out[5 * d + 0] = fma1_pair<3, 4>(a, k4, c2);
            out[5 * d + 1] = fma1_pair<3, 7>(a, k7, c2);
            out[5 * d + 2] = fma1_pair<3, 4>(b, k4, c2);
            out[5 * d + 3] = fma1_pair<3, 7>(b, k7, c2);
            out[5 * d + 4] = fma1_pair<3, 4>(c, k4, c2);
        }
        const uint32_t e = ((w[0] >> 8) & 0x00800080u) | ((w[1] >> 7) & 0x01000100u) | ((w[2] >> 6) & 0x02000200u);
        out[15] = fma1_pair<3, 7>(e, k7, c2);
    }
}

// ---------------------------------------------------------------------------
// Group-scale unpack (MATH_GROUPSCALE of the GEMV kernel, MODE_HQQ): only the FIRST of the reference's two roundings is taken per weight,
//     d = RN16((q - z) * 2^E)          one packed fma on the subnormal field, as above,
// the tile's four MFMAs sum x * d in fp32 and the scale is applied ONCE per (row, group) to that sum:  y += (s * 2^-E) * sum_k x_k d_k.
// What is dropped is the second rounding, RN16(d * s): one fp16 rounding of each weight (<= 2^-11 relative, ~2e-4 of rms(y) on the output;
// the reference's own CUDA kernels round once as well, auto_gptq_kernel.cu:206-218).  Because the scale no longer has to be an fp16
// constant, E is free -- and with E = -9 a field may stay where the packing put it: every field that lies inside the fp16 mantissa
// (bit SH + BITS <= 10 of its half) is a subnormal q * 2^(SH-24) and takes the multiplier 2^(E+24-SH) <= 2^15 straight from a scalar
// register.  4 bit: slots 0, 1 of u and of u >> 8; 3 bit: slots 0..2 of u, 3..4 from u >> 9; 2 bit: slots 0..4 of u, 5..7 from u >> 10 --
// ONE shift per dword instead of one per pair, and + fma per pair: 6.5 VALU cycles per pair instead of 12.
// d is the exact first rounding wherever (q - z) * 2^-9 and z * 2^-9 are normal halves (|q - z|, |z| >= 2^-5); below that the small value
// (z, or q - z) is taken to a multiple of 2^-15 quantization steps first -- 2^-16 of a step off, which at worst moves the first rounding
// by one fp16 ulp of (q - z) (tests/test_gpu_kernels.py::test_matmul_weights_equal_oracle_weights bounds both).
constexpr int GS_E = -9;
template <int BITS, int SH>
__device__ __forceinline__ h2 gs_pair(uint32_t t, h2 zc) {
    constexpr uint32_t fm = (1u << BITS) - 1u;
    constexpr uint32_t msk = (fm << SH) | ((fm << SH) << 16);
    constexpr int B = GS_E + 24 - SH;
    static_assert(B <= 15 && B >= 0 && SH + BITS <= 10, "field must sit in the mantissa");
    return __builtin_elementwise_fma(as_h2(t & msk), bcast((_Float16)(float)(1 << B)), zc);
}
// zc for gs_pair: -(z * 2^E)
__device__ __forceinline__ h2 gs_zero(h2 meta) { return bcast(meta.y) * bcast((_Float16)(-1.0f / (float)(1 << (-GS_E)))); }
// the fp32 factor of a tile's sum: s * 2^-E
__device__ __forceinline__ float gs_scale(h2 meta) { return (float)meta.x * (float)(1 << (-GS_E)); }

template <int BITS>
__device__ __forceinline__ void dequant_lane_gs(const uint32_t* w, h2 zc, h2* out) {
    if (BITS == 4) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t u = w[t], v = u >> 8;
            out[4 * t + 0] = gs_pair<4, 0>(u, zc);
            out[4 * t + 1] = gs_pair<4, 4>(u, zc);
            out[4 * t + 2] = gs_pair<4, 0>(v, zc);
            out[4 * t + 3] = gs_pair<4, 4>(v, zc);
        }
    } else if (BITS == 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const uint32_t u = w[d], v = u >> 10;
            out[8 * d + 0] = gs_pair<2, 0>(u, zc);
            out[8 * d + 1] = gs_pair<2, 2>(u, zc);
            out[8 * d + 2] = gs_pair<2, 4>(u, zc);
            out[8 * d + 3] = gs_pair<2, 6>(u, zc);
            out[8 * d + 4] = gs_pair<2, 8>(u, zc);
            out[8 * d + 5] = gs_pair<2, 0>(v, zc);
            out[8 * d + 6] = gs_pair<2, 2>(v, zc);
            out[8 * d + 7] = gs_pair<2, 4>(v, zc);
        }
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t u = w[d], v = u >> 9;
            out[5 * d + 0] = gs_pair<3, 0>(u, zc);
            out[5 * d + 1] = gs_pair<3, 3>(u, zc);
            out[5 * d + 2] = gs_pair<3, 6>(u, zc);
            out[5 * d + 3] = gs_pair<3, 0>(v, zc);
            out[5 * d + 4] = gs_pair<3, 3>(v, zc);
        }
        // pair 15: value bit b of the (low, high) weight = bit (15, 31) of dword b -> bits 7+b
        const uint32_t e = ((w[0] >> 8) & 0x00800080u) | ((w[1] >> 7) & 0x01000100u) | ((w[2] >> 6) & 0x02000200u);
        out[15] = gs_pair<3, 7>(e, zc);
    }
}

// One pair of a lane's tile (compile-time pair index P = 4t + p), same arithmetic as dequant_lane_sd: lets a kernel spread
// the unpack of a tile over its MFMA steps instead of doing all 16 pairs in one block.
template <int BITS, int MODE, int P>
__device__ __forceinline__ h2 dequant_pair_sd(const uint32_t* w, const SdMeta& m) {
    if (BITS == 4) {
        const uint32_t u = w[P / 4];
        constexpr int p = P % 4;
        return sd_pair<4, MODE, 6>(p == 0 ? u << 6 : p == 1 ? u << 2 : p == 2 ? u >> 2 : u >> 6, m);
    } else if (BITS == 2) {
        const uint32_t u = w[P / 8];
        constexpr int q = P % 8;
        const uint32_t t = q < 3 ? u << 4 : q < 6 ? u >> 2 : u >> 8;
        constexpr int SH = 4 + 2 * (q % 3);
        return sd_pair<2, MODE, SH>(t, m);
    } else {
        if (P == 15) {
            const uint32_t e = ((w[0] >> 8) & 0x00800080u) | ((w[1] >> 7) & 0x01000100u) | ((w[2] >> 6) & 0x02000200u);
            return sd_pair<3, MODE, 7>(e, m);
        }
        const uint32_t u = w[P / 5 < 3 ? P / 5 : 2];
        constexpr int q = P % 5;
        const uint32_t t = q < 2 ? u << 4 : q < 4 ? u >> 2 : u >> 8;
        constexpr int SH = (q & 1) && q < 4 ? 7 : 4;
        return sd_pair<3, MODE, SH>(t, m);
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
__host__ __device__ __forceinline__ size_t native_meta_bytes(int N, int K, int gp = 1) {      // gp: (scale, zero) pairs per (row, tile) = 128 / group
    return (size_t)(N / TILE_N) * (size_t)(K / TILE_K) * TILE_N * 4u * (size_t)gp;
}
// pairs per tile for a caller's group size: 1 for 128 and its multiples (each source group's pair is replicated per tile), 2 / 4 for 64 / 32
__host__ __device__ __forceinline__ int meta_pairs(int group) { return group >= TILE_K ? 1 : TILE_K / group; }

// ---- wave-load of one lane's tile payload (BITS dwords), non-temporal: the
// weights are read exactly once per token, keep them from displacing x / KV.
template <int BITS>
struct LanePayload { uint32_t w[BITS]; };

#ifndef AMQ_NT
#define AMQ_NT 1
#endif
#if AMQ_NT
#define AMQ_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define AMQ_STREAM_LOAD(p) (*(p))
#endif

template <int BITS>
__device__ __forceinline__ LanePayload<BITS> load_payload(const uint32_t* tile_base, int lane) {
    LanePayload<BITS> r;
    if (BITS == 4) {
        u4 v = AMQ_STREAM_LOAD((const u4*)tile_base + lane);
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    } else if (BITS == 2) {
        u2 v = AMQ_STREAM_LOAD((const u2*)tile_base + lane);
        r.w[0] = v.x; r.w[1] = v.y;
    } else {
        // ONE 12-byte load (global_load_dwordx3; 4-byte alignment is enough on gfx950).  Written as one access rather than left to the compiler's
        // merging of three dword loads: the GEMV's counted waits (x_finish_dma) rely on a tile being two vector-memory operations at every bit-width.
        typedef uint32_t u3a __attribute__((ext_vector_type(3), aligned(4)));
        const u3a v = AMQ_STREAM_LOAD((const u3a*)(tile_base + 3 * lane));
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z;
    }
    return r;
}

// ---- groups finer than 128 (64 / 32 input columns per (scale, zero): HQQ's default group_size is 64, hqq/core/quantize.py:1078; the reference's GPTQ
// kernels index their scales by k / groupsize for any groupsize, AutoGPTQ/auto_gptq_kernel.cu:203, 303, 419) -----------------------------------
// The payload layout does not change; the meta array holds GP = 128 / group pairs per (row, tile): [N/16][K/128][16][GP] (one 8- or 16-byte load
// per lane and tile), pair s covering k in [group s, group (s + 1)) of the tile = the MFMA steps t with (t GP) >> 2 == s.  A lane's register
// pair P = 4 t + p therefore takes meta pair (P >> 2) GP >> 2: a compile-time choice of operand registers, no extra arithmetic.
template <int GP> struct MetaG { h2 p[GP]; };
template <int GP>
__device__ __forceinline__ MetaG<GP> load_meta_g(const h2* lane_base) {          // lane_base: this lane's first pair of the tile
    MetaG<GP> m;
    if (GP == 1) m.p[0] = as_h2(AMQ_STREAM_LOAD((const uint32_t*)lane_base));
    else if (GP == 2) { const u2 v = AMQ_STREAM_LOAD((const u2*)lane_base); m.p[0] = as_h2(v.x); m.p[1 % GP] = as_h2(v.y); }
    else { const u4 v = AMQ_STREAM_LOAD((const u4*)lane_base); m.p[0] = as_h2(v.x); m.p[1 % GP] = as_h2(v.y); m.p[2 % GP] = as_h2(v.z); m.p[3 % GP] = as_h2(v.w); }
    return m;
}
template <int GP> __host__ __device__ constexpr int meta_sub(int P) { return ((P >> 2) * GP) >> 2; }

// dequant_lane with per-pair meta (the reference-exact two-rounding form; used by the dequantize kernel)
template <int BITS, int MODE, int GP>
__device__ __forceinline__ void dequant_lane_g(const uint32_t* w, const MetaG<GP>& meta, h2* out) {
    h2 q[16];
    dequant_lane<BITS, MODE_FMA>(w, (h2){(_Float16)1.0f, (_Float16)0.0f}, q);   // fma(q, 1, 0) = q exactly: the integer fields as halves
#pragma unroll
    for (int P = 0; P < 16; ++P) {
        const h2 m = meta.p[meta_sub<GP>(P)];
        out[P] = apply_meta<MODE>(q[P], bcast(m.x), bcast(m.y));
    }
}
// dequant_lane_sd with per-pair meta (the matmul kernels' form, bit-identical to dequant_lane_g under dequant_lane_sd's conditions)
template <int BITS, int MODE, int P, int PEND>
__device__ __forceinline__ void dequant_sd_g_range(const uint32_t* w, const SdMeta& m, h2* out) {
    out[P] = dequant_pair_sd<BITS, MODE, P>(w, m);
    if constexpr (P + 1 < PEND) dequant_sd_g_range<BITS, MODE, P + 1, PEND>(w, m, out);
}
template <int BITS, int MODE, int GP, int S>
__device__ __forceinline__ void dequant_sd_g_sub(const uint32_t* w, const MetaG<GP>& meta, h2* out) {
    // the 16 / GP register pairs of sub-group S, with ITS scaled (scale, zero) only live meanwhile
    const SdMeta m = sd_meta<BITS, MODE>(meta.p[S]);
    dequant_sd_g_range<BITS, MODE, S * (16 / GP), (S + 1) * (16 / GP)>(w, m, out);
    if constexpr (S + 1 < GP) dequant_sd_g_sub<BITS, MODE, GP, S + 1>(w, meta, out);
}
template <int BITS, int MODE, int GP>
__device__ __forceinline__ void dequant_lane_sd_g(const uint32_t* w, const MetaG<GP>& meta, h2* out) {
    static_assert(meta_sub<GP>(16 / GP - 1) == 0 && meta_sub<GP>(16 / GP) == 1 % GP, "pairs of a sub-group are consecutive");
    dequant_sd_g_sub<BITS, MODE, GP, 0>(w, meta, out);
}


}  // namespace amq
