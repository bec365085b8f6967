// amq_bf16.hip -- the optional bfloat16 entry points (gfx950): dequantize and the few-row weight-streaming matmul for models quantized with
// compute_dtype = torch.bfloat16 (HQQLinear keeps scale / zero in the compute dtype and dequantizes in it: hqq/core/quantize.py:184-199, 396-407, 516).
// AMQ's own scripts quantize in fp16 (amq/amq_quantization_proxy.py:37) and the reference's GPTQ / FT modules assert fp16 (hqq/backends/ft.py:62), so
// this is a widening of the module boundary, not the product path: one straightforward kernel per operation, bit-exact weights, no fused prologues.
//
// Layout: the AMQ-T16 payload is the fp16 path's (amq_common.cuh); the meta words hold bfloat16 (scale, zero) pairs instead of fp16 ones --
// amq_repack_from_hqq copies the 16-bit patterns, so a bf16 model is repacked by the same call.
//
// Arithmetic (what torch does with bf16 tensors, on any device: operands widened to fp32, one operation, round to nearest even):
//     d = bf16(float(q) - float(z))      exact difference for every z >= 2^-16 (q <= 15, z has 8 significant bits)
//     w = bf16(float(d) * float(s))      the product of two 8-bit significands is exact in fp32
// => bit-identical to Quantizer.dequantize under compute_dtype = bfloat16 (tests: the reference's own output as a golden vector).
// There is no packed bf16 VALU arithmetic on gfx950: the unpack runs in fp32 (v_cvt_f32_ubyte, v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_bf16_f32),
// ~4.7 instructions per weight against 1.5 - 3 on the fp16 path, so the bf16 GEMV is VALU-bound at 2 - 4 bit: 5.9 / 11.5 / 12.1 us per launch on the 7B
// shapes (4096 x 4096 / gate-up / down, 4 bit, one row) against 4.6 / 7.6 / 7.7 for the fp16 kernel (tools/bf16_bench.py, profiles/r05_bf16.txt); two or
// four tile loads in flight per wave measure the same.
//
// gemv_bf16_kernel: 8 waves own one row-tile (16 output rows) at a time and split its K / 128 tiles round-robin; the unpacked 16 x 32 block is the A
// operand of v_mfma_f32_16x16x32_bf16, 16 x rows (clamped to M) the B operand, fp32 accumulation, a fixed-order cross-wave sum through LDS (deterministic).
// x is staged in LDS once per workgroup when M rows fit (XL), otherwise read per tile from global memory (L2-resident).
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// pair P = 4 t + p of a lane's payload: low weight in bits 0 .., high weight in bits 16 .. (amq_common.cuh, "bit placement")
template <int BITS>
__device__ __forceinline__ uint32_t pair_raw(const uint32_t* w, int P) {
    if (BITS == 4) return (w[P >> 2] >> (4 * (P & 3))) & 0x000F000Fu;
    if (BITS == 2) return (w[P >> 3] >> (2 * (P & 7))) & 0x00030003u;
    if (P < 15) return (w[P / 5] >> (3 * (P % 5))) & 0x00070007u;
    return ((w[0] >> 15) & 0x00010001u) | ((w[1] >> 14) & 0x00020002u) | ((w[2] >> 13) & 0x00040004u);
}

// (float) of byte B of a dword, where it lies: v_cvt_f32_ubyte<B>.  Written as asm: the compiler turns the C expression into shift + and + ubyte0
// (measured: 180 -> 150 VALU instructions per tile, -10 % per launch -- the kernel is VALU-bound)
template <int B>
__device__ __forceinline__ float cvt_ubyte(uint32_t v) {
    float f;
    if (B == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(v));
    else if (B == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(v));
    else if (B == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(v));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(v));
    return f;
}

// one lane's 8 weights of MFMA step t as bfloat16, from the meta word (scale | zero << 16).
// 4 / 2 bit: one mask isolates FOUR weights in the four bytes of a dword (byte 0 / 2 = low / high weight of a pair, byte 1 / 3 = of the pair 8 bits on), and
// v_cvt_f32_ubyte0..3 converts a byte where it lies; 3 bit: a pair at a time (fields do not fall on byte boundaries).
template <int BITS>
__device__ __forceinline__ b8 dequant_step_bf16(const uint32_t* w, int t, float s, float z) {
    float q[8];                                                                   // q[2 p], q[2 p + 1] = pair 4 t + p
    if (BITS == 4) {
        const uint32_t a = w[t] & 0x0F0F0F0Fu, b = (w[t] >> 4) & 0x0F0F0F0Fu;     // a: pairs 0 (bytes 0, 2) and 2 (bytes 1, 3); b: pairs 1 and 3
        q[0] = cvt_ubyte<0>(a); q[1] = cvt_ubyte<2>(a); q[4] = cvt_ubyte<1>(a); q[5] = cvt_ubyte<3>(a);
        q[2] = cvt_ubyte<0>(b); q[3] = cvt_ubyte<2>(b); q[6] = cvt_ubyte<1>(b); q[7] = cvt_ubyte<3>(b);
    } else if (BITS == 2) {
        // pairs P = 4 t + p live in dword P / 8 at bits 2 (P % 8): t even -> slots 0 .. 3 (bits 0 .. 7 of each half), t odd -> slots 4 .. 7 (bits 8 .. 15):
        // with the odd steps' fields left where they are, slot p + 4 is byte 1 / 3 of (u >> 2 p) & 0x03000300
        const uint32_t u = w[t >> 1];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (t & 1) {
                const uint32_t a = (u >> (2 * p)) & 0x03000300u;
                q[2 * p] = cvt_ubyte<1>(a); q[2 * p + 1] = cvt_ubyte<3>(a);
            } else {
                const uint32_t a = (u >> (2 * p)) & 0x00030003u;
                q[2 * p] = cvt_ubyte<0>(a); q[2 * p + 1] = cvt_ubyte<2>(a);
            }
        }
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t a = pair_raw<BITS>(w, 4 * t + p);
            q[2 * p] = cvt_ubyte<0>(a); q[2 * p + 1] = cvt_ubyte<2>(a);
        }
    }
    b8 out;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f2 qq = {q[2 * p], q[2 * p + 1]};
        const b2 d = __builtin_convertvector(qq - (f2){z, z}, b2);                 // rounding 1: W_r - zero
        const b2 v = __builtin_convertvector(__builtin_convertvector(d, f2) * (f2){s, s}, b2);   // rounding 2: * scale
        out[2 * p] = v[0];
        out[2 * p + 1] = v[1];
    }
    return out;
}
__device__ __forceinline__ float bf_lo(uint32_t m) { return __builtin_bit_cast(float, m << 16); }
__device__ __forceinline__ float bf_hi(uint32_t m) { return __builtin_bit_cast(float, m & 0xFFFF0000u); }

template <int BITS>
__global__ __launch_bounds__(256) void dequant_native_bf16_kernel(const uint32_t* qn, const uint32_t* mn, int N, int K, __bf16* out) {
    const int G = K >> 7;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t ntiles = (size_t)(N >> 4) * G;
    if (gid >= ntiles * 64) return;
    const int lane = (int)(gid & 63);
    const size_t tile = gid >> 6;
    const int rt = (int)(tile / G), g = (int)(tile % G);
    const int r = lane & 15, o = lane >> 4;
    const LanePayload<BITS> p = load_payload<BITS>(qn + tile * 64 * BITS, lane);
    const uint32_t m = mn[tile * 16 + r];
    const float s = bf_lo(m), z = bf_hi(m);
    __bf16* row = out + (size_t)(rt * 16 + r) * K + g * 128 + 8 * o;
#pragma unroll
    for (int t = 0; t < 4; ++t) *(b8*)(row + 32 * t) = dequant_step_bf16<BITS>(p.w, t, s, z);
}

// HQQ Format A -> bfloat16 W[N,K] without going through the native layout: amq_repack.hip's dequant_hqq_kernel (one thread = 8 consecutive columns of one
// packed row = C output rows' 16-byte stores) with the bf16 arithmetic above; any group size the fp16 kernel takes.
template <int BITS>
__global__ __launch_bounds__(256) void dequant_hqq_bf16_kernel(const void* wq, const uint16_t* scale, const uint16_t* zero, int R, __bf16* out, int gs) {
    constexpr int C = BITS == 4 ? 2 : BITS == 2 ? 4 : 10;
    const int step = BITS == 3 ? (R + 9) / 10 : R / C;     // packed rows (3 bit: rows zero-padded to a multiple of 10)
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int tpr = gs >> 3;
    const int i = (int)(gid / tpr), c8 = (int)(gid % tpr) * 8;
    if (i >= step) return;
    uint32_t q[8];
    if (BITS == 3) {
        const u4 a = *(const u4*)((const uint32_t*)wq + (size_t)i * gs + c8);
        const u4 b = *(const u4*)((const uint32_t*)wq + (size_t)i * gs + c8 + 4);
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    } else {
        const u2 p = *(const u2*)((const uint8_t*)wq + (size_t)i * gs + c8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { q[e] = (p.x >> (8 * e)) & 0xFFu; q[4 + e] = (p.y >> (8 * e)) & 0xFFu; }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int row = c * step + i;
        if (BITS == 3 && row >= R) continue;               // padding rows of the last chunks
        const float s = bf_lo(scale[row]), z = bf_lo(zero[row]);
        constexpr int width = BITS == 3 ? 3 : BITS;
        const int shift = BITS == 3 ? 27 - 3 * c : 8 - width * (c + 1);      // chunk 0 sits in the top bits
        b8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 d = (__bf16)((float)((q[e] >> shift) & ((1u << width) - 1u)) - z);      // rounding 1
            v[e] = (__bf16)((float)d * s);                                                        // rounding 2
        }
        *(b8*)(out + (size_t)row * gs + c8) = v;
    }
}

constexpr int BG_WAVES = 8, BG_THREADS = BG_WAVES * 64;
#ifndef AMQ_BF16_DEPTH
#define AMQ_BF16_DEPTH 2
#endif
constexpr int BG_DEPTH = AMQ_BF16_DEPTH;                                // tile loads in flight per wave
constexpr int BG_XPAD = 8;                                 // halves of padding per staged x row: 16 rows' ds_read_b128 then start in different banks
constexpr size_t BG_RED_BYTES = (size_t)BG_WAVES * 64 * sizeof(f4);
constexpr size_t BG_LDS_LIMIT = 152 * 1024;

struct GemvBf16Args {
    const void* x; const uint32_t* qn; const uint32_t* mn; const void* bias; const void* residual; void* y;
    int M, N, K, x_stride, y_stride;
};

template <int BITS, bool XL>
__global__ __launch_bounds__(BG_THREADS) void gemv_bf16_kernel(GemvBf16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f4* const red = (f4*)smem;                              // [wave][lane]
    __bf16* const xs = (__bf16*)(smem + BG_RED_BYTES);      // XL: [M][K + BG_XPAD]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: the ring's conditions are wave-uniform)
    const int r = lane & 15, o = lane >> 4;
    const int G = a.K >> 7, n_rt = a.N >> 4;
    const __bf16* const x = (const __bf16*)a.x;
    const int xrow = r < a.M ? r : a.M - 1;                 // B-operand rows past M repeat the last one: their output columns are never stored
    // a wave's tiles of a row-tile are g = wave, wave + 8, ...; BG_DEPTH of them are in flight in a register ring.  The first row-tile's loads leave
    // BEFORE x is staged (they do not depend on it), the next row-tile's before the cross-wave sum of the current one.
    LanePayload<BITS> p[BG_DEPTH];
    uint32_t mw[BG_DEPTH];
    auto issue = [&](int d, int rt, int g) {
        p[d] = load_payload<BITS>(a.qn + ((size_t)rt * G + g) * 64 * BITS, lane);
        mw[d] = a.mn[((size_t)rt * G + g) * 16 + r];
    };
    auto prime = [&](int rt) {
#pragma unroll
        for (int d = 0; d < BG_DEPTH; ++d)
            if (wave + BG_WAVES * d < G) issue(d, rt, wave + BG_WAVES * d);
    };
    int rt = blockIdx.x;
    if (rt < n_rt) prime(rt);
    if (XL) {
        const int k8 = a.K >> 3;                            // 16-byte pieces per row
        for (int i = tid; i < a.M * k8; i += BG_THREADS) {
            const int m = i / k8, c = i - m * k8;
            *(b8*)(xs + (size_t)m * (a.K + BG_XPAD) + 8 * c) = *(const b8*)(x + (size_t)m * a.x_stride + 8 * c);
        }
        __syncthreads();
    }
    const __bf16* const xl = XL ? xs + (size_t)xrow * (a.K + BG_XPAD) + 8 * o : x + (size_t)xrow * a.x_stride + 8 * o;

    for (; rt < n_rt; rt += gridDim.x) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int g0 = wave; g0 < G; g0 += BG_WAVES * BG_DEPTH) {
#pragma unroll
            for (int d = 0; d < BG_DEPTH; ++d) {
                const int g = g0 + BG_WAVES * d;            // (wave-uniform conditions)
                if (g < G) {
                    const float s = bf_lo(mw[d]), z = bf_hi(mw[d]);
                    b8 xf[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) xf[t] = *(const b8*)(xl + g * 128 + 32 * t);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dequant_step_bf16<BITS>(p[d].w, t, s, z), xf[t], acc, 0, 0, 0);
                    const int gn = g + BG_WAVES * BG_DEPTH;
                    if (gn < G) issue(d, rt, gn);           // the slot is free: its next tile of this row-tile
                }
            }
        }
        if (rt + (int)gridDim.x < n_rt) prime(rt + (int)gridDim.x);
        // acc[i] = partial y[m = r][n = 16 rt + 4 o + i]: sum over the waves in wave order
        red[wave * 64 + lane] = acc;
        __syncthreads();
        if (tid < 256) {
            const int l = tid >> 2, i = tid & 3;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < BG_WAVES; ++w) sum += red[w * 64 + l][i];
            const int mrow = l & 15, n = 16 * rt + 4 * (l >> 4) + i;
            if (mrow < a.M) {
                __bf16 v = (__bf16)sum;
                if (a.bias) v = (__bf16)((float)v + (float)((const __bf16*)a.bias)[n]);
                if (a.residual) v = (__bf16)((float)((const __bf16*)a.residual)[(size_t)mrow * a.y_stride + n] + (float)v);
                ((__bf16*)a.y)[(size_t)mrow * a.y_stride + n] = v;
            }
        }
        __syncthreads();
    }
}

static int bf16_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cus[dev & 63];
    if (c == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        c = v;
    }
    return c;
}

template <int BITS>
static hipError_t launch_gemv_bf16_bits(const GemvBf16Args& a, hipStream_t st) {
    const size_t xbytes = (size_t)a.M * (size_t)(a.K + BG_XPAD) * 2;
    const bool xl = BG_RED_BYTES + xbytes <= BG_LDS_LIMIT;           // (x_stride % 8 == 0 is the entry point's precondition: 16-byte pieces)
    const size_t lds = BG_RED_BYTES + (xl ? xbytes : 0);
    const int n_rt = a.N >> 4;
    // workgroups: every row-tile its own while they all fit the chip at once (a row-tile is the unit of work), else a grid-stride walk
    const int per_cu = lds <= 20 * 1024 ? 4 : lds <= 48 * 1024 ? 3 : lds <= 76 * 1024 ? 2 : 1;
    const int cap = bf16_cu_count() * per_cu;
    const int grid = n_rt < cap ? n_rt : cap;
    if (xl) {
        static unsigned long long done = 0;
        const hipError_t e = ensure_dyn_lds(done, (const void*)gemv_bf16_kernel<BITS, true>, (int)BG_LDS_LIMIT);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gemv_bf16_kernel<BITS, true>), dim3(grid), dim3(BG_THREADS), lds, st, a);
    } else {
        hipLaunchKernelGGL((gemv_bf16_kernel<BITS, false>), dim3(grid), dim3(BG_THREADS), lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_dequantize_bf16(int bits, const void* qn, const void* mn, int N, int K, void* w, hipStream_t st) {
    const size_t threads = (size_t)(N >> 4) * (K >> 7) * 64;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (bits == 4) hipLaunchKernelGGL((dequant_native_bf16_kernel<4>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)qn, (const uint32_t*)mn, N, K, (__bf16*)w);
    else if (bits == 3) hipLaunchKernelGGL((dequant_native_bf16_kernel<3>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)qn, (const uint32_t*)mn, N, K, (__bf16*)w);
    else hipLaunchKernelGGL((dequant_native_bf16_kernel<2>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)qn, (const uint32_t*)mn, N, K, (__bf16*)w);
    return hipGetLastError();
}

hipError_t launch_dequantize_hqq_bf16(int bits, const void* wq, const void* scale, const void* zero, int N, int K, void* w, hipStream_t st, int gs) {
    const int R = (int)((size_t)N * K / gs);
    const int step = bits == 3 ? (R + 9) / 10 : bits == 4 ? R / 2 : R / 4;
    const unsigned blocks = (unsigned)(((size_t)step * (gs >> 3) + 255) / 256);
    if (bits == 4) hipLaunchKernelGGL((dequant_hqq_bf16_kernel<4>), dim3(blocks), dim3(256), 0, st, wq, (const uint16_t*)scale, (const uint16_t*)zero, R, (__bf16*)w, gs);
    else if (bits == 3) hipLaunchKernelGGL((dequant_hqq_bf16_kernel<3>), dim3(blocks), dim3(256), 0, st, wq, (const uint16_t*)scale, (const uint16_t*)zero, R, (__bf16*)w, gs);
    else hipLaunchKernelGGL((dequant_hqq_bf16_kernel<2>), dim3(blocks), dim3(256), 0, st, wq, (const uint16_t*)scale, (const uint16_t*)zero, R, (__bf16*)w, gs);
    return hipGetLastError();
}

hipError_t launch_gemv_bf16(int bits, const void* x, const void* qn, const void* mn, const void* bias, const void* residual, void* y,
                            int M, int N, int K, int x_stride, int y_stride, hipStream_t st) {
    StreamDevice sd_(st);
    const GemvBf16Args a{x, (const uint32_t*)qn, (const uint32_t*)mn, bias, residual, y, M, N, K, x_stride, y_stride};
    if (bits == 4) return launch_gemv_bf16_bits<4>(a, st);
    if (bits == 3) return launch_gemv_bf16_bits<3>(a, st);
    return launch_gemv_bf16_bits<2>(a, st);
}

}  // namespace amq
