// amq_repack.hip -- load-time conversion of the reference's three weight
// formats into the native AMQ-T16 layout, and native/HQQ -> fp16 dequantize.
//
// Replaces the reference's host-side numpy packers
//   GPTQLinear.pack  (hqq/backends/autogptq.py:111-156, minutes for a 7B model)
//   pack_intweight   (hqq/backends/ft.py:15-55)
// and the standalone dequant of hqq/kernels/hqq_aten_cuda_kernel.cu (axis=0
// only there; AMQ uses axis=1, which is what is implemented here).
// One thread builds one lane's payload of one tile (32 weights): integer
// gathers only, bit-exact by construction; runs once per layer at load time.
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

// ---- read integer q[n][k] out of a reference format -----------------------
// FMT_HQQ  (Format A, hqq/core/bitpack.py:24-110; axis=1 grouping quantize.py:106-111)
// FMT_GPTQ (Format B, autogptq.py:121-156)
// FMT_AWQ  (Format C, ft.py:15-55)
// gs: the SOURCE format's group size: a multiple of 128 dividing K (the native layout keeps one (scale, zero) per 128 k: the source pair is
// replicated), or 64 / 32 (the native meta holds 128 / gs pairs per (row, tile), amq_common.cuh)
template <int FMT, int BITS>
__device__ __forceinline__ uint32_t fetch_q(const void* src, int n, int k, int N, int K, int gs) {
    if (FMT == FMT_HQQ) {
        const int G = K / gs;
        const int R = N * G;                 // rows of the [R, gs] grouped view
        const int row = n * G + k / gs;
        const int col = k % gs;
        if (BITS == 4) {
            const int step = R >> 1;
            const uint8_t b = ((const uint8_t*)src)[(size_t)(row % step) * gs + col];
            return row < step ? (b >> 4) : (b & 15);
        } else if (BITS == 2) {
            const int step = R >> 2;
            const uint8_t b = ((const uint8_t*)src)[(size_t)(row % step) * gs + col];
            return (b >> (6 - 2 * (row / step))) & 3;
        } else {
            const int step = (R + 9) / 10;   // rows zero-padded to a multiple of 10
            const uint32_t w = ((const uint32_t*)src)[(size_t)(row % step) * gs + col];
            return (w >> (27 - 3 * (row / step))) & 7;
        }
    } else if (FMT == FMT_GPTQ) {
        const uint32_t* qw = (const uint32_t*)src;
        if (BITS == 4) return (qw[(size_t)(k >> 3) * N + n] >> (4 * (k & 7))) & 15;
        if (BITS == 2) return (qw[(size_t)(k >> 4) * N + n] >> (2 * (k & 15))) & 3;
        const int j = k & 31;
        const uint32_t* r = qw + (size_t)(k >> 5) * 3 * N + n;
        if (j < 10) return (r[0] >> (3 * j)) & 7;
        if (j == 10) return (r[0] >> 30) | ((r[N] & 1) << 2);
        if (j < 21) return (r[N] >> (3 * (j - 11) + 1)) & 7;
        if (j == 21) return (r[N] >> 31) | ((r[2 * (size_t)N] & 3) << 1);
        return (r[2 * (size_t)N] >> (3 * (j - 22) + 2)) & 7;
    } else {  // FMT_AWQ, 4-bit only
        const int i = k & 31;
        const int p = i >> 3, a = (i & 7) >> 1, e = i & 1;
        const int i1 = 8 * a + 2 * p + e;                     // ft.py:21-24
        const int g = i1 >> 3, c = (i1 & 7) >> 1, e1 = i1 & 1;
        const int i2 = 8 * g + 4 * e1 + c;                    // ft.py:27-30
        const int kk = ((k >> 5) & 1) * 32 + i2;              // position in the 64-chunk
        const int v = 64 * (n & 3) + kk;                      // ft.py:33-41
        const uint16_t h = ((const uint16_t*)src)[(size_t)(n >> 2) * K + (size_t)(k >> 6) * 64 + (v >> 2)];
        return (h >> (4 * (v & 3))) & 15;
    }
}

template <int FMT, int BITS>
__global__ __launch_bounds__(256) void repack_kernel(const void* qsrc, const void* s_src, const void* z_src,
                                                     int N, int K, uint32_t* qn, h2* mn, int gs) {
    const int G = K >> 7;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t ntiles = (size_t)(N >> 4) * G;
    if (gid >= ntiles * 64) return;
    const int lane = (int)(gid & 63);
    const size_t tile = gid >> 6;
    const int rt = (int)(tile / G), g = (int)(tile % G);
    const int r = lane & 15, o = lane >> 4;
    const int n = rt * 16 + r;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = g * 128 + 32 * t + 8 * o + j;
            const uint32_t q = fetch_q<FMT, BITS>(qsrc, n, k, N, K, gs);
            if (BITS == 3 && t == 3 && j >= 6) {
                const int pos = (j & 1) ? 31 : 15;
                w[0] |= (q & 1u) << pos;
                w[1] |= ((q >> 1) & 1u) << pos;
                w[2] |= ((q >> 2) & 1u) << pos;
            } else {
                int dw, sh;
                native_slot(BITS, t, j, &dw, &sh);
                w[dw] |= q << sh;
            }
        }
    }
    uint32_t* dst = qn + (tile * 64 + lane) * BITS;
#pragma unroll
    for (int d = 0; d < BITS; ++d) dst[d] = w[d];

    const int gp = gs >= 128 ? 1 : 128 / gs;
    if (o < gp) {   // lane group o also writes the tile row's meta pair o (one pair per tile for groups >= 128)
        h2 m;
        const int gsrc = (g * 128 + o * (128 / gp)) / gs;  // the source group of this pair (groups > 128: its pair is replicated per tile)
        if (FMT == FMT_HQQ) {            // meta['scale'], meta['zero']: fp16 [N*K/gs, 1]
            const size_t row = (size_t)n * (K / gs) + gsrc;
            m.x = ((const _Float16*)s_src)[row];
            m.y = ((const _Float16*)z_src)[row];
        } else if (FMT == FMT_GPTQ) {    // scales fp32 [K/gs,N] = s ; zeros fp32 = fp16(z*s)
            m.x = (_Float16)((const float*)s_src)[(size_t)gsrc * N + n];
            m.y = -(_Float16)((const float*)z_src)[(size_t)gsrc * N + n];   // c = -zeros (auto_gptq_kernel.cu:200)
        } else {                          // scales fp16 [K/gs,N] ; scaled_zeros = -(z*s)
            m.x = ((const _Float16*)s_src)[(size_t)gsrc * N + n];
            m.y = ((const _Float16*)z_src)[(size_t)gsrc * N + n];
        }
        mn[(tile * 16 + r) * gp + o] = m;
    }
}

template <int BITS, int MODE, int GP = 1>
__global__ __launch_bounds__(256) void dequant_native_kernel(const uint32_t* qn, const h2* mn, int N, int K, _Float16* out) {
    const int G = K >> 7;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t ntiles = (size_t)(N >> 4) * G;
    if (gid >= ntiles * 64) return;
    const int lane = (int)(gid & 63);
    const size_t tile = gid >> 6;
    const int rt = (int)(tile / G), g = (int)(tile % G);
    const int r = lane & 15, o = lane >> 4;
    LanePayload<BITS> p = load_payload<BITS>(qn + tile * 64 * BITS, lane);
    h2 wv[16];
    if constexpr (GP == 1) dequant_lane<BITS, MODE>(p.w, mn[tile * 16 + r], wv);
    else dequant_lane_g<BITS, MODE, GP>(p.w, load_meta_g<GP>(mn + (tile * 16 + r) * GP), wv);
    _Float16* row = out + (size_t)(rt * 16 + r) * K + g * 128 + 8 * o;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        h8 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = wv[4 * t + q].x; v[2 * q + 1] = wv[4 * t + q].y; }
        *(h8*)(row + 32 * t) = v;
    }
}

// HQQ Format A -> fp16 W[N,K] without going through the native layout:
// W = ((unpack(W_q) - zero) * scale)  (quantize.py:184-199), the standalone dequant the search-time proxy evaluation runs
// thousands of times (amq/evaluation/evaluator.py:71-100; reference counterpart hqq/kernels/hqq_aten_cuda_kernel.cu:36-418,
// axis=0 only there).  Format A packs C row-chunks of the grouped view [R = N*K/128, 128] into one element
// (bitpack.py:24-110: C = 2 / 4 / 10 for 4 / 2 / 3 bit), so ONE packed row of 128 elements expands to C output rows.
// A thread takes 8 consecutive columns of one packed row -- one 8-byte (32-byte for 3 bit) load, no per-element index
// arithmetic -- and writes the C output rows' 8 halves as 16-byte stores; 16 threads cover a packed row, so every load
// instruction reads whole 128-byte lines and every store instruction writes 256 contiguous bytes per output row.
template <int BITS>
__global__ __launch_bounds__(256) void dequant_hqq_kernel(const void* wq, const _Float16* scale, const _Float16* zero,
                                                          int R, _Float16* out, int gs) {
    constexpr int C = BITS == 4 ? 2 : BITS == 2 ? 4 : 10;
    const int step = BITS == 3 ? (R + 9) / 10 : R / C;     // packed rows (3 bit: rows zero-padded to a multiple of 10)
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int tpr = gs >> 3;                               // threads per packed row of gs elements (gs = 128: 16)
    const int i = (int)(gid / tpr), c8 = (int)(gid % tpr) * 8;
    if (i >= step) return;
    uint32_t q[8];                                         // the 8 packed elements, widened
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
        const _Float16 s = scale[row], z = zero[row];
        constexpr int width = BITS == 3 ? 3 : BITS;
        const int shift = BITS == 3 ? 27 - 3 * c : 8 - width * (c + 1);      // chunk 0 sits in the top bits
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 f = (_Float16)(float)((q[e] >> shift) & ((1u << width) - 1u));
            const _Float16 d = f - z;                      // fp16 rounding #1
            v[e] = d * s;                                  // fp16 rounding #2
        }
        *(h8*)(out + (size_t)row * gs + c8) = v;
    }
}

// mul[i] += (float)y[i]  -- the in-place fp32 accumulation of vecquant*matmul_faster_old (auto_gptq_kernel.cu:224)
__global__ __launch_bounds__(256) void accumulate_f32_kernel(float* mul, const _Float16* y, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) mul[i] += (float)y[i];
}

hipError_t launch_accumulate_f32(void* mul, const void* y, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(accumulate_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (float*)mul, (const _Float16*)y, n);
    return hipGetLastError();
}

template <int FMT>
static hipError_t repack_bits(int bits, const void* q, const void* s, const void* z, int N, int K,
                              void* qn, void* mn, hipStream_t st, int gs) {
    const size_t threads = (size_t)(N >> 4) * (K >> 7) * 64;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (bits == 4) hipLaunchKernelGGL((repack_kernel<FMT, 4>), dim3(blocks), dim3(256), 0, st, q, s, z, N, K, (uint32_t*)qn, (h2*)mn, gs);
    else if (bits == 3) hipLaunchKernelGGL((repack_kernel<FMT, 3>), dim3(blocks), dim3(256), 0, st, q, s, z, N, K, (uint32_t*)qn, (h2*)mn, gs);
    else hipLaunchKernelGGL((repack_kernel<FMT, 2>), dim3(blocks), dim3(256), 0, st, q, s, z, N, K, (uint32_t*)qn, (h2*)mn, gs);
    return hipGetLastError();
}

hipError_t launch_repack(int fmt, int bits, const void* q, const void* s, const void* z, int N, int K,
                         void* qn, void* mn, hipStream_t st, int gs) {
    if (fmt == FMT_HQQ) return repack_bits<FMT_HQQ>(bits, q, s, z, N, K, qn, mn, st, gs);
    if (fmt == FMT_GPTQ) return repack_bits<FMT_GPTQ>(bits, q, s, z, N, K, qn, mn, st, gs);
    const size_t threads = (size_t)(N >> 4) * (K >> 7) * 64;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL((repack_kernel<FMT_AWQ, 4>), dim3(blocks), dim3(256), 0, st, q, s, z, N, K, (uint32_t*)qn, (h2*)mn, gs);
    return hipGetLastError();
}

hipError_t launch_dequantize(int bits, int mode, const void* qn, const void* mn, int N, int K, void* w, hipStream_t st, int gp) {
    const size_t threads = (size_t)(N >> 4) * (K >> 7) * 64;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
#define AMQ_DQ(B, MD, G_) hipLaunchKernelGGL((dequant_native_kernel<B, MD, G_>), dim3(blocks), dim3(256), 0, st, \
                                             (const uint32_t*)qn, (const h2*)mn, N, K, (_Float16*)w)
#define AMQ_DQ_BM(G_)                                                                                                       \
    do {                                                                                                                    \
        if (mode == MODE_HQQ) { if (bits == 4) AMQ_DQ(4, MODE_HQQ, G_); else if (bits == 3) AMQ_DQ(3, MODE_HQQ, G_); else AMQ_DQ(2, MODE_HQQ, G_); } \
        else { if (bits == 4) AMQ_DQ(4, MODE_FMA, G_); else if (bits == 3) AMQ_DQ(3, MODE_FMA, G_); else AMQ_DQ(2, MODE_FMA, G_); }             \
    } while (0)
    if (gp == 1) AMQ_DQ_BM(1);
    else if (gp == 2) AMQ_DQ_BM(2);
    else if (gp == 4) AMQ_DQ_BM(4);
    else return hipErrorInvalidValue;
#undef AMQ_DQ_BM
#undef AMQ_DQ
    return hipGetLastError();
}

hipError_t launch_dequantize_hqq(int bits, const void* wq, const void* scale, const void* zero, int N, int K,
                                 void* w, hipStream_t st, int gs) {
    const int R = (int)((size_t)N * K / gs);
    const int step = bits == 3 ? (R + 9) / 10 : bits == 4 ? R / 2 : R / 4;
    const unsigned blocks = (unsigned)(((size_t)step * (gs >> 3) + 255) / 256);
    if (bits == 4) hipLaunchKernelGGL((dequant_hqq_kernel<4>), dim3(blocks), dim3(256), 0, st, wq, (const _Float16*)scale, (const _Float16*)zero, R, (_Float16*)w, gs);
    else if (bits == 3) hipLaunchKernelGGL((dequant_hqq_kernel<3>), dim3(blocks), dim3(256), 0, st, wq, (const _Float16*)scale, (const _Float16*)zero, R, (_Float16*)w, gs);
    else hipLaunchKernelGGL((dequant_hqq_kernel<2>), dim3(blocks), dim3(256), 0, st, wq, (const _Float16*)scale, (const _Float16*)zero, R, (_Float16*)w, gs);
    return hipGetLastError();
}

}  // namespace amq
