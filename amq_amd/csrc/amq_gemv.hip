// amq_gemv.hip -- weight-streaming y = x . W^T for few rows (decode), gfx950.
//
// Replaces, for rows < 8/128, the reference's
//   VecQuant{2,3,4}MatMulKernelFaster_old (amq/kernel/AutoGPTQ/auto_gptq_kernel.cu:160-440)
//   gemv_kernel<2,Batch,256,128>          (amq/kernel/ft/quantization_new/gemv/gemv_cuda.cu:73-204)
// with one kernel family over the native AMQ-T16 layout (amq_common.cuh).
//
// Structure (HBM-bound: every byte of W is read exactly once, nothing else matters):
//   * one workgroup = 16 output rows x all of K; its weight bytes are one
//     contiguous range.  NW waves split the K/128 tiles round-robin.
//   * each wave issues its first U tile loads (16 B/lane, non-temporal)
//     BEFORE x is staged, so the HBM round trip overlaps the prologue.
//   * x (optionally RMSNorm'ed or SiLU(gate)*up) is staged once in LDS as fp16;
//     lanes read their 8-wide k-octets with ds_read_b128 (4 addresses per
//     instruction, broadcast over 16 lanes -> conflict free).
//   * M == 1: unpack (v_and_or + v_pk_*_f16) -> v_dot2c_f32_f16 into fp32,
//     2-step wavefront reduction over the 4 k-octet lane groups, fixed-order
//     cross-wave sum through LDS (deterministic, no atomics).
//   * 2 <= M <= 64: the unpacked fp16x8 register block IS the MFMA B operand;
//     v_mfma_f32_16x16x32_f16 against x rows read from LDS.  W never touches LDS.
//   * several linears that share x (q/k/v, gate/up) with different bit-widths
//     run as segments of ONE launch (wave-uniform switch on bits).
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

constexpr int GEMV_NW = 8;           // waves per workgroup
constexpr int GEMV_THREADS = GEMV_NW * 64;
constexpr int GEMV_U = 4;            // tiles in flight per wave
constexpr int XPAD = 8;              // halves of padding per staged x row (16 B)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + __expf(-g)); }

// ---------------------------------------------------------------- staging
// Writes the (transformed) activations into LDS as fp16 [M][xs].
template <int PRO>
__device__ __forceinline__ void stage_x(const GemvArgs& a, _Float16* xl, float* red, int xs) {
    const int tid = threadIdx.x;
    const int K = a.K;
    const int chunks = K >> 3;      // 8 halves per chunk
    for (int m = 0; m < a.M; ++m) {
        const _Float16* xrow = (const _Float16*)a.x + (size_t)m * a.x_stride;
        _Float16* lrow = xl + (size_t)m * xs;
        if (PRO == PRO_NONE) {
            for (int c = tid; c < chunks; c += GEMV_THREADS)
                *(h8*)(lrow + 8 * c) = *(const h8*)(xrow + 8 * c);
        } else if (PRO == PRO_SILU_MUL) {
            // x = fp16(fp16(silu(gate)) * up)  -- HF LlamaMLP: act_fn(gate) * up
            const _Float16* urow = (const _Float16*)a.x2 + (size_t)m * a.x_stride;
            for (int c = tid; c < chunks; c += GEMV_THREADS) {
                h8 g = *(const h8*)(xrow + 8 * c);
                h8 u = *(const h8*)(urow + 8 * c);
                h8 r;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    _Float16 s = (_Float16)silu_f((float)g[i]);
                    r[i] = s * u[i];
                }
                *(h8*)(lrow + 8 * c) = r;
            }
        } else {  // PRO_RMSNORM
            float ss = 0.f;
            for (int c = tid; c < chunks; c += GEMV_THREADS) {
                h8 v = *(const h8*)(xrow + 8 * c);
                *(h8*)(lrow + 8 * c) = v;
#pragma unroll
                for (int i = 0; i < 8; ++i) { float f = (float)v[i]; ss += f * f; }
            }
            ss = wave_sum(ss);
            __syncthreads();            // previous row's readers of red[] are done
            if ((tid & 63) == 0) red[tid >> 6] = ss;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < GEMV_NW; ++w) tot += red[w];
            const float rstd = rsqrtf(tot / (float)K + a.eps);
            // HF LlamaRMSNorm: weight * (x.float() * rstd).to(fp16)
            for (int c = tid; c < chunks; c += GEMV_THREADS) {
                h8 v = *(h8*)(lrow + 8 * c);
                h8 gm = *(const h8*)((const _Float16*)a.gamma + 8 * c);
                h8 r;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    _Float16 nrm = (_Float16)((float)v[i] * rstd);
                    r[i] = gm[i] * nrm;
                }
                *(h8*)(lrow + 8 * c) = r;
            }
        }
    }
}

// ---------------------------------------------------------------- epilogue
__device__ __forceinline__ void store_out(const GemvSeg& s, int m, int n, float acc) {
    _Float16 y = (_Float16)acc;                                   // fp16(matmul)
    if (s.bias) y = y + ((const _Float16*)s.bias)[n];             // out + bias      (fp16 add)
    if (s.residual) y = ((const _Float16*)s.residual)[(size_t)m * s.y_stride + n] + y;  // residual + out
    ((_Float16*)s.y)[(size_t)m * s.y_stride + n] = y;
}

// ---------------------------------------------------------------- body
template <int BITS, int MODE, int PRO, bool MFMA>
__device__ __forceinline__ void gemv_body(const GemvArgs& a, const GemvSeg& s, int rt,
                                          _Float16* xl, float* red, int xs) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = a.K >> 7;
    const int r = lane & 15, o = lane >> 4;
    const uint32_t* qw = (const uint32_t*)s.qweight + (size_t)rt * G * (64 * BITS);
    const h2* mt = (const h2*)s.meta + (size_t)rt * G * 16 + r;

    LanePayload<BITS> pay[GEMV_U];
    h2 meta[GEMV_U];
    auto issue = [&](int g0) {
#pragma unroll
        for (int u = 0; u < GEMV_U; ++u) {
            const int g = g0 + u * GEMV_NW;
            if (g < G) {
                pay[u] = load_payload<BITS>(qw + (size_t)g * (64 * BITS), lane);
                meta[u] = as_h2(__builtin_nontemporal_load((const uint32_t*)(mt + (size_t)g * 16)));
            }
        }
    };
    issue(wave);                    // HBM requests leave before the prologue
    stage_x<PRO>(a, xl, red, xs);
    __syncthreads();

    constexpr int MB = MFMA ? 4 : 1;           // 16-row m-blocks (MFMA) / scalar acc
    float acc1[4] = {0.f, 0.f, 0.f, 0.f};
    f4 accm[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) accm[i] = (f4){0.f, 0.f, 0.f, 0.f};
    const int mblocks = (a.M + 15) >> 4;

    for (int g0 = wave; g0 < G; g0 += GEMV_NW * GEMV_U) {
        LanePayload<BITS> cur[GEMV_U];
        h2 cmeta[GEMV_U];
#pragma unroll
        for (int u = 0; u < GEMV_U; ++u) { cur[u] = pay[u]; cmeta[u] = meta[u]; }
        if (g0 + GEMV_NW * GEMV_U < G) issue(g0 + GEMV_NW * GEMV_U);   // prefetch next batch
#pragma unroll
        for (int u = 0; u < GEMV_U; ++u) {
            const int g = g0 + u * GEMV_NW;
            if (g < G) {
                h2 wv[16];
                dequant_lane<BITS, MODE>(cur[u].w, cmeta[u], wv);
                const int kbase = (g << 7) + 8 * o;
                if (!MFMA) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const h8 xv = *(const h8*)(xl + kbase + 32 * t);
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            h2 xp = {xv[2 * p], xv[2 * p + 1]};
                            acc1[t] = __builtin_amdgcn_fdot2(wv[4 * t + p], xp, acc1[t], false);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        h8 b;
#pragma unroll
                        for (int p = 0; p < 4; ++p) { b[2 * p] = wv[4 * t + p].x; b[2 * p + 1] = wv[4 * t + p].y; }
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) {
                            if (mb < mblocks) {
                                int m = mb * 16 + r;
                                m = m < a.M ? m : a.M - 1;      // rows >= M: any finite data, result unused
                                const h8 av = *(const h8*)(xl + (size_t)m * xs + kbase + 32 * t);
                                accm[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b, accm[mb], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- reduction: lanes (k-octet groups) -> waves (fixed order) -> y
    __syncthreads();                // everyone is done reading xl; reuse red[]
    if (!MFMA) {
        float v = (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) red[wave * 16 + lane] = v;
        __syncthreads();
        if (threadIdx.x < 16) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < GEMV_NW; ++w) tot += red[w * 16 + threadIdx.x];
            store_out(s, 0, rt * 16 + threadIdx.x, tot);
        }
    } else {
        // accm[mb][i] = D[m = mb*16 + 4*o + i][n = r]
        float* redm = red;          // [NW][64 rows][16 cols]
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            if (mb < mblocks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    redm[(wave * 64 + mb * 16 + 4 * o + i) * 16 + r] = accm[mb][i];
        __syncthreads();
        for (int e = threadIdx.x; e < a.M * 16; e += GEMV_THREADS) {
            const int m = e >> 4, n = e & 15;
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < GEMV_NW; ++w) tot += redm[(w * 64 + m) * 16 + n];
            store_out(s, m, rt * 16 + n, tot);
        }
    }
}

template <int PRO, bool MFMA>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xs = a.K + XPAD;
    _Float16* xl = (_Float16*)smem;
    float* red = (float*)(smem + (((size_t)a.M * xs * 2 + 15) & ~(size_t)15));

    int sidx = 0;
#pragma unroll
    for (int i = 1; i < GEMV_MAX_SEG; ++i)
        if (i < a.nseg && (int)blockIdx.x >= a.seg[i].wg_begin) sidx = i;
    const GemvSeg& s = a.seg[sidx];
    const int rt = (int)blockIdx.x - s.wg_begin;
    const int key = s.bits * 2 + s.mode;
    switch (key) {
        case 4 * 2 + MODE_HQQ: gemv_body<4, MODE_HQQ, PRO, MFMA>(a, s, rt, xl, red, xs); break;
        case 3 * 2 + MODE_HQQ: gemv_body<3, MODE_HQQ, PRO, MFMA>(a, s, rt, xl, red, xs); break;
        case 2 * 2 + MODE_HQQ: gemv_body<2, MODE_HQQ, PRO, MFMA>(a, s, rt, xl, red, xs); break;
        case 4 * 2 + MODE_FMA: gemv_body<4, MODE_FMA, PRO, MFMA>(a, s, rt, xl, red, xs); break;
        case 3 * 2 + MODE_FMA: gemv_body<3, MODE_FMA, PRO, MFMA>(a, s, rt, xl, red, xs); break;
        default:               gemv_body<2, MODE_FMA, PRO, MFMA>(a, s, rt, xl, red, xs); break;
    }
}

size_t gemv_lds_bytes(int M, int K) {
    const size_t xbytes = (((size_t)M * (K + XPAD) * 2) + 15) & ~(size_t)15;
    const size_t red = (M == 1) ? (size_t)GEMV_NW * 16 * 4 : (size_t)GEMV_NW * 64 * 16 * 4;
    return xbytes + red;
}

template <int PRO, bool MFMA>
static hipError_t launch_one(const GemvArgs& a, int total_wg, size_t lds, hipStream_t st) {
    auto kern = gemv_kernel<PRO, MFMA>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(total_wg), dim3(GEMV_THREADS), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_gemv(const GemvArgs& a, int total_wg, hipStream_t st) {
    const size_t lds = gemv_lds_bytes(a.M, a.K);
    const bool mfma = a.M > 1;
    switch (a.prologue) {
        case PRO_NONE:
            return mfma ? launch_one<PRO_NONE, true>(a, total_wg, lds, st) : launch_one<PRO_NONE, false>(a, total_wg, lds, st);
        case PRO_RMSNORM:
            return mfma ? launch_one<PRO_RMSNORM, true>(a, total_wg, lds, st) : launch_one<PRO_RMSNORM, false>(a, total_wg, lds, st);
        default:
            return mfma ? launch_one<PRO_SILU_MUL, true>(a, total_wg, lds, st) : launch_one<PRO_SILU_MUL, false>(a, total_wg, lds, st);
    }
}

}  // namespace amq
