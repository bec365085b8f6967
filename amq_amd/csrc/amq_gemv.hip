// amq_gemv.hip -- weight-streaming y = x . W^T for few rows (decode), gfx950.
//
// Replaces, for rows < 8/128, the reference's
//   VecQuant{2,3,4}MatMulKernelFaster_old (amq/kernel/AutoGPTQ/auto_gptq_kernel.cu:160-440)
//   gemv_kernel<2,Batch,256,128>          (amq/kernel/ft/quantization_new/gemv/gemv_cuda.cu:73-204)
// with one kernel family over the native AMQ-T16 layout (amq_common.cuh).
//
// Structure (HBM-bound by design: every byte of W is read exactly once):
//   * one workgroup = 16 output rows x all of K; its weight bytes are one
//     contiguous range.  NW waves (4 / 8 / 16, picked from the grid size) take
//     the K/128 tiles round-robin.
//   * each wave keeps U tile loads (16/12/8 B per lane, non-temporal) in flight
//     in a rolling register ring: a slot is re-issued as soon as it has been
//     consumed, and the first U leave BEFORE x is staged, so the HBM round trip
//     overlaps the prologue.
//   * x (optionally RMSNorm'ed or SiLU(gate)*up) is staged once in LDS as fp16;
//     lanes read their 8-wide k-octets with ds_read_b128 (4 addresses per
//     instruction, broadcast over 16 lanes -> conflict free).
//   * the unpacked fp16x8 register block IS the MFMA B operand (layout chosen
//     for that): v_mfma_f32_16x16x32_f16 against the x rows; W never touches
//     LDS, the k-octet reduction happens inside the matrix core, and M = 1..16
//     costs the same VALU work.  (M == 1 can alternatively run the
//     v_dot2c_f32_f16 + wavefront-shuffle reduction body: GEMV_FLAG_DOT; it is
//     slower on gfx950 because dot2c is a 4-cycle VOP3 issue -- DESIGN.md.)
//   * fixed-order cross-wave sum through LDS: deterministic, no atomics.
//   * several linears that share x (q/k/v, gate/up) with different bit-widths
//     run as segments of ONE launch (wave-uniform switch on bits).
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

constexpr int XPAD = 8;              // halves of padding per staged x row (16 B)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + __expf(-g)); }

// ---------------------------------------------------------------- staging
// Writes the (transformed) activations into LDS as fp16 [M][xs].
template <int PRO, int NW>
__device__ __forceinline__ void stage_x(const GemvArgs& a, _Float16* xl, float* red, int xs) {
    constexpr int THREADS = NW * 64;
    const int tid = threadIdx.x;
    const int K = a.K;
    const int chunks = K >> 3;      // 8 halves per chunk
    for (int m = 0; m < a.M; ++m) {
        const _Float16* xrow = (const _Float16*)a.x + (size_t)m * a.x_stride;
        _Float16* lrow = xl + (size_t)m * xs;
        if (PRO == PRO_NONE) {
            for (int c = tid; c < chunks; c += THREADS)
                *(h8*)(lrow + 8 * c) = *(const h8*)(xrow + 8 * c);
        } else if (PRO == PRO_SILU_MUL) {
            // x = fp16(fp16(silu(gate)) * up)  -- HF LlamaMLP: act_fn(gate) * up
            const _Float16* urow = (const _Float16*)a.x2 + (size_t)m * a.x_stride;
            for (int c = tid; c < chunks; c += THREADS) {
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
            for (int c = tid; c < chunks; c += THREADS) {
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
            for (int w = 0; w < NW; ++w) tot += red[w];
            const float rstd = rsqrtf(tot / (float)K + a.eps);
            // HF LlamaRMSNorm: weight * (x.float() * rstd).to(fp16)
            for (int c = tid; c < chunks; c += THREADS) {
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
template <int BITS, int MODE, int PRO, int NW, bool DOT, int GEMV_U>
__device__ __forceinline__ void gemv_body(const GemvArgs& a, const GemvSeg& s, int rt,
                                          _Float16* xl, float* red, int xs) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = a.K >> 7;
    const int r = lane & 15, o = lane >> 4;
    const uint32_t* qw = (const uint32_t*)s.qweight + (size_t)rt * G * (64 * BITS);
    const h2* mt = (const h2*)s.meta + (size_t)rt * G * 16 + r;
    const int nt = (G - wave + NW - 1) / NW;          // tiles owned by this wave: g = wave + i*NW

    LanePayload<BITS> pay[GEMV_U];
    h2 meta[GEMV_U];
#define AMQ_ISSUE(slot, i)                                                                       \
    do {                                                                                         \
        const int g_ = wave + (i) * NW;                                                          \
        pay[slot] = load_payload<BITS>(qw + (size_t)g_ * (64 * BITS), lane);                      \
        meta[slot] = as_h2(AMQ_STREAM_LOAD((const uint32_t*)(mt + (size_t)g_ * 16)));            \
    } while (0)

#pragma unroll
    for (int u = 0; u < GEMV_U; ++u)
        if (u < nt) AMQ_ISSUE(u, u);                  // HBM requests leave before the prologue

    stage_x<PRO, NW>(a, xl, red, xs);
    __syncthreads();

    float acc1[4] = {0.f, 0.f, 0.f, 0.f};
    f4 accm = (f4){0.f, 0.f, 0.f, 0.f};
    int mrow = r < a.M ? r : a.M - 1;                 // A rows >= M: any finite data, result unused
    const _Float16* xrow = xl + (size_t)mrow * xs + 8 * o;

    // consume one tile out of ring slot `slot`
#define AMQ_COMPUTE(slot, i)                                                                     \
    do {                                                                                         \
        h2 wv[16];                                                                               \
        dequant_lane_sd<BITS, MODE>(pay[slot].w, meta[slot], wv);                                \
        const int kbase = (wave + (i) * NW) << 7;                                                \
        if (DOT) {                                                                               \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                      \
                const h8 xv = *(const h8*)(xl + kbase + 8 * o + 32 * t);                         \
                _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                  \
                    h2 xp = {xv[2 * p], xv[2 * p + 1]};                                          \
                    acc1[t] = __builtin_amdgcn_fdot2(wv[4 * t + p], xp, acc1[t], false);         \
                }                                                                                \
            }                                                                                    \
        } else {                                                                                 \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                      \
                h8 b;                                                                            \
                _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                  \
                    b[2 * p] = wv[4 * t + p].x; b[2 * p + 1] = wv[4 * t + p].y;                  \
                }                                                                                \
                const h8 av = *(const h8*)(xrow + kbase + 32 * t);                               \
                accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b, accm, 0, 0, 0);             \
            }                                                                                    \
        }                                                                                        \
    } while (0)

    // Software pipeline over this wave's tiles.  A slot is refilled only after
    // its tile has been fully consumed (sched_barrier keeps the compiler from
    // hoisting the load into temporaries + a vmcnt(0)/v_mov rotation, which is
    // what a naive ring compiles to); the main loop refills unconditionally so
    // the waits stay counted (vmcnt((U-1)*loads)), the tail drains.
    int i = 0;
    for (; i + 2 * GEMV_U <= nt; i += GEMV_U) {
#pragma unroll
        for (int u = 0; u < GEMV_U; ++u) {
            AMQ_COMPUTE(u, i + u);
            __builtin_amdgcn_sched_barrier(0);
            AMQ_ISSUE(u, i + u + GEMV_U);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < GEMV_U; ++u) {
        if (i + u < nt) {
            AMQ_COMPUTE(u, i + u);
            __builtin_amdgcn_sched_barrier(0);
            if (i + u + GEMV_U < nt) AMQ_ISSUE(u, i + u + GEMV_U);
        }
    }
    i += GEMV_U;
#pragma unroll
    for (int u = 0; u < GEMV_U; ++u)
        if (i + u < nt) AMQ_COMPUTE(u, i + u);
#undef AMQ_ISSUE
#undef AMQ_COMPUTE

    // ---- cross-wave reduction in a fixed order -> y
    __syncthreads();                // everyone is done reading xl; red[] is free
    if (DOT) {
        float v = (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) red[wave * 16 + lane] = v;
        __syncthreads();
        if (threadIdx.x < 16) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[w * 16 + threadIdx.x];
            store_out(s, 0, rt * 16 + threadIdx.x, tot);
        }
    } else {
        // accm[i] = D[m = 4*o + i][n = r]
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * o + i < a.M) red[(wave * 16 + 4 * o + i) * 16 + r] = accm[i];
        __syncthreads();
        for (int e = threadIdx.x; e < a.M * 16; e += NW * 64) {
            const int m = e >> 4, n = e & 15;
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[(w * 16 + m) * 16 + n];
            store_out(s, m, rt * 16 + n, tot);
        }
    }
}

template <int PRO, int NW, bool DOT, int U>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(GemvArgs a) {
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
        case 4 * 2 + MODE_HQQ: gemv_body<4, MODE_HQQ, PRO, NW, DOT, U>(a, s, rt, xl, red, xs); break;
        case 3 * 2 + MODE_HQQ: gemv_body<3, MODE_HQQ, PRO, NW, DOT, U>(a, s, rt, xl, red, xs); break;
        case 2 * 2 + MODE_HQQ: gemv_body<2, MODE_HQQ, PRO, NW, DOT, U>(a, s, rt, xl, red, xs); break;
        case 4 * 2 + MODE_FMA: gemv_body<4, MODE_FMA, PRO, NW, DOT, U>(a, s, rt, xl, red, xs); break;
        case 3 * 2 + MODE_FMA: gemv_body<3, MODE_FMA, PRO, NW, DOT, U>(a, s, rt, xl, red, xs); break;
        default:               gemv_body<2, MODE_FMA, PRO, NW, DOT, U>(a, s, rt, xl, red, xs); break;
    }
}

size_t gemv_lds_bytes(int M, int K) {
    const size_t xbytes = (((size_t)M * (K + XPAD) * 2) + 15) & ~(size_t)15;
    const size_t red = (size_t)16 /*max NW*/ * 16 * 16 * 4;
    return xbytes + red;
}

int gemv_pick_waves(int total_wg, int K) {
    // measured on MI355X (tools/microbench.py sweeps, profiles/): 8 waves x 2 tiles
    // in flight is the best or within 3% of it for every Llama shape; tiny K
    // falls back to 4 waves so each wave still owns >= 2 tiles.
    (void)total_wg;
    const int G = K >> 7;
    return G >= 16 ? 8 : 4;
}

template <int PRO, int NW, bool DOT, int U>
static hipError_t launch_one(const GemvArgs& a, int total_wg, size_t lds, hipStream_t st) {
    auto kern = gemv_kernel<PRO, NW, DOT, U>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(total_wg), dim3(NW * 64), lds, st, a);
    return hipGetLastError();
}

template <int PRO, int NW>
static hipError_t launch_nw(const GemvArgs& a, int total_wg, size_t lds, hipStream_t st) {
    const bool dot = (a.flags & GEMV_FLAG_DOT) && a.M == 1;
    const int u = a.force_depth ? a.force_depth : 2;
    if (dot) return launch_one<PRO, NW, true, 4>(a, total_wg, lds, st);
    if (u == 2) return launch_one<PRO, NW, false, 2>(a, total_wg, lds, st);
    return launch_one<PRO, NW, false, 4>(a, total_wg, lds, st);
}

template <int PRO>
static hipError_t launch_pro(const GemvArgs& a, int total_wg, size_t lds, hipStream_t st) {
    int nw = a.force_waves ? a.force_waves : gemv_pick_waves(total_wg, a.K);
    if (nw == 2) return launch_nw<PRO, 2>(a, total_wg, lds, st);
    if (nw == 4) return launch_nw<PRO, 4>(a, total_wg, lds, st);
    if (nw == 8) return launch_nw<PRO, 8>(a, total_wg, lds, st);
    return launch_nw<PRO, 16>(a, total_wg, lds, st);
}

hipError_t launch_gemv(const GemvArgs& a, int total_wg, hipStream_t st) {
    const size_t lds = gemv_lds_bytes(a.M, a.K);
    switch (a.prologue) {
        case PRO_NONE: return launch_pro<PRO_NONE>(a, total_wg, lds, st);
        case PRO_RMSNORM: return launch_pro<PRO_RMSNORM>(a, total_wg, lds, st);
        default: return launch_pro<PRO_SILU_MUL>(a, total_wg, lds, st);
    }
}

}  // namespace amq
