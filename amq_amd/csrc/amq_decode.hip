// amq_decode.hip -- what surrounds the quantized linears in one decode step
// (SURVEY.md 8 f-1), so that per-layer GB/s turns into end-to-end tokens/s:
//
//   rmsnorm_kernel      LlamaRMSNorm; replaces FT generalT5LayerNorm
//                       (amq/kernel/ft/layernorm/layernorm.cu:25-51)
//   gemv_f16w_kernel    y = x . W^T with fp16 W (lm_head is NOT quantized in AMQ:
//                       monkeypatch/ftllama_modeling.py:465), optional RMSNorm prologue
//   attn_decode_kernel  RoPE(q,k) + KV-cache append + softmax(q K^T / sqrt(d)) V for one
//                       new token; replaces FT masked_multihead_attention
//                       (amq/kernel/ft/attention/decoder_masked_multihead_attention_template.hpp:865)
//                       with HF-Llama numerics (rotate_half RoPE, fp16 q/k/v, fp32 softmax).
// The token position is read from device memory so a captured hipGraph can be
// replayed for every token.
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// ------------------------------------------------------------------ RMSNorm
// one workgroup (256 threads) per row; y = gamma * fp16(x * rsqrt(mean(x^2) + eps)).  FRAG: the row is written in
// fragment order (amq_hip.h: amq_xfrag_f16) for the few-row GEMMs that follow -- one workgroup per row of the padded
// 64-row groups, rows >= M written as zeros.
template <bool FRAG>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const _Float16* x, const _Float16* gamma, _Float16* y, int M, int K, float eps) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const bool live = m < M;
    const _Float16* xr = x + (size_t)(live ? m : 0) * K;
    float ss = 0.f;
    for (int c = threadIdx.x; c < (K >> 3); c += 256) {
        h8 v = *(const h8*)(xr + 8 * c);
#pragma unroll
        for (int i = 0; i < 8; ++i) { float f = (float)v[i]; ss += f * f; }
    }
    ss = wave_sum_f(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
    const int G = K >> 7, gy = m >> 6, mb = (m & 63) >> 4, r = m & 15;
    for (int c = threadIdx.x; c < (K >> 3); c += 256) {
        h8 v = *(const h8*)(xr + 8 * c);
        h8 g = *(const h8*)(gamma + 8 * c);
        h8 o8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { _Float16 n = (_Float16)((float)v[i] * rstd); o8[i] = g[i] * n; }
        if (FRAG) {
            if (!live) o8 = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            const int kt = c >> 4, t = (c >> 2) & 3, o = c & 3;     // k = 8c = kt*128 + 32t + 8o
            *(h8*)(y + ((((size_t)gy * G + kt) * 16 + mb * 4 + t) * 64 + o * 16 + r) * 8) = o8;
        } else {
            *(h8*)(y + (size_t)m * K + 8 * c) = o8;
        }
    }
}

hipError_t launch_rmsnorm(const void* x, const void* gamma, void* y, int M, int K, float eps, hipStream_t st) {
    hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3(M), dim3(256), 0, st, (const _Float16*)x, (const _Float16*)gamma, (_Float16*)y, M, K, eps);
    return hipGetLastError();
}

hipError_t launch_rmsnorm_xfrag(const void* x, const void* gamma, void* xf, int M, int K, float eps, hipStream_t st) {
    hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3(((M + 63) / 64) * 64), dim3(256), 0, st, (const _Float16*)x,
                       (const _Float16*)gamma, (_Float16*)xf, M, K, eps);
    return hipGetLastError();
}

// ------------------------------------------------------- fp16-weight GEMV
// M = 1 .. 8 rows of x (batched decode: the lm_head is streamed ONCE for all sequences).  x (optionally RMSNorm'ed) staged in
// LDS; each wave owns rows r = first + i * stride of W and streams them 16 B per lane (512 k per wave-load), four loads
// in flight; 6-step wavefront reduction per (row of W, row of x).  M is a template parameter: the M = 1 instantiation
// is the decode step's lm_head kernel unchanged.
constexpr int F16W_WAVES = 4;

// WV: waves per workgroup.  One row: 4 (many small workgroups per CU).  Several rows: the staged x (M * K halves) is what limits the
// workgroups per CU, so 16 waves share one stage (8 rows of K = 4096: 2 x 16 waves per CU instead of 2 x 4: 96 -> us per launch below).
template <bool NORM, int M, int WV>
__global__ __launch_bounds__(WV * 64) void gemv_f16w_kernel(const _Float16* x, const _Float16* W, const _Float16* bias,
                                                                     _Float16* y, const _Float16* gamma, float eps, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* xl = (_Float16*)smem;                 // [M][K]
    float* red = (float*)(smem + (size_t)M * K * 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunks = K >> 3;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const _Float16* xm = x + (size_t)m * K;
        _Float16* xlm = xl + (size_t)m * K;
        if (!NORM) {
            for (int c = tid; c < chunks; c += WV * 64) *(h8*)(xlm + 8 * c) = *(const h8*)(xm + 8 * c);
        } else {
            float ss = 0.f;
            for (int c = tid; c < chunks; c += WV * 64) {
                h8 v = *(const h8*)(xm + 8 * c);
                *(h8*)(xlm + 8 * c) = v;
#pragma unroll
                for (int i = 0; i < 8; ++i) { float f = (float)v[i]; ss += f * f; }
            }
            ss = wave_sum_f(ss);
            if (M > 1) __syncthreads();             // the previous row's readers of red[] are done
            if (lane == 0) red[wave] = ss;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WV; ++w) tot += red[w];
            const float rstd = rsqrtf(tot / (float)K + eps);
            for (int c = tid; c < chunks; c += WV * 64) {
                h8 v = *(h8*)(xlm + 8 * c);
                h8 g = *(const h8*)(gamma + 8 * c);
                h8 r;
#pragma unroll
                for (int i = 0; i < 8; ++i) { _Float16 n = (_Float16)((float)v[i] * rstd); r[i] = g[i] * n; }
                *(h8*)(xlm + 8 * c) = r;
            }
        }
    }
    __syncthreads();
    const int gw = blockIdx.x * WV + wave, nw = gridDim.x * WV;
    const int steps = (K + 511) >> 9;               // 512 k per wave-load; lanes past K are masked (K % 8 == 0)
    for (int row = gw; row < N; row += nw) {
        const _Float16* wr = W + (size_t)row * K + 8 * lane;
        float acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = 0.f;
        for (int s = 0; s < steps; s += 4) {
            u4 buf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (s + j < steps && (s + j) * 512 + 8 * lane < K) buf[j] = AMQ_STREAM_LOAD((const u4*)(wr + (size_t)(s + j) * 512));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (s + j < steps && (s + j) * 512 + 8 * lane < K) {
                    const uint32_t wv[4] = {buf[j].x, buf[j].y, buf[j].z, buf[j].w};
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const h8 xv = *(const h8*)(xl + (size_t)m * K + (s + j) * 512 + 8 * lane);
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            h2 xp = {xv[2 * p], xv[2 * p + 1]};
                            acc[m] = __builtin_amdgcn_fdot2(as_h2(wv[p]), xp, acc[m], false);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float t = wave_sum_f(acc[m]);
            if (lane == 0) {
                _Float16 o = (_Float16)t;
                if (bias) o = o + bias[row];
                y[(size_t)m * N + row] = o;
            }
        }
    }
}

template <bool NORM, int M>
static hipError_t launch_gemv_f16w_m(const void* x, const void* W, const void* bias, void* y, const void* gamma, float eps,
                                     int N, int K, hipStream_t st) {
    const size_t lds = (size_t)M * K * 2 + 64;
    constexpr int WV = M == 1 ? F16W_WAVES : 16;
    auto k = gemv_f16w_kernel<NORM, M, WV>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(M == 1 ? 1024 : 512), dim3(WV * 64), lds, st, (const _Float16*)x, (const _Float16*)W, (const _Float16*)bias,
                       (_Float16*)y, (const _Float16*)gamma, eps, N, K);
    return hipGetLastError();
}

template <bool NORM>
static hipError_t launch_gemv_f16w_n(const void* x, const void* W, const void* bias, void* y, const void* gamma, float eps,
                                     int M, int N, int K, hipStream_t st) {
    switch (M) {
        case 1: return launch_gemv_f16w_m<NORM, 1>(x, W, bias, y, gamma, eps, N, K, st);
        case 2: return launch_gemv_f16w_m<NORM, 2>(x, W, bias, y, gamma, eps, N, K, st);
        case 3: return launch_gemv_f16w_m<NORM, 3>(x, W, bias, y, gamma, eps, N, K, st);
        case 4: return launch_gemv_f16w_m<NORM, 4>(x, W, bias, y, gamma, eps, N, K, st);
        case 5: return launch_gemv_f16w_m<NORM, 5>(x, W, bias, y, gamma, eps, N, K, st);
        case 6: return launch_gemv_f16w_m<NORM, 6>(x, W, bias, y, gamma, eps, N, K, st);
        case 7: return launch_gemv_f16w_m<NORM, 7>(x, W, bias, y, gamma, eps, N, K, st);
        default: return launch_gemv_f16w_m<NORM, 8>(x, W, bias, y, gamma, eps, N, K, st);
    }
}

// x: fp16 [M, K] contiguous, y: fp16 [M, N] contiguous, 1 <= M <= 8
hipError_t launch_gemv_f16w(const void* x, const void* W, const void* bias, void* y, const void* gamma, float eps,
                            int N, int K, hipStream_t st, int M) {
    StreamDevice sd_(st);                                  // kernel attributes are per device: the stream's, not the current one
    if (gamma) return launch_gemv_f16w_n<true>(x, W, bias, y, gamma, eps, M, N, K, st);
    return launch_gemv_f16w_n<false>(x, W, bias, y, nullptr, eps, M, N, K, st);
}

// ------------------------------------------------ end of a token step
// greedy next token + what the next step needs, in ONE single-workgroup launch (replaces argmax, position increment and
// embedding gather = three framework kernels and their boundaries): token = argmax(logits) (first maximum, like
// torch.argmax), pos += 1, x = embed[token].
// Batched decode: one workgroup per sequence (logits / token / x rows of blockIdx.x); the shared position and the cos/sin row are
// advanced by workgroup 0 only (nobody else reads them here).
// suppress: up to 8 token ids (device int32, -1 = unused slot) that are never chosen -- what HF's MinNewTokensLengthLogitsProcessor does to the EOS
// ids while min_new_tokens has not been reached (generate(min_new_tokens = max_new_tokens = n): speed.py:31-36): their logits count as -inf.
// Checked only where a thread finds a new maximum (a handful of times per thread), not per element.
constexpr int TAIL_MAX_SUPPRESS = 8;
__global__ __launch_bounds__(1024) void decode_tail_kernel(const _Float16* logits, int vocab, const _Float16* embed, int hidden,
                                                            long long* token, int* pos, _Float16* x, const _Float16* rope_table,
                                                            _Float16* rope_cur, int rope_rows, const int* suppress) {
    __shared__ float smax[16];
    __shared__ int sidx[16];
    __shared__ int stok, spos;
    const int tid = threadIdx.x;
    logits += (size_t)blockIdx.x * vocab;
    token += blockIdx.x;
    x += (size_t)blockIdx.x * hidden;
    if (blockIdx.x != 0) rope_cur = nullptr;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    int sup[TAIL_MAX_SUPPRESS];
#pragma unroll
    for (int j = 0; j < TAIL_MAX_SUPPRESS; ++j) sup[j] = suppress ? suppress[j] : -1;
    auto allowed = [&](int idx) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < TAIL_MAX_SUPPRESS; ++j) ok = ok && idx != sup[j];
        return ok;
    };
    const int chunks = vocab >> 3;
    for (int c = tid; c < chunks; c += 1024) {
        const h8 v = *(const h8*)(logits + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            if (f > best && (!suppress || allowed(8 * c + e))) { best = f; bi = 8 * c + e; }   // ascending index inside a thread: first maximum kept
        }
    }
    for (int i = 8 * chunks + tid; i < vocab; i += 1024) {       // vocab % 8 tail
        const float f = (float)logits[i];
        if ((f > best || (f == best && i < bi)) && (!suppress || allowed(i))) { best = f; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((tid & 63) == 0) { smax[tid >> 6] = best; sidx[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        float b = smax[0];
        int ix = sidx[0];
        for (int w = 1; w < 16; ++w)
            if (smax[w] > b || (smax[w] == b && sidx[w] < ix)) { b = smax[w]; ix = sidx[w]; }
        if (ix == 0x7fffffff) ix = 0;                            // all NaN / empty: torch returns 0-ish; keep it in range
        stok = ix;
        token[0] = (long long)ix;
        if (blockIdx.x == 0) {
            spos = pos[0] + 1;
            if (rope_table && spos > rope_rows) spos = rope_rows;    // saturate at the end of the cache: the attention kernel
            pos[0] = spos;                                           // treats pos == max_seq as "out of range" (no-op + error word)
        }
    }
    __syncthreads();
    if (rope_cur && tid < 128) {                                      // cos/sin row of the new position (last row once the cache is full)
        const int rr = spos < rope_rows ? spos : rope_rows - 1;
        rope_cur[tid] = rope_table[(size_t)rr * 128 + tid];
    }
    const _Float16* row = embed + (size_t)stok * hidden;
    for (int c = tid; c < (hidden >> 3); c += 1024) *(h8*)(x + 8 * c) = *(const h8*)(row + 8 * c);
}

hipError_t launch_decode_tail(const void* logits, int vocab, const void* embed, int hidden, void* token, void* pos, void* x,
                              const void* rope_table, void* rope_cur, int rope_rows, hipStream_t st, int batch, const void* suppress) {
    hipLaunchKernelGGL(decode_tail_kernel, dim3(batch), dim3(1024), 0, st, (const _Float16*)logits, vocab, (const _Float16*)embed, hidden,
                       (long long*)token, (int*)pos, (_Float16*)x, (const _Float16*)rope_table, (_Float16*)rope_cur, rope_rows, (const int*)suppress);
    return hipGetLastError();
}

// The step-state side of "this is the next input token" in one launch (what a caller that feeds tokens itself -- model(ids, start_pos=...) token by
// token, speed.py:76-90 -- otherwise does with an index copy, an embedding gather and a table-row gather): token[b] = token_in[b] (one id broadcast
// when n_in == 1), x[b] = embed[token[b]], rope_cur = rope_table[min(pos, rope_rows - 1)].  The position itself is not touched.
__global__ __launch_bounds__(256) void set_token_kernel(const long long* token_in, int n_in, const _Float16* embed, int vocab, int hidden, long long* token,
                                                         const int* pos, _Float16* x, const _Float16* rope_table, _Float16* rope_cur, int rope_rows) {
    const int b = blockIdx.x, tid = threadIdx.x;
    long long t = token_in[n_in == 1 ? 0 : b];
    t = t < 0 ? 0 : t >= vocab ? vocab - 1 : t;                      // (an id outside the vocabulary must not become an out-of-bounds gather)
    if (tid == 0) token[b] = t;
    if (b == 0 && rope_cur && tid < 128) {
        int p = pos[0];
        p = p < 0 ? 0 : p < rope_rows ? p : rope_rows - 1;
        rope_cur[tid] = rope_table[(size_t)p * 128 + tid];
    }
    const _Float16* row = embed + (size_t)t * hidden;
    _Float16* xr = x + (size_t)b * hidden;
    for (int c = tid; c < (hidden >> 3); c += 256) *(h8*)(xr + 8 * c) = *(const h8*)(row + 8 * c);
}

hipError_t launch_set_token(const void* token_in, int n_in, const void* embed, int vocab, int hidden, void* token, const void* pos, void* x,
                            const void* rope_table, void* rope_cur, int rope_rows, int batch, hipStream_t st) {
    hipLaunchKernelGGL(set_token_kernel, dim3(batch), dim3(256), 0, st, (const long long*)token_in, n_in, (const _Float16*)embed, vocab, hidden,
                       (long long*)token, (const int*)pos, (_Float16*)x, (const _Float16*)rope_table, (_Float16*)rope_cur, rope_rows);
    return hipGetLastError();
}

// ------------------------------------------ RoPE + KV append + attention
// grid = (n_heads, batch); 512 threads; head_dim == 128.
// KV cache layout: [batch][kv_head][max_seq][128] fp16 (keys already rotated).
// HF Llama numerics: cos/sin computed in fp32, cast to fp16; q' = q*cos + rotate_half(q)*sin in fp16;
// scores and softmax in fp32; probabilities cast to fp16 before P.V (eager attention path).
constexpr int ATT_D = 128;

// cos/sin of (pos * theta^(-2i/128)), i = 0..63, as fp16 -- the values HF's rotary embedding feeds
// apply_rotary_pos_emb.  Computed by lanes 0..63 (accurate sincosf; one call per lane).
__device__ __forceinline__ void rope_cs(float theta, int pos, int i, _Float16* c16, _Float16* s16) {
    const float inv_freq = 1.0f / powf(theta, (float)(2 * i) / (float)ATT_D);   // LlamaRotaryEmbedding: 1 / base^(2i/d), fp32
    float sn, cs;
    sincosf((float)pos * inv_freq, &sn, &cs);
    *c16 = (_Float16)cs;
    *s16 = (_Float16)sn;
}

// 16-lane (DPP row) all-reduce: after it every lane of a row holds the row's sum
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));
    return v;
}
// whole-wave reductions without LDS-crossbar shuffles (6 dependent ds_bpermute round trips cost ~0.35 us each):
// DPP inside the four 16-lane rows, then four v_readlane
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = row16_sum(v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48))));
}

// One workgroup of 512 threads per (head, sequence) = 32 groups of 16 lanes; a group owns the cached keys / values
// t = group + 32 i and a lane the 16-byte slice [8 lane, 8 lane + 8) of each 256-byte row, so every wave-load is four
// whole contiguous rows (the first version gave each thread one key row: 64 different cache lines per load
// instruction).  A single CU pulls its head's K/V rows at what it keeps in flight, so ALL rows of contexts up to
// 32 * ATT_PF = 384 tokens are requested before anything is computed (the first 32 * ATT_SPEC keys even before the position
// has arrived); longer contexts continue with a plain loop.  The leading arguments are kernarg-preloaded (Makefile).
constexpr int ATT_THREADS = 512;
constexpr int ATT_GROUPS = ATT_THREADS / 16;
constexpr int ATT_PF = 12;             // rows of K and of V per group held in registers
// of those, K rows requested before the position is known.  Eight (256 keys) until round 5: at the benchmark's 64 .. 330 cached keys most of them were rows the
// context does not have yet -- real reads all the same, 17 MB per launch when 8 sequences decode together.  Two: 8 sequences 1.683 -> 1.664 ms per step,
// 4 sequences 1.505 -> 1.485, one sequence 832 -> 834 - 837 tokens/s (inside the run-to-run spread); 0 / 1 / 3 / 4 measured beside it
// (profiles/r05_attn_decode_long.txt, last section)
#ifndef AMQ_ATT_SPEC
#define AMQ_ATT_SPEC 2
#endif
constexpr int ATT_SPEC = AMQ_ATT_SPEC;

struct AttnRest {
    void* out; const void* rope_table; int pos; float rope_theta;
#ifdef AMQ_STAMP
    unsigned long long* stamps;     // diagnostic build: [heads][16] s_memrealtime phase stamps
#endif
};
#ifdef AMQ_STAMP
extern unsigned long long* g_stamp_ptr;
#define ATT_STAMP(slot_) do { if (rest.stamps && threadIdx.x == 0) rest.stamps[(size_t)blockIdx.x * 16 + (slot_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ATT_STAMP(slot_) do { } while (0)
#endif

// p_state: mode 0 -> device int32 position (or null: rest.pos); mode 1 ("cur") -> the step-state block
// {fp16 cos/sin [64][2] of the current position; int32 position at byte 256} that amq_decode_tail_f16 maintains: the
// rotation inputs AND the position are then fetched by the first instructions of the kernel, with no dependent load.
__global__ __launch_bounds__(ATT_THREADS) void attn_decode_kernel(void* p_kc, void* p_vc, const void* p_state, int p_heads,
                                                                   int p_max_seq, const void* p_q, const void* p_k,
                                                                   const void* p_v, AttnRest rest) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* qs = (_Float16*)smem;                 // [128] rotated q
    _Float16* ks = (_Float16*)smem + ATT_D;         // [128] rotated new key (also what is appended)
    _Float16* vs = (_Float16*)smem + 2 * ATT_D;     // [128] new value
    float* sc = (float*)(smem + 6 * ATT_D);         // [T] scores / probabilities
    __shared__ float red[2 * ATT_THREADS / 64];
    __shared__ float part[ATT_GROUPS][ATT_D];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int n_heads = p_heads & 0xFF, n_kv_heads = (p_heads >> 8) & 0xFF, max_seq = p_max_seq;
    const bool cur_mode = (p_heads >> 16) & 1;
    const void* p_cur = cur_mode ? p_state : nullptr;
    // the position: a scalar load issued by the kernel's first instructions, first needed after the speculative rows.
    // (`rest` is only touched on the paths that need it: a select that also reads rest.pos would put an lgkmcnt(0) wait --
    // kernarg block AND position -- in front of every vector load)
    int pos;
    if (cur_mode) pos = *(const int*)((const char*)p_state + 256);
    else if (p_state) pos = *(const int*)p_state;
    else pos = rest.pos;
    const int group = n_heads / n_kv_heads;
    const int kvh = h / group;
    const int grp = tid >> 4, l16 = tid & 15;
    const _Float16* q = (const _Float16*)p_q + ((size_t)b * n_heads + h) * ATT_D;
    const _Float16* kn = (const _Float16*)p_k + ((size_t)b * n_kv_heads + kvh) * ATT_D;
    const _Float16* vn = (const _Float16*)p_v + ((size_t)b * n_kv_heads + kvh) * ATT_D;
    _Float16* kc = (_Float16*)p_kc + ((size_t)b * n_kv_heads + kvh) * (size_t)max_seq * ATT_D;
    _Float16* vc = (_Float16*)p_vc + ((size_t)b * n_kv_heads + kvh) * (size_t)max_seq * ATT_D;

    ATT_STAMP(0);
    // vmcnt waits are in issue order: what the rotation needs (raw q / k / v of the new token, later the cos/sin pair) is
    // requested BEFORE the wave's share of the cache rows, or wave 0 would sit behind its 24 row loads (measured 2.5-3.4 us
    // from position to barrier, profiles/r01b_attn_stamps.txt)
    // p_cur (kernarg-preloaded): cos/sin of the CURRENT position, kept up to date by the step's tail kernel
    // (amq_decode_tail_f16) -- the rotation then no longer waits for the position -> table-row chain
    _Float16 q0 = 0, q1 = 0, k0 = 0, k1 = 0, v0 = 0, v1 = 0;
    // (two registers for the two possible sources of cos/sin: one variable with two defining loads makes the compiler wait
    // for the LATER one -- issued behind the speculative rows -- on both paths)
    h2 cs_cur = {(_Float16)1.f, (_Float16)0.f}, cs_tab = {(_Float16)1.f, (_Float16)0.f};
    if (tid < 64) {
        q0 = q[tid]; q1 = q[tid + 64];
        k0 = kn[tid]; k1 = kn[tid + 64];
        if (p_cur) cs_cur = ((const h2*)p_cur)[tid];
        v0 = vn[tid]; v1 = vn[tid + 64];
    }
    // speculative: the first ATT_SPEC * 32 key rows (clamped to the cache; rows >= pos are discarded later)
    h8 krow[ATT_PF], vrow[ATT_PF];
#pragma unroll
    for (int i = 0; i < ATT_SPEC; ++i) {
        int t = grp + ATT_GROUPS * i;
        t = t < max_seq - 1 ? t : max_seq - 1;
        krow[i] = *(const h8*)(kc + (size_t)t * ATT_D + 8 * l16);
    }
    // A position outside the cache (a graph replayed past max_seq, a corrupted step state) must not index the cache or
    // the LDS score array: the whole workgroup leaves (pos is wave-uniform), nothing is appended or written, and in
    // step-state mode the sticky error word at byte 260 of the block is raised for the host to read.
    if (pos < 0 || pos >= max_seq) {
        if (cur_mode && tid == 0) *(int*)((char*)const_cast<void*>(p_state) + 260) = 1;
        return;
    }
    const int T = pos + 1;
    ATT_STAMP(1);
    if (tid < 64 && !p_cur && rest.rope_table) cs_tab = ((const h2*)rest.rope_table)[(size_t)pos * 64 + tid];
    const int last_old = pos > 0 ? pos - 1 : 0;      // rows >= pos are never used; clamp keeps every load inside rows already written
    // rows past the context are not requested at all (wave-uniform skip): every wave-load costs the CU's texture
    // addresser >= 16 cycles whether or not its lanes point at the same clamped row, and at T ~ 200 half of the 24 loads
    // per wave were such duplicates (2.2 us from position to barrier, profiles/r01b_attn_stamps.txt)
#pragma unroll
    for (int i = ATT_SPEC; i < ATT_PF; ++i) {
        if (ATT_GROUPS * i < pos) {
            int t = grp + ATT_GROUPS * i;
            t = t < last_old ? t : last_old;
            krow[i] = *(const h8*)(kc + (size_t)t * ATT_D + 8 * l16);
        }
    }
    auto load_v = [&](int i) {
        if (ATT_GROUPS * i < pos) {
            int t = grp + ATT_GROUPS * i;
            t = t < last_old ? t : last_old;
            vrow[i] = *(const h8*)(vc + (size_t)t * ATT_D + 8 * l16);
        }
    };
    // (V's rows are requested as the scores are taken, not here behind K's: with a workgroup per CU -- 8 sequences x 32 heads -- everything at once costs
    //  more than it hides: 8 sequences 1.668 -> 1.640 ms per step, 4 sequences 1.480 -> 1.465, one sequence unchanged; profiles/r05_attn_decode_long.txt)
    auto rotate_and_append = [&](_Float16 c16, _Float16 s16) {
        const int i = tid;                          // rotary pair (i, i + 64)
        qs[i] = q0 * c16 + (-q1) * s16;             // q*cos + rotate_half(q)*sin  (fp16 ops, HF apply_rotary_pos_emb)
        qs[i + 64] = q1 * c16 + q0 * s16;
        const _Float16 r0 = k0 * c16 + (-k1) * s16, r1 = k1 * c16 + k0 * s16;
        ks[i] = r0;
        ks[i + 64] = r1;
        vs[i] = v0;
        vs[i + 64] = v1;
        if (h % group == 0) {                       // one query head per kv group appends to the cache
            kc[(size_t)pos * ATT_D + i] = r0;
            kc[(size_t)pos * ATT_D + i + 64] = r1;
            vc[(size_t)pos * ATT_D + i] = v0;
            vc[(size_t)pos * ATT_D + i + 64] = v1;
        }
    };
    if (tid < 64) {
        if (p_cur) {
            rotate_and_append(cs_cur.x, cs_cur.y);
        } else {
            _Float16 c16 = cs_tab.x, s16 = cs_tab.y;
            if (!rest.rope_table) rope_cs(rest.rope_theta, pos, tid, &c16, &s16);
            rotate_and_append(c16, s16);
        }
    }
    __syncthreads();
    ATT_STAMP(2);

    // scores: 16 lanes x 8 dims per key, DPP row reduction; the new key comes from LDS (its cache row may still be in flight)
    const float scale = rsqrtf((float)ATT_D);
    const h8 qv = *(const h8*)(qs + 8 * l16);
    const h8 knew = *(const h8*)(ks + 8 * l16);
    auto score = [&](const h8& kv) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            s = __builtin_amdgcn_fdot2((h2){qv[2 * e], qv[2 * e + 1]}, (h2){kv[2 * e], kv[2 * e + 1]}, s, false);
        s = row16_sum(s);
        // HF eager attention: matmul(q, k^T) -> fp16, * scaling -> fp16, softmax in fp32
        return (float)(_Float16)((float)(_Float16)s * scale);
    };
#pragma unroll
    for (int i = 0; i < ATT_PF; ++i) {
        load_v(i);                                   // V's row i leaves as score i is taken (the split kernel's order at one workgroup per CU)
        if (ATT_GROUPS * i < T) {                   // wave-uniform: iterations wholly past the context never touch their row
            const int t = grp + ATT_GROUPS * i;
            const float sv = score(t == pos ? knew : krow[i]);
            if (t < T && l16 == 0) sc[t] = sv;
        }
    }
    for (int t = grp + ATT_GROUPS * ATT_PF; t < T; t += ATT_GROUPS) {       // contexts beyond the register prefetch
        const h8 kv = (t == pos) ? knew : *(const h8*)(kc + (size_t)t * ATT_D + 8 * l16);
        const float sv = score(kv);
        if (l16 == 0) sc[t] = sv;
    }
    __syncthreads();
    ATT_STAMP(3);
    float lmax = -INFINITY;
    for (int t = tid; t < T; t += ATT_THREADS) lmax = fmaxf(lmax, sc[t]);
    lmax = wave_max_dpp(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    float gmax = red[0];
#pragma unroll
    for (int w = 1; w < ATT_THREADS / 64; ++w) gmax = fmaxf(gmax, red[w]);
    float lsum = 0.f;
    for (int t = tid; t < T; t += ATT_THREADS) {
        const float e = __expf(sc[t] - gmax);
        sc[t] = e;
        lsum += e;
    }
    lsum = wave_sum_dpp(lsum);
    if ((tid & 63) == 0) red[ATT_THREADS / 64 + (tid >> 6)] = lsum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_THREADS / 64; ++w) tot += red[ATT_THREADS / 64 + w];
    const float inv = 1.0f / tot;
    ATT_STAMP(4);

    // out = sum_t p_t * V[t]: 32 key groups x 16 lanes, 8 dims (16 B) per lane
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const h8 vnew = *(const h8*)(vs + 8 * l16);
#pragma unroll
    for (int i = 0; i < ATT_PF; ++i) {
        const int t = grp + ATT_GROUPS * i;
        if (ATT_GROUPS * i < T && t < T) {
            const _Float16 p16 = (_Float16)(sc[t] * inv);        // softmax(...).to(fp16)
            const h8 vv = (t == pos) ? vnew : vrow[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += (float)p16 * (float)vv[e];
        }
    }
    for (int t = grp + ATT_GROUPS * ATT_PF; t < T; t += ATT_GROUPS) {
        const _Float16 p16 = (_Float16)(sc[t] * inv);
        const h8 vv = (t == pos) ? vnew : *(const h8*)(vc + (size_t)t * ATT_D + 8 * l16);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += (float)p16 * (float)vv[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[grp][8 * l16 + e] = o[e];
    __syncthreads();
    ATT_STAMP(5);
    if (tid < ATT_D) {
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < ATT_GROUPS; ++g) acc += part[g][tid];
        _Float16* out = (_Float16*)rest.out + ((size_t)b * n_heads + h) * ATT_D;
        out[tid] = (_Float16)acc;
    }
    ATT_STAMP(6);
}

// ------------------------------------------ the same step with the cached keys split over several workgroups per head
// attn_decode_kernel walks a head's whole context on ONE CU: fine for the few hundred keys of the reference's default
// benchmark, but a CU pulls its K / V rows at 15-60 GB/s, so at 4000 cached keys a 7B token spent 4.2 of its 5.4 ms there
// (profiles/r02_decode_context.txt).  Here grid.z workgroups share a head: the context 0 .. pos is cut into n_act <= grid.z
// chunks of >= 256 keys (a multiple of 32; at most 384 when grid.z = ceil(max_seq / 384): the register-prefetch path), the
// workgroup of chunk z computes its scores, its own softmax statistics (m_z, l_z) and its un-normalised fp32 output O_z;
// the chunk holding the new token rotates / appends it.  With one active chunk the arithmetic and the result are those of
// attn_decode_kernel bit for bit (same expressions, probabilities rounded to fp16 after normalisation).  With several, every
// workgroup leaves (O_z, m_z, l_z) in the workspace as agent-scope stores, takes a ticket, and the LAST one to arrive -- no
// workgroup waits -- combines them in chunk order: out = sum_z O_z e^(m_z - M) / sum_z l_z e^(m_z - M).  That is the exact
// softmax with UNROUNDED probabilities (HF's eager path rounds them to fp16 first): within fp16 output rounding of the
// one-chunk result, tests bound it.  Tickets are zero before and after every launch.
struct AttnSplit { float* ws; int* tickets; int n_splits; };
constexpr int ATT_WS_STRIDE = ATT_D + 4;       // floats per (head, chunk): O[128], m, l, pad

// cached K / V rows are read once per token: non-temporal, so that 2 x 33 MB per layer at 2048 keys do not displace what the step re-reads
#ifndef AMQ_ATT_NT
#define AMQ_ATT_NT 1
#endif
#if AMQ_ATT_NT
#define ATT_KV_LOAD(p) __builtin_nontemporal_load((const h8*)(p))
#else
#define ATT_KV_LOAD(p) (*(const h8*)(p))
#endif
template <int RING>
__global__ __launch_bounds__(ATT_THREADS) void attn_decode_split_kernel(void* p_kc, void* p_vc, const void* p_state, int p_heads,
                                                                         int p_max_seq, const void* p_q, const void* p_k,
                                                                         const void* p_v, AttnRest rest, AttnSplit sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* qs = (_Float16*)smem;                 // [128] rotated q
    _Float16* ks = (_Float16*)smem + ATT_D;         // [128] rotated new key (also what is appended)
    _Float16* vs = (_Float16*)smem + 2 * ATT_D;     // [128] new value
    float* sc = (float*)(smem + 6 * ATT_D);         // [chunk] scores / probabilities of this workgroup's keys
    __shared__ float red[2 * ATT_THREADS / 64];
    __shared__ float part[ATT_GROUPS][ATT_D];
    __shared__ int last_flag;
    const int b = blockIdx.y, z = blockIdx.z, tid = threadIdx.x;
    const int n_heads = p_heads & 0xFF, n_kv_heads = (p_heads >> 8) & 0xFF, max_seq = p_max_seq;
    // Grouped-query models: the query heads of one kv head read the same K / V chunk.  Workgroups are dealt round-robin over
    // the 8 XCDs (linear id % 8 = blockIdx.x % 8 when the head count is a multiple of 8), so the heads of a kv group are given
    // ids with the same residue: the chunk is then fetched into ONE XCD's L2 instead of eight.  (kv heads % 8 != 0: identity.)
    int h = blockIdx.x;
    if (n_kv_heads != n_heads && (n_kv_heads & 7) == 0) {
        const int G = n_heads / n_kv_heads, x = h & 7, j = h >> 3;
        h = (x + 8 * (j / G)) * G + j % G;
    }
    const bool cur_mode = (p_heads >> 16) & 1;
    const void* p_cur = cur_mode ? p_state : nullptr;
    int pos;
    if (cur_mode) pos = *(const int*)((const char*)p_state + 256);
    else if (p_state) pos = *(const int*)p_state;
    else pos = rest.pos;
    const int group = n_heads / n_kv_heads;
    const int kvh = h / group;
    const int grp = tid >> 4, l16 = tid & 15;
    const _Float16* q = (const _Float16*)p_q + ((size_t)b * n_heads + h) * ATT_D;
    const _Float16* kn = (const _Float16*)p_k + ((size_t)b * n_kv_heads + kvh) * ATT_D;
    const _Float16* vn = (const _Float16*)p_v + ((size_t)b * n_kv_heads + kvh) * ATT_D;
    _Float16* kc = (_Float16*)p_kc + ((size_t)b * n_kv_heads + kvh) * (size_t)max_seq * ATT_D;
    _Float16* vc = (_Float16*)p_vc + ((size_t)b * n_kv_heads + kvh) * (size_t)max_seq * ATT_D;

    _Float16 q0 = 0, q1 = 0, k0 = 0, k1 = 0, v0 = 0, v1 = 0;
    h2 cs_cur = {(_Float16)1.f, (_Float16)0.f}, cs_tab = {(_Float16)1.f, (_Float16)0.f};
    if (tid < 64) {
        q0 = q[tid]; q1 = q[tid + 64];
        k0 = kn[tid]; k1 = kn[tid + 64];
        if (p_cur) cs_cur = ((const h2*)p_cur)[tid];
        v0 = vn[tid]; v1 = vn[tid + 64];
    }
    // a position outside the cache: nothing is appended or written (attn_decode_kernel's guard)
    if (pos < 0 || pos >= max_seq) {
        if (cur_mode && tid == 0 && z == 0) *(int*)((char*)const_cast<void*>(p_state) + 260) = 1;
        return;
    }
    const int T = pos + 1;
    int chunk = (((T + sp.n_splits - 1) / sp.n_splits) + 31) & ~31;
    chunk = chunk < ATT_MIN_CHUNK ? ATT_MIN_CHUNK : chunk;
    const int n_act = (T + chunk - 1) / chunk;       // chunks that hold keys: workgroups z >= n_act have nothing to do
    if (z >= n_act) return;
    const int t0 = z * chunk;
    const int t1 = t0 + chunk < T ? t0 + chunk : T;  // this workgroup's keys: t0 .. t1 - 1
    const int Tl = t1 - t0;
    const bool has_new = t1 == T;                    // the chunk that holds the new token (the last active one)
    if (tid < 64 && !p_cur && rest.rope_table) cs_tab = ((const h2*)rest.rope_table)[(size_t)pos * 64 + tid];
    const int last_old = pos > 0 ? pos - 1 : 0;      // rows >= pos are never read from the cache
    // The chunk's K and V rows travel through a RING of row loads per thread: RING rows of K leave here, a row's successor when the row is about to be
    // used, V's first rows behind K's last, the rest of V as the output accumulates.  All 2 x 9 rows of a thread at once (the first form) put 144 KB per
    // workgroup in flight -- 37 MB over the chip at 2048 keys, more than the launch reads -- and the rows then arrive at 3.9 TB/s at the margin; the memory
    // system takes less in flight better: RING = ATT_PF (all of K at once, V trailing the scores) for launches of at most one workgroup per CU, 4 beyond
    // (profiles/r05_attn_decode_long.txt: 13.7 -> 11.9 us at 2048 keys, 21.9 -> 18.1 at 4000).  Same values in the same order of arithmetic: results
    // unchanged bit for bit.
    h8 krow[ATT_PF], vrow[ATT_PF];
    auto load_k = [&](int i) {
        if (ATT_GROUPS * i < Tl) {
            int t = t0 + grp + ATT_GROUPS * i;
            t = t < last_old ? t : last_old;
            krow[i] = ATT_KV_LOAD(kc + (size_t)t * ATT_D + 8 * l16);
        }
    };
    auto load_v = [&](int i) {
        if (ATT_GROUPS * i < Tl) {
            int t = t0 + grp + ATT_GROUPS * i;
            t = t < last_old ? t : last_old;
            vrow[i] = ATT_KV_LOAD(vc + (size_t)t * ATT_D + 8 * l16);
        }
    };
    static_assert(RING >= 1 && RING <= ATT_PF, "ring depth");
#pragma unroll
    for (int i = 0; i < RING; ++i) load_k(i);
    if (tid < 64) {
        _Float16 c16, s16;
        if (p_cur) { c16 = cs_cur.x; s16 = cs_cur.y; }
        else {
            c16 = cs_tab.x; s16 = cs_tab.y;
            if (!rest.rope_table) rope_cs(rest.rope_theta, pos, tid, &c16, &s16);
        }
        const int i = tid;                          // rotary pair (i, i + 64): attn_decode_kernel's rotate_and_append
        qs[i] = q0 * c16 + (-q1) * s16;
        qs[i + 64] = q1 * c16 + q0 * s16;
        const _Float16 r0 = k0 * c16 + (-k1) * s16, r1 = k1 * c16 + k0 * s16;
        ks[i] = r0;
        ks[i + 64] = r1;
        vs[i] = v0;
        vs[i + 64] = v1;
        if (has_new && h % group == 0) {            // one workgroup per kv head appends to the cache
            kc[(size_t)pos * ATT_D + i] = r0;
            kc[(size_t)pos * ATT_D + i + 64] = r1;
            vc[(size_t)pos * ATT_D + i] = v0;
            vc[(size_t)pos * ATT_D + i + 64] = v1;
        }
    }
    __syncthreads();

    const float scale = rsqrtf((float)ATT_D);
    const h8 qv = *(const h8*)(qs + 8 * l16);
    const h8 knew = *(const h8*)(ks + 8 * l16);
    auto score = [&](const h8& kv) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            s = __builtin_amdgcn_fdot2((h2){qv[2 * e], qv[2 * e + 1]}, (h2){kv[2 * e], kv[2 * e + 1]}, s, false);
        s = row16_sum(s);
        return (float)(_Float16)((float)(_Float16)s * scale);
    };
#pragma unroll
    for (int i = 0; i < ATT_PF; ++i) {
        if (i + RING < ATT_PF) load_k(i + RING);
        else load_v(i + RING - ATT_PF);
        if (ATT_GROUPS * i < Tl) {
            const int t = t0 + grp + ATT_GROUPS * i;
            const float sv = score(t == pos ? knew : krow[i]);
            if (t < t1 && l16 == 0) sc[t - t0] = sv;
        }
    }
    for (int t = t0 + grp + ATT_GROUPS * ATT_PF; t < t1; t += ATT_GROUPS) {       // chunks beyond the register prefetch
        const h8 kv = (t == pos) ? knew : *(const h8*)(kc + (size_t)t * ATT_D + 8 * l16);
        const float sv = score(kv);
        if (l16 == 0) sc[t - t0] = sv;
    }
    __syncthreads();
    float lmax = -INFINITY;
    for (int t = tid; t < Tl; t += ATT_THREADS) lmax = fmaxf(lmax, sc[t]);
    lmax = wave_max_dpp(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    float gmax = red[0];
#pragma unroll
    for (int w = 1; w < ATT_THREADS / 64; ++w) gmax = fmaxf(gmax, red[w]);
    float lsum = 0.f;
    for (int t = tid; t < Tl; t += ATT_THREADS) {
        const float e = __expf(sc[t] - gmax);
        sc[t] = e;
        lsum += e;
    }
    lsum = wave_sum_dpp(lsum);
    if ((tid & 63) == 0) red[ATT_THREADS / 64 + (tid >> 6)] = lsum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_THREADS / 64; ++w) tot += red[ATT_THREADS / 64 + w];
    const bool single = n_act == 1;                 // wave-uniform
    const float inv = 1.0f / tot;

    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const h8 vnew = *(const h8*)(vs + 8 * l16);
    auto weight = [&](int tl) {                     // one chunk: softmax(...).to(fp16) as HF's eager path; several: exp(s - m_z), unrounded
        return single ? (float)(_Float16)(sc[tl] * inv) : sc[tl];
    };
#pragma unroll
    for (int i = 0; i < ATT_PF; ++i) {
        if (i + RING < ATT_PF) load_v(i + RING);
        const int t = t0 + grp + ATT_GROUPS * i;
        if (ATT_GROUPS * i < Tl && t < t1) {
            const float p = weight(t - t0);
            const h8 vv = (t == pos) ? vnew : vrow[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += p * (float)vv[e];
        }
    }
    for (int t = t0 + grp + ATT_GROUPS * ATT_PF; t < t1; t += ATT_GROUPS) {
        const float p = weight(t - t0);
        const h8 vv = (t == pos) ? vnew : *(const h8*)(vc + (size_t)t * ATT_D + 8 * l16);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += p * (float)vv[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[grp][8 * l16 + e] = o[e];
    __syncthreads();
    float acc = 0.f;
    if (tid < ATT_D) {
#pragma unroll
        for (int g = 0; g < ATT_GROUPS; ++g) acc += part[g][tid];
    }
    _Float16* out = (_Float16*)rest.out + ((size_t)b * n_heads + h) * ATT_D;
    if (single) {
        if (tid < ATT_D) out[tid] = (_Float16)acc;
        return;
    }
    // ---- several chunks: publish (O_z, m_z, l_z), take a ticket, the last arriver combines
    float* const wsh = sp.ws + ((size_t)b * n_heads + h) * (size_t)sp.n_splits * ATT_WS_STRIDE;
    if (tid < ATT_D) __hip_atomic_store(wsh + (size_t)z * ATT_WS_STRIDE + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == ATT_D) __hip_atomic_store(wsh + (size_t)z * ATT_WS_STRIDE + ATT_D, gmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == ATT_D + 1) __hip_atomic_store(wsh + (size_t)z * ATT_WS_STRIDE + ATT_D + 1, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Publish protocol (cdna_hip_programming.md Guideline 16, R1): the payload stores above are agent-scope (sc1,
    // write-through); EVERY storing wave (waves 0 .. 2 hold tid 0 .. 129) drains its own vector-memory queue before the
    // workgroup barrier, and only then does ONE lane take the ticket.  A workgroup-scope fence emits no s_waitcnt on
    // gfx950 -- the previous form let wave 0's ticket overtake the stores of waves 1 and 2 -- and the
    // wait is inline asm so that the compiler's wait-count pass cannot drop it.
    AMQ_WAIT_VM("attn.split.partials", 0, "");
    __syncthreads();
    if (tid == 0) {
        int* const ticket = sp.tickets + (size_t)b * n_heads + h;
        const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == n_act - 1;
        if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every other chunk has taken its ticket
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    // The combine's loads are all issued side by side: every (m_c, l_c) pair in ONE round trip into LDS, then the O_c rows eight chunks at a time.
    // One dependent agent-scope load after the other (the first form: a load -> max loop, then a three-load loop, per chunk) cost the last
    // arriver ~2 L2 round trips per chunk -- at 6 - 16 chunks as long as the chunk's own stream (profiles/r05_attn_decode_long.txt).
    // Same expressions in the same chunk order as before: bit-identical results.
    float* const ml = sc;                            // [n_act][2] (the score array is free by now; 2 * n_act <= chunk checked below)
    if (2 * n_act <= chunk) {
        // (the first eight chunks' O rows leave together with the (m, l) pairs: up to eight chunks combine in ONE round trip)
        float ov[8];
        if (tid < ATT_D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cc = j < n_act ? j : n_act - 1;
                ov[j] = __hip_atomic_load(wsh + (size_t)cc * ATT_WS_STRIDE + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        for (int c = tid; c < n_act; c += ATT_THREADS) {
            const float* const wc = wsh + (size_t)c * ATT_WS_STRIDE;
            ml[2 * c] = __hip_atomic_load(wc + ATT_D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ml[2 * c + 1] = __hip_atomic_load(wc + ATT_D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid < ATT_D) {
            float M = -INFINITY;
            for (int c = 0; c < n_act; ++c) M = fmaxf(M, ml[2 * c]);
            float L = 0.f, O = 0.f;
            for (int c0 = 0; c0 < n_act; c0 += 8) {
                if (c0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int cc = c0 + j < n_act ? c0 + j : n_act - 1;
                        ov[j] = __hip_atomic_load(wsh + (size_t)cc * ATT_WS_STRIDE + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (c0 + j < n_act) {           // chunk order: deterministic whatever the arrival order
                        const float f = __expf(ml[2 * (c0 + j)] - M);
                        L += ml[2 * (c0 + j) + 1] * f;
                        O += ov[j] * f;
                    }
                }
            }
            out[tid] = (_Float16)(O / L);
        }
        return;
    }
    if (tid < ATT_D) {
        float M = -INFINITY;
        for (int c = 0; c < n_act; ++c)
            M = fmaxf(M, __hip_atomic_load(wsh + (size_t)c * ATT_WS_STRIDE + ATT_D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float L = 0.f, O = 0.f;
        for (int c = 0; c < n_act; ++c) {           // chunk order: deterministic whatever the arrival order
            const float* const wc = wsh + (size_t)c * ATT_WS_STRIDE;
            const float f = __expf(__hip_atomic_load(wc + ATT_D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - M);
            L += __hip_atomic_load(wc + ATT_D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * f;
            O += __hip_atomic_load(wc + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * f;
        }
        out[tid] = (_Float16)(O / L);
    }
}

// cos/sin table for positions 0..max_seq-1 (HF LlamaRotaryEmbedding values, fp32 math, fp16 storage)
__global__ void rope_table_kernel(_Float16* tab, int max_seq, float theta) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= max_seq * 64) return;
    _Float16 c16, s16;
    rope_cs(theta, idx >> 6, idx & 63, &c16, &s16);
    tab[2 * idx] = c16;
    tab[2 * idx + 1] = s16;
}

hipError_t launch_rope_table(void* tab, int max_seq, float theta, hipStream_t st) {
    hipLaunchKernelGGL(rope_table_kernel, dim3((max_seq * 64 + 255) / 256), dim3(256), 0, st, (_Float16*)tab, max_seq, theta);
    return hipGetLastError();
}

// The same table from explicit inverse frequencies (fp32 [64]: what HF's rotary embedding holds in `inv_freq` after any static rope_scaling --
// Llama-3.1's "llama3" wavelength-dependent factors, "linear", ...) and its attention_scaling: fp16(cosf(pos * inv_freq[i]) * scale), the
// expression of LlamaRotaryEmbedding.forward (fp32 product, fp32 cos / sin, one rounding to fp16).
__global__ void rope_table_freqs_kernel(_Float16* tab, int max_seq, const float* inv_freq, float scale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= max_seq * 64) return;
    float sn, cs;
    sincosf((float)(idx >> 6) * inv_freq[idx & 63], &sn, &cs);
    tab[2 * idx] = (_Float16)(cs * scale);
    tab[2 * idx + 1] = (_Float16)(sn * scale);
}

hipError_t launch_rope_table_freqs(void* tab, int max_seq, const void* inv_freq, float scale, hipStream_t st) {
    hipLaunchKernelGGL(rope_table_freqs_kernel, dim3((max_seq * 64 + 255) / 256), dim3(256), 0, st, (_Float16*)tab, max_seq, (const float*)inv_freq, scale);
    return hipGetLastError();
}

// ---- prefill glue ------------------------------------------------------------------------------------
// One 64-thread workgroup per (prompt row s, head): query heads are rotated in place, key heads are rotated INTO the
// cache row pos0 + s, value heads are copied there.  Same fp16 expression and cos/sin table as the decode kernel's
// rotate_and_append, so a prefilled cache row equals the row a decode step would have appended.
// (eight threads per (row, head), 16-byte accesses: see rope_rows_kernel below)
// Several sequences (rows = batch * seq_len, caches [batch][n_kv_heads][max_seq][128]): row s belongs to sequence s / seq_len
// at position pos0 + s % seq_len.
__global__ __launch_bounds__(256) void rope_cache_kernel(_Float16* q, const _Float16* k, const _Float16* v, _Float16* kc,
                                                         _Float16* vc, const h2* tab, int rope_rows, int pos0, int nh, int nkv,
                                                         int max_seq, long units, int seq_len) {
    const long u = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (u >= units) return;
    const int c = threadIdx.x & 7, nhk = nh + nkv;
    const int s = (int)(u / nhk), hx = (int)(u - (long)s * nhk);
    const int bseq = s / seq_len;
    const int pos = pos0 + (s - bseq * seq_len);
    const h2* cs = tab + (size_t)(pos < rope_rows ? pos : rope_rows - 1) * 64 + 8 * c;
    const h8 cs0 = *(const h8*)cs, cs1 = *(const h8*)(cs + 4);
    const _Float16* src = hx < nh ? q + ((size_t)s * nh + hx) * ATT_D : k + ((size_t)s * nkv + (hx - nh)) * ATT_D;
    const h8 a0 = *(const h8*)(src + 8 * c), a1 = *(const h8*)(src + 64 + 8 * c);
    h8 r0, r1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 c16 = e < 4 ? cs0[2 * e] : cs1[2 * (e - 4)], s16 = e < 4 ? cs0[2 * e + 1] : cs1[2 * (e - 4) + 1];
        r0[e] = a0[e] * c16 + (-a1[e]) * s16;
        r1[e] = a1[e] * c16 + a0[e] * s16;
    }
    if (hx < nh) {
        _Float16* row = q + ((size_t)s * nh + hx) * ATT_D;
        *(h8*)(row + 8 * c) = r0;
        *(h8*)(row + 64 + 8 * c) = r1;
    } else {
        const int h = hx - nh;
        const _Float16* vr = v + ((size_t)s * nkv + h) * ATT_D;
        const size_t dst = (((size_t)bseq * nkv + h) * max_seq + pos) * ATT_D;
        *(h8*)(kc + dst + 8 * c) = r0;
        *(h8*)(kc + dst + 64 + 8 * c) = r1;
        *(h8*)(vc + dst + 8 * c) = *(const h8*)(vr + 8 * c);
        *(h8*)(vc + dst + 64 + 8 * c) = *(const h8*)(vr + 64 + 8 * c);
    }
}

hipError_t launch_rope_cache(void* q, const void* k, const void* v, void* kcache, void* vcache, const void* rope_table,
                             int rope_rows, int pos0, int S, int n_heads, int n_kv_heads, int max_seq, hipStream_t st, int batch) {
    const long units = (long)S * batch * (n_heads + n_kv_heads);
    hipLaunchKernelGGL(rope_cache_kernel, dim3((unsigned)((units + 31) / 32)), dim3(256), 0, st, (_Float16*)q, (const _Float16*)k,
                       (const _Float16*)v, (_Float16*)kcache, (_Float16*)vcache, (const h2*)rope_table, rope_rows, pos0, n_heads,
                       n_kv_heads, max_seq, units, S);
    return hipGetLastError();
}

// RoPE in place on q AND k for `rows` rows that are `rows / seq_len` sequences of seq_len positions each (batched prompt
// pass without a cache: the reference harness' GeMM mode at batch > 1): position of row s = pos0 + s % seq_len.
// Eight threads per (row, head): thread c rotates elements [8c, 8c + 8) with their partners [64 + 8c, 64 + 8c + 8) -- 16-byte
// accesses, 32 (row, head) units per 256-thread workgroup (the first version ran one 64-thread workgroup of 2-byte accesses
// per unit: 2.3 TB/s on the 1.3 GB of a 16 x 2048 x 40-head pass).  Same fp16 expression per element.
__global__ __launch_bounds__(256) void rope_rows_kernel(_Float16* q, _Float16* k, const h2* tab, int rope_rows, int pos0, int seq_len,
                                                        int nh, int nkv, long units) {
    const long u = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (u >= units) return;
    const int c = threadIdx.x & 7, nhk = nh + nkv;
    const long s = u / nhk;
    const int hx = (int)(u - s * nhk);
    const int pos = pos0 + (int)(s % seq_len);
    const h2* cs = tab + (size_t)(pos < rope_rows ? pos : rope_rows - 1) * 64 + 8 * c;
    const h8 cs0 = *(const h8*)cs, cs1 = *(const h8*)(cs + 4);                 // (cos, sin) of pairs 8c .. 8c+3 and 8c+4 .. 8c+7
    _Float16* row = hx < nh ? q + ((size_t)s * nh + hx) * ATT_D : k + ((size_t)s * nkv + (hx - nh)) * ATT_D;
    const h8 a0 = *(const h8*)(row + 8 * c), a1 = *(const h8*)(row + 64 + 8 * c);
    h8 r0, r1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 c16 = e < 4 ? cs0[2 * e] : cs1[2 * (e - 4)], s16 = e < 4 ? cs0[2 * e + 1] : cs1[2 * (e - 4) + 1];
        r0[e] = a0[e] * c16 + (-a1[e]) * s16;
        r1[e] = a1[e] * c16 + a0[e] * s16;
    }
    *(h8*)(row + 8 * c) = r0;
    *(h8*)(row + 64 + 8 * c) = r1;
}

hipError_t launch_rope_rows(void* q, void* k, const void* rope_table, int rope_rows, int pos0, int rows, int seq_len, int n_heads,
                            int n_kv_heads, hipStream_t st) {
    const long units = (long)rows * (n_heads + n_kv_heads);
    hipLaunchKernelGGL(rope_rows_kernel, dim3((unsigned)((units + 31) / 32)), dim3(256), 0, st, (_Float16*)q, (_Float16*)k,
                       (const h2*)rope_table, rope_rows, pos0, seq_len, n_heads, n_kv_heads, units);
    return hipGetLastError();
}

// out = fp16(silu(gate)) * up, 8 elements per thread (the GEMV SiLU prologue's expression, for many-row launches)
__global__ __launch_bounds__(256) void silu_mul_kernel(const h8* g, const h8* u, h8* o, long n8) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n8) return;
    const h8 gv = g[idx], uv = u[idx];
    h8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float gf = (float)gv[e];
        const _Float16 sg = (_Float16)(gf / (1.0f + __expf(-gf)));
        r[e] = sg * uv[e];
    }
    o[idx] = r;
}

hipError_t launch_silu_mul(const void* gate, const void* up, void* out, long n, hipStream_t st) {
    const long n8 = n >> 3;
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const h8*)gate, (const h8*)up,
                       (h8*)out, n8);
    return hipGetLastError();
}

// chunk length bound of the split kernel for a cache of max_seq rows cut n_splits ways
static int att_chunk_max(int max_seq, int n_splits) {
    int c = (((max_seq + n_splits - 1) / n_splits) + 31) & ~31;
    return c < ATT_MIN_CHUNK ? ATT_MIN_CHUNK : c;
}

template <int RING>
static hipError_t launch_attn_decode_split_ring(const AttnArgs& a, int batch, int n_splits, void* ws, void* tickets, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_decode_split_kernel<RING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    AttnRest rest{a.out, a.rope_table, a.pos, a.rope_theta};
#ifdef AMQ_STAMP
    rest.stamps = nullptr;
#endif
    AttnSplit sp{(float*)ws, (int*)tickets, n_splits};
    const bool cur = a.rope_cur != nullptr;
    hipLaunchKernelGGL(attn_decode_split_kernel<RING>, dim3(a.n_heads, batch, n_splits), dim3(ATT_THREADS), lds, st, a.kcache, a.vcache,
                       cur ? a.rope_cur : (const void*)a.pos_dev, a.n_heads | (a.n_kv_heads << 8) | ((int)cur << 16), a.max_seq,
                       a.q, a.k, a.v, rest, sp);
    return hipGetLastError();
}

hipError_t launch_attn_decode_split(const AttnArgs& a, int batch, int n_splits, void* ws, void* tickets, hipStream_t st) {
    if (attn_decode_takes_gqa(a.n_heads, a.n_kv_heads, a.max_seq, n_splits)) return launch_attn_decode_gqa(a, batch, n_splits, ws, tickets, st);
    StreamDevice sd_(st);                                  // kernel attributes are per device: the stream's, not the current one
    const size_t lds = 6 * ATT_D + (size_t)att_chunk_max(a.max_seq, n_splits) * 4;
    // row loads in flight per thread (the kernel's RING): while the launch is at most one workgroup per CU all of K at once and V behind it as the scores
    // are taken (RING = ATT_PF), four rows beyond (256 CUs on MI355X); measured 8 / 9 / 12 and 2 / 3 / 4 / 6 / 8: profiles/r05_attn_decode_long.txt
#ifndef AMQ_ATT_RING_WIDE
#define AMQ_ATT_RING_WIDE 12
#endif
#ifndef AMQ_ATT_RING_NARROW
#define AMQ_ATT_RING_NARROW 4
#endif
    if ((long)a.n_heads * batch * n_splits <= 256) return launch_attn_decode_split_ring<AMQ_ATT_RING_WIDE>(a, batch, n_splits, ws, tickets, lds, st);
    return launch_attn_decode_split_ring<AMQ_ATT_RING_NARROW>(a, batch, n_splits, ws, tickets, lds, st);
}

hipError_t launch_attn_decode(const AttnArgs& a, int batch, hipStream_t st) {
    StreamDevice sd_(st);                                  // kernel attributes are per device: the stream's, not the current one
    const size_t lds = 6 * ATT_D + (size_t)a.max_seq * 4;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    AttnRest rest{a.out, a.rope_table, a.pos, a.rope_theta};
#ifdef AMQ_STAMP
    rest.stamps = g_stamp_ptr;
#endif
    const bool cur = a.rope_cur != nullptr;
    hipLaunchKernelGGL(attn_decode_kernel, dim3(a.n_heads, batch), dim3(ATT_THREADS), lds, st, a.kcache, a.vcache,
                       cur ? a.rope_cur : (const void*)a.pos_dev, a.n_heads | (a.n_kv_heads << 8) | ((int)cur << 16), a.max_seq,
                       a.q, a.k, a.v, rest);
    return hipGetLastError();
}

}  // namespace amq
