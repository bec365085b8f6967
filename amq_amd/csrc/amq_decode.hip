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
// one workgroup (256 threads) per row; y = gamma * fp16(x * rsqrt(mean(x^2) + eps))
__global__ __launch_bounds__(256) void rmsnorm_kernel(const _Float16* x, const _Float16* gamma, _Float16* y, int K, float eps) {
    __shared__ float red[4];
    const _Float16* xr = x + (size_t)blockIdx.x * K;
    _Float16* yr = y + (size_t)blockIdx.x * K;
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
    for (int c = threadIdx.x; c < (K >> 3); c += 256) {
        h8 v = *(const h8*)(xr + 8 * c);
        h8 g = *(const h8*)(gamma + 8 * c);
        h8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) { _Float16 n = (_Float16)((float)v[i] * rstd); r[i] = g[i] * n; }
        *(h8*)(yr + 8 * c) = r;
    }
}

hipError_t launch_rmsnorm(const void* x, const void* gamma, void* y, int M, int K, float eps, hipStream_t st) {
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(M), dim3(256), 0, st, (const _Float16*)x, (const _Float16*)gamma, (_Float16*)y, K, eps);
    return hipGetLastError();
}

// ------------------------------------------------------- fp16-weight GEMV
// M == 1.  x (optionally RMSNorm'ed) staged in LDS; each wave owns rows
// r = first + i * stride and streams them 16 B per lane (512 k per wave-load),
// two rows in flight; 6-step wavefront reduction per row.
constexpr int F16W_WAVES = 4;

template <bool NORM>
__global__ __launch_bounds__(F16W_WAVES * 64) void gemv_f16w_kernel(const _Float16* x, const _Float16* W, const _Float16* bias,
                                                                     _Float16* y, const _Float16* gamma, float eps, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* xl = (_Float16*)smem;
    float* red = (float*)(smem + (size_t)K * 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunks = K >> 3;
    if (!NORM) {
        for (int c = tid; c < chunks; c += F16W_WAVES * 64) *(h8*)(xl + 8 * c) = *(const h8*)(x + 8 * c);
    } else {
        float ss = 0.f;
        for (int c = tid; c < chunks; c += F16W_WAVES * 64) {
            h8 v = *(const h8*)(x + 8 * c);
            *(h8*)(xl + 8 * c) = v;
#pragma unroll
            for (int i = 0; i < 8; ++i) { float f = (float)v[i]; ss += f * f; }
        }
        ss = wave_sum_f(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < F16W_WAVES; ++w) tot += red[w];
        const float rstd = rsqrtf(tot / (float)K + eps);
        for (int c = tid; c < chunks; c += F16W_WAVES * 64) {
            h8 v = *(h8*)(xl + 8 * c);
            h8 g = *(const h8*)(gamma + 8 * c);
            h8 r;
#pragma unroll
            for (int i = 0; i < 8; ++i) { _Float16 n = (_Float16)((float)v[i] * rstd); r[i] = g[i] * n; }
            *(h8*)(xl + 8 * c) = r;
        }
    }
    __syncthreads();
    const int gw = blockIdx.x * F16W_WAVES + wave, nw = gridDim.x * F16W_WAVES;
    const int steps = (K + 511) >> 9;               // 512 k per wave-load; lanes past K are masked (K % 8 == 0)
    for (int row = gw; row < N; row += nw) {
        const _Float16* wr = W + (size_t)row * K + 8 * lane;
        float acc = 0.f;
        for (int s = 0; s < steps; s += 4) {
            u4 buf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (s + j < steps && (s + j) * 512 + 8 * lane < K) buf[j] = AMQ_STREAM_LOAD((const u4*)(wr + (size_t)(s + j) * 512));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (s + j < steps && (s + j) * 512 + 8 * lane < K) {
                    const h8 xv = *(const h8*)(xl + (s + j) * 512 + 8 * lane);
                    const uint32_t wv[4] = {buf[j].x, buf[j].y, buf[j].z, buf[j].w};
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        h2 xp = {xv[2 * p], xv[2 * p + 1]};
                        acc = __builtin_amdgcn_fdot2(as_h2(wv[p]), xp, acc, false);
                    }
                }
            }
        }
        acc = wave_sum_f(acc);
        if (lane == 0) {
            _Float16 o = (_Float16)acc;
            if (bias) o = o + bias[row];
            y[row] = o;
        }
    }
}

hipError_t launch_gemv_f16w(const void* x, const void* W, const void* bias, void* y, const void* gamma, float eps,
                            int N, int K, hipStream_t st) {
    const size_t lds = (size_t)K * 2 + 64;
    const int grid = 1024;
    if (gamma)
        hipLaunchKernelGGL((gemv_f16w_kernel<true>), dim3(grid), dim3(F16W_WAVES * 64), lds, st, (const _Float16*)x, (const _Float16*)W,
                           (const _Float16*)bias, (_Float16*)y, (const _Float16*)gamma, eps, N, K);
    else
        hipLaunchKernelGGL((gemv_f16w_kernel<false>), dim3(grid), dim3(F16W_WAVES * 64), lds, st, (const _Float16*)x, (const _Float16*)W,
                           (const _Float16*)bias, (_Float16*)y, (const _Float16*)nullptr, eps, N, K);
    return hipGetLastError();
}

// ------------------------------------------ RoPE + KV append + attention
// grid = (n_heads, batch); 256 threads; head_dim == 128.
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

__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* qs = (_Float16*)smem;                 // [128] rotated q
    _Float16* ks = (_Float16*)smem + ATT_D;         // [128] rotated new key (also what is appended)
    float* sc = (float*)(smem + 4 * ATT_D);         // [T] scores / probabilities
    __shared__ float red[8];
    __shared__ float part[16][ATT_D];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int pos = a.pos_dev ? *a.pos_dev : a.pos;
    const int T = pos + 1;
    const int group = a.n_heads / a.n_kv_heads;
    const int kvh = h / group;
    const _Float16* q = (const _Float16*)a.q + ((size_t)b * a.n_heads + h) * ATT_D;
    const _Float16* kn = (const _Float16*)a.k + ((size_t)b * a.n_kv_heads + kvh) * ATT_D;
    const _Float16* vn = (const _Float16*)a.v + ((size_t)b * a.n_kv_heads + kvh) * ATT_D;
    _Float16* kc = (_Float16*)a.kcache + ((size_t)b * a.n_kv_heads + kvh) * (size_t)a.max_seq * ATT_D;
    _Float16* vc = (_Float16*)a.vcache + ((size_t)b * a.n_kv_heads + kvh) * (size_t)a.max_seq * ATT_D;

    // every thread fetches its cached key row while wave 0 rotates q / k: one key per thread per pass
    h8 krow[16];
    const int t_own = tid;
    if (t_own < pos) {
#pragma unroll
        for (int j = 0; j < 16; ++j) krow[j] = *(const h8*)(kc + (size_t)t_own * ATT_D + 8 * j);
    }
    // ... and the first 8 value rows of its (key group, 8-dim slice): the K and V HBM round trips overlap
    constexpr int VPRE = 8;
    h8 vpre[VPRE];
    {
        const int kg0 = tid >> 4, l0 = tid & 15;
#pragma unroll
        for (int i = 0; i < VPRE; ++i) {
            const int t = kg0 + 16 * i;
            if (t < pos) vpre[i] = *(const h8*)(vc + (size_t)t * ATT_D + 8 * l0);
        }
    }
    if (tid < 64) {
        const int i = tid;                          // rotary pair (i, i + 64)
        _Float16 c16, s16;
        if (a.rope_table) {
            const h2 cs2 = ((const h2*)a.rope_table)[(size_t)pos * 64 + i];
            c16 = cs2.x; s16 = cs2.y;
        } else {
            rope_cs(a.rope_theta, pos, i, &c16, &s16);
        }
        const _Float16 q0 = q[i], q1 = q[i + 64];
        qs[i] = q0 * c16 + (-q1) * s16;             // q*cos + rotate_half(q)*sin  (fp16 ops, HF apply_rotary_pos_emb)
        qs[i + 64] = q1 * c16 + q0 * s16;
        const _Float16 k0 = kn[i], k1 = kn[i + 64];
        const _Float16 r0 = k0 * c16 + (-k1) * s16, r1 = k1 * c16 + k0 * s16;
        ks[i] = r0;
        ks[i + 64] = r1;
        if (h % group == 0) {                       // one query head per kv group appends to the cache
            kc[(size_t)pos * ATT_D + i] = r0;
            kc[(size_t)pos * ATT_D + i + 64] = r1;
            vc[(size_t)pos * ATT_D + i] = vn[i];
            vc[(size_t)pos * ATT_D + i + 64] = vn[i + 64];
        }
    }
    __syncthreads();

    // scores: one key per thread (the new key comes from LDS)
    const float scale = rsqrtf((float)ATT_D);
    float lmax = -INFINITY;
    for (int t = tid; t < T; t += 256) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            h8 kv;
            if (t == pos) kv = *(const h8*)(ks + 8 * j);
            else if (t == t_own) kv = krow[j];
            else kv = *(const h8*)(kc + (size_t)t * ATT_D + 8 * j);
            const h8 qv = *(const h8*)(qs + 8 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s = __builtin_amdgcn_fdot2((h2){qv[2 * e], qv[2 * e + 1]}, (h2){kv[2 * e], kv[2 * e + 1]}, s, false);
        }
        // HF eager attention: matmul(q, k^T) -> fp16, * scaling -> fp16, softmax in fp32
        const float sv = (float)(_Float16)((float)(_Float16)s * scale);
        sc[t] = sv;
        lmax = fmaxf(lmax, sv);
    }
    lmax = wave_max_f(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int t = tid; t < T; t += 256) {
        const float e = __expf(sc[t] - gmax);
        sc[t] = e;
        lsum += e;
    }
    lsum = wave_sum_f(lsum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = lsum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);

    // out = sum_t p_t * V[t]: 16 key groups x 16 lanes, 8 dims (16 B) per lane
    const int kg = tid >> 4, l = tid & 15;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < VPRE; ++i) {
        const int t = kg + 16 * i;
        if (t < T) {
            const _Float16 p16 = (_Float16)(sc[t] * inv);        // softmax(...).to(fp16)
            const h8 vv = (t == pos) ? *(const h8*)(vn + 8 * l) : vpre[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += (float)p16 * (float)vv[e];
        }
    }
    for (int t = kg + 16 * VPRE; t < T; t += 16) {
        const _Float16 p16 = (_Float16)(sc[t] * inv);
        const h8 vv = (t == pos) ? *(const h8*)(vn + 8 * l) : *(const h8*)(vc + (size_t)t * ATT_D + 8 * l);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += (float)p16 * (float)vv[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[kg][8 * l + e] = o[e];
    __syncthreads();
    if (tid < ATT_D) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) tot += part[g][tid];
        _Float16* out = (_Float16*)a.out + ((size_t)b * a.n_heads + h) * ATT_D;
        out[tid] = (_Float16)tot;
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

hipError_t launch_attn_decode(const AttnArgs& a, int batch, hipStream_t st) {
    const size_t lds = 4 * ATT_D + (size_t)a.max_seq * 4;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(attn_decode_kernel, dim3(a.n_heads, batch), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace amq
