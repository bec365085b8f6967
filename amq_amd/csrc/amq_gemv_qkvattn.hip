// amq_gemv_qkvattn.hip -- A/B route (libamq_hip_ab.so only, make ab): the q/k/v GEMV and the decode attention as ONE launch.
// Slower than the two launches it replaces (DESIGN.md / HISTORY.md, profiles/r03_qkv_attn_fused_negative.txt); kept for the record.
#include "amq_gemv_body.cuh"

namespace amq {

// ---------------------------------------------------------------- q/k/v GEMV + attention in ONE launch (decode, batch 1)
// VERDICT r2 item 1(a).  The q/k/v launch and the attention launch of a decode block become one: a workgroup stores its row-tiles
// of q, k or v as agent-scope stores, drains them, and adds its row-tile count to the ticket of every query head they belong to
// (24 row-tiles per head: 8 of q, 8 of k, 8 of v); the workgroup that owns the head's FIRST q row-tile then requests the head's
// cached K / V rows, waits for the ticket (bounded poll, one lane), reads q / k / v back with agent-scope loads and runs
// amq::attn_decode_kernel's arithmetic, expression for expression (amq_decode.hip; 512 threads, so NW = 8 here).  Nobody else
// waits.  Removes one launch boundary + one prologue per block; the outputs are bit-identical to the two launches.
// Hand-off protocol: cdna_hip_programming.md Guideline 16 (R1) -- every storing wave drains, workgroup barrier, ONE lane adds;
// consumer: relaxed poll, workgroup barrier, every load of the handed-off bytes an sc1 load.  Tickets are zero before and
// after every launch (the attention workgroup resets its head's once all 24 arrivals are in).
struct AttnTail { void* kc; void* vc; const void* state; void* out; int* tickets; int n_heads, n_kv_heads, max_seq; };
constexpr int QA_PF = 4;               // cached K / V rows per 16-lane group held in registers (128 keys): the kernel must stay at 80 VGPRs
constexpr int QA_GROUPS = 32;
constexpr unsigned QA_SPIN_LIMIT = 1u << 20;

__device__ __forceinline__ float qa_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ float qa_row16_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));
    return v;
}
__device__ __forceinline__ float qa_wave4(float v, bool mx) {
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return mx ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}
__device__ __forceinline__ _Float16 qa_ldh(const _Float16* p) {
    return __builtin_bit_cast(_Float16, __hip_atomic_load((const unsigned short*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// attention of head h by this (512-thread) workgroup; q / kn / vn: the new token's projections (handed off inside the launch)
__device__ __forceinline__ void qkv_attention(const AttnTail& at, int h, const _Float16* qv_, const _Float16* kn_, const _Float16* vn_,
                                              unsigned char* sm) {
    constexpr int D = 128, THREADS = 512;
    _Float16* qs = (_Float16*)sm;
    _Float16* ks = qs + D;
    _Float16* vs = qs + 2 * D;
    float* red = (float*)(sm + 6 * D);                          // [16]
    float* part = (float*)(sm + 6 * D + 64);                    // [32][128]
    float* sc = (float*)(sm + 6 * D + 64 + QA_GROUPS * D * 4);  // [T]
    int* flag = (int*)(sm + 6 * D + 60);
    const int tid = threadIdx.x;
    const int group = at.n_heads / at.n_kv_heads, kvh = h / group;
    const int grp = tid >> 4, l16 = tid & 15;
    const int pos = *(const int*)((const char*)at.state + 256);
    const bool pos_ok = pos >= 0 && pos < at.max_seq;
    const _Float16* q = qv_ + (size_t)h * D;
    const _Float16* kn = kn_ + (size_t)kvh * D;
    const _Float16* vn = vn_ + (size_t)kvh * D;
    _Float16* kc = (_Float16*)at.kc + (size_t)kvh * (size_t)at.max_seq * D;
    _Float16* vc = (_Float16*)at.vc + (size_t)kvh * (size_t)at.max_seq * D;
    const int T = pos + 1;
    const int last_old = pos > 0 ? pos - 1 : 0;
    // the cached rows do not depend on this token: requested before the wait for the ticket
    h8 krow[QA_PF], vrow[QA_PF];
    if (pos_ok) {
#pragma unroll
        for (int i = 0; i < QA_PF; ++i) {
            if (QA_GROUPS * i < pos) {
                int t = grp + QA_GROUPS * i;
                t = t < last_old ? t : last_old;
                krow[i] = *(const h8*)(kc + (size_t)t * D + 8 * l16);
            }
        }
#pragma unroll
        for (int i = 0; i < QA_PF; ++i) {
            if (QA_GROUPS * i < pos) {
                int t = grp + QA_GROUPS * i;
                t = t < last_old ? t : last_old;
                vrow[i] = *(const h8*)(vc + (size_t)t * D + 8 * l16);
            }
        }
    }
    if (tid == 0) {
        int bad = 0;
        unsigned spins = 0;
        while (__hip_atomic_load(at.tickets + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 24) {
            if (++spins > QA_SPIN_LIMIT) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        // all 24 arrivals are in: reset for the next launch.  After a time-out the ticket is LEFT ALONE -- late producers still add to it, and
        // a reset here would leave it non-zero behind them without anybody knowing; the sticky error word tells the host to re-zero the tickets
        if (!bad) __hip_atomic_store(at.tickets + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bad || !pos_ok) *(int*)((char*)const_cast<void*>(at.state) + 260) = 1;               // sticky error word (amq_decode.hip)
        *flag = bad;
    }
    __syncthreads();
    if (*flag || !pos_ok) return;

    _Float16 q0 = 0, q1 = 0, k0 = 0, k1 = 0, v0 = 0, v1 = 0;
    h2 cs = {(_Float16)1.f, (_Float16)0.f};
    if (tid < 64) {
        q0 = qa_ldh(q + tid); q1 = qa_ldh(q + tid + 64);
        k0 = qa_ldh(kn + tid); k1 = qa_ldh(kn + tid + 64);
        cs = ((const h2*)at.state)[tid];
        v0 = qa_ldh(vn + tid); v1 = qa_ldh(vn + tid + 64);
        const _Float16 c16 = cs.x, s16 = cs.y;
        const int i = tid;
        qs[i] = q0 * c16 + (-q1) * s16;
        qs[i + 64] = q1 * c16 + q0 * s16;
        const _Float16 r0 = k0 * c16 + (-k1) * s16, r1 = k1 * c16 + k0 * s16;
        ks[i] = r0;
        ks[i + 64] = r1;
        vs[i] = v0;
        vs[i + 64] = v1;
        if (h % group == 0) {
            kc[(size_t)pos * D + i] = r0;
            kc[(size_t)pos * D + i + 64] = r1;
            vc[(size_t)pos * D + i] = v0;
            vc[(size_t)pos * D + i + 64] = v1;
        }
    }
    __syncthreads();
    const float scale = rsqrtf((float)D);
    {
        const h8 qv = *(const h8*)(qs + 8 * l16);
        const h8 knew = *(const h8*)(ks + 8 * l16);
        auto score = [&](const h8& kv) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s = __builtin_amdgcn_fdot2((h2){qv[2 * e], qv[2 * e + 1]}, (h2){kv[2 * e], kv[2 * e + 1]}, s, false);
            s = qa_row16_sum(s);
            return (float)(_Float16)((float)(_Float16)s * scale);
        };
#pragma unroll
        for (int i = 0; i < QA_PF; ++i) {
            if (QA_GROUPS * i < T) {
                const int t = grp + QA_GROUPS * i;
                const float sv = score(t == pos ? knew : krow[i]);
                if (t < T && l16 == 0) sc[t] = sv;
            }
        }
        for (int t = grp + QA_GROUPS * QA_PF; t < T; t += QA_GROUPS) {
            const h8 kv = (t == pos) ? knew : *(const h8*)(kc + (size_t)t * D + 8 * l16);
            const float sv = score(kv);
            if (l16 == 0) sc[t] = sv;
        }
    }
    __syncthreads();
    float lmax = -INFINITY;
    for (int t = tid; t < T; t += THREADS) lmax = fmaxf(lmax, sc[t]);
    lmax = qa_wave4(qa_row16_max(lmax), true);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    float gmax = red[0];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w) gmax = fmaxf(gmax, red[w]);
    float lsum = 0.f;
    for (int t = tid; t < T; t += THREADS) {
        const float e = __expf(sc[t] - gmax);
        sc[t] = e;
        lsum += e;
    }
    lsum = qa_wave4(qa_row16_sum(lsum), false);
    if ((tid & 63) == 0) red[THREADS / 64 + (tid >> 6)] = lsum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) tot += red[THREADS / 64 + w];
    const float inv = 1.0f / tot;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const h8 vnew = *(const h8*)(vs + 8 * l16);
#pragma unroll
    for (int i = 0; i < QA_PF; ++i) {
        const int t = grp + QA_GROUPS * i;
        if (QA_GROUPS * i < T && t < T) {
            const _Float16 p16 = (_Float16)(sc[t] * inv);
            const h8 vv = (t == pos) ? vnew : vrow[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += (float)p16 * (float)vv[e];
        }
    }
    for (int t = grp + QA_GROUPS * QA_PF; t < T; t += QA_GROUPS) {
        const _Float16 p16 = (_Float16)(sc[t] * inv);
        const h8 vv = (t == pos) ? vnew : *(const h8*)(vc + (size_t)t * D + 8 * l16);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += (float)p16 * (float)vv[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[grp * D + 8 * l16 + e] = o[e];
    __syncthreads();
    if (tid < D) {
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < QA_GROUPS; ++g) acc += part[g * D + tid];
        ((_Float16*)at.out)[(size_t)h * D + tid] = (_Float16)acc;
    }
}

template <int U, int XCH>
__global__ __launch_bounds__(512, 6) void gemv_qkv_attn_kernel(const void* p_x, const void* p_xw, const void* p_qw0, const void* p_mt0, int p_K,
                                                            int p_m_nseg, int p_rpt, int p_n_rt0, int p_key0, float p_eps, GemvKArgs blk,
                                                            AttnTail at) {
    constexpr int NW = 8, PRO = PRO_RMSNORM, MATH = MATH_EXACT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifndef AMQ_NO_PRIO_PROGRESS
    __builtin_amdgcn_s_setprio(3);
#endif
    GemvHot a;
    a.x = p_x; a.x2 = p_xw; a.gamma = p_xw;
    a.K = p_K; a.M = 1; a.rpt = p_rpt; a.eps = p_eps;
    a.x_stride = a.K;
    const int xs = a.K + XPAD;
    _Float16* xl = (_Float16*)smem;
    const size_t xbytes = ((size_t)xs * 2 + 15) & ~(size_t)15;
    float* xg = (float*)(smem + xbytes);
    float* red = (float*)(smem + xbytes);                                       // [2][NW][16][16]
    const int bid = (int)blockIdx.x;
    int sidx = 0, wgb = 0, nrt = p_n_rt0, key = p_key0;
    const void* qwp = p_qw0;
    const void* mtp = p_mt0;
    asm volatile("" ::"s"(blk.wg_begin[1]), "s"(blk.wg_begin[2]), "s"(blk.n_rt[1]), "s"(blk.n_rt[2]), "s"(blk.key[1]), "s"(blk.key[2]),
                 "s"(blk.qweight[1]), "s"(blk.qweight[2]), "s"(blk.meta[1]), "s"(blk.meta[2]));
#pragma unroll
    for (int i = 1; i < 3; ++i) {
        const bool take = bid >= blk.wg_begin[i];
        sidx = take ? i : sidx;
        wgb = take ? blk.wg_begin[i] : wgb;
        nrt = take ? blk.n_rt[i] : nrt;
        key = take ? blk.key[i] : key;
        qwp = take ? blk.qweight[i] : qwp;
        mtp = take ? blk.meta[i] : mtp;
    }
    const int local = bid - wgb;
    XRegs xr;
    x_issue<PRO, NW, XCH>(a, xr);
    switch (key) {
        case 4 * 4 + MODE_HQQ: gemv_body<4, MODE_HQQ, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, 1, xr); break;
        case 3 * 4 + MODE_HQQ: gemv_body<3, MODE_HQQ, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, 1, xr); break;
        case 2 * 4 + MODE_HQQ: gemv_body<2, MODE_HQQ, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, 1, xr); break;
        case 4 * 4 + MODE_FMA: gemv_body<4, MODE_FMA, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, 1, xr); break;
        case 3 * 4 + MODE_FMA: gemv_body<3, MODE_FMA, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, 1, xr); break;
        default:               gemv_body<2, MODE_FMA, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, 1, xr); break;
    }
    // ---- publish this workgroup's row-tiles: drain (the storing threads sit in wave 0), barrier, one lane adds to the tickets
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int sp_base = nrt & GEMV_SPLIT_BASE_MASK, sp_rem = nrt >> GEMV_SPLIT_BASE_BITS;           // (nrt = gemv_split() of the segment: gemv_body's row-tile range)
    const int rt0 = local * sp_base + (local < sp_rem ? local : sp_rem);
    const int n_my = sp_base + (local < sp_rem ? 1 : 0);
    const int group = at.n_heads / at.n_kv_heads;
    if (threadIdx.x == 0) {
        for (int rt = rt0; rt < rt0 + n_my; ++rt) {
            if (sidx == 0) {
                __hip_atomic_fetch_add(at.tickets + (rt >> 3), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (int j = 0; j < group; ++j)
                    __hip_atomic_fetch_add(at.tickets + (rt >> 3) * group + j, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- the workgroup that owns a head's first q row-tile runs that head's attention
    if (sidx == 0) {
        for (int h = (rt0 + 7) >> 3; 8 * h < rt0 + n_my && h < at.n_heads; ++h) {
            __syncthreads();                               // (LDS of the GEMV body / of the previous head is free)
            qkv_attention(at, h, (const _Float16*)blk.y[0], (const _Float16*)blk.y[1], (const _Float16*)blk.y[2], smem);
        }
    }
}

size_t gemv_qkv_attn_lds_bytes(int K, int max_seq) {
    const size_t g = gemv_lds_bytes(1, K, 16);
    const size_t t = 6 * 128 + 64 + (size_t)QA_GROUPS * 128 * 4 + (size_t)max_seq * 4;
    return g > t ? g : t;
}

// q / k / v GEMV (segments 0 .. 2 of `a`: fused RMSNorm prologue, M = 1, no bias / residual) + attention.  tickets: int32 [n_heads], zero.
hipError_t launch_gemv_qkv_attn(GemvArgs& a, const AttnArgs& t, int* tickets, hipStream_t st) {
    StreamDevice sd_(st);                                  // kernel attributes are per device: the stream's, not the current one
    int total_rt = 0;
    for (int i = 0; i < 3; ++i) { a.seg[i].n_rt = a.seg[i].N / 16; total_rt += a.seg[i].n_rt; }
    const int chunks = a.K >> 3;
    if (chunks > 1024) return hipErrorInvalidValue;               // K <= 8192: the 8-wave decode geometry (every Llama-2 hidden size)
    const bool two = chunks > 512;
    const int target = two ? 512 : 768;                           // launch_gemv's grid for these shapes
    int rpt = (total_rt + target - 1) / target;
    if (rpt < 1) rpt = 1;
    int wg = 0;
    for (int i = 0; i < 3; ++i) {
        a.seg[i].wg_begin = wg;
        a.seg[i].wg_count = (a.seg[i].n_rt + rpt - 1) / rpt;
        wg += a.seg[i].wg_count;
    }
    GemvKArgs k{};
    k.x = a.x; k.x2 = nullptr; k.gamma = a.gamma;
    k.M = 1; k.K = a.K; k.x_stride = a.K; k.nseg = 3; k.eps = a.eps; k.rpt = rpt;
    for (int i = 0; i < 3; ++i) {
        const GemvSeg& s = a.seg[i];
        k.wg_begin[i] = s.wg_begin; k.n_rt[i] = gemv_split(s.n_rt, s.wg_count); k.key[i] = s.bits * 4 + (s.mode == MODE_FMA1 ? (int)MODE_FMA : s.mode);
        k.qweight[i] = s.qweight; k.meta[i] = s.meta; k.bias[i] = nullptr; k.residual[i] = nullptr; k.y[i] = s.y; k.y_stride[i] = s.N;
    }
    k.wg_begin[3] = 0x7fffffff;
#ifdef AMQ_STAMP
    k.stamps = nullptr;
#endif
    AttnTail at{t.kcache, t.vcache, t.rope_cur, t.out, tickets, t.n_heads, t.n_kv_heads, t.max_seq};
    const size_t lds = gemv_qkv_attn_lds_bytes(a.K, t.max_seq);
    static unsigned long long attr2_done = 0, attr1_done = 0;        // (the LDS need is bounded by the limit the C ABI checks: one attribute value per kernel)
    if (two) {
        auto kern = gemv_qkv_attn_kernel<2, 2>;
        if (lds > 64 * 1024) { hipError_t e = ensure_dyn_lds(attr2_done, (const void*)kern, 160 * 1024); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(kern, dim3(wg), dim3(512), lds, st, k.x, k.gamma, k.qweight[0], k.meta[0], k.K, 1 | (3 << 16), rpt, k.n_rt[0], k.key[0], k.eps, k, at);
    } else {
        auto kern = gemv_qkv_attn_kernel<2, 1>;
        if (lds > 64 * 1024) { hipError_t e = ensure_dyn_lds(attr1_done, (const void*)kern, 160 * 1024); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(kern, dim3(wg), dim3(512), lds, st, k.x, k.gamma, k.qweight[0], k.meta[0], k.K, 1 | (3 << 16), rpt, k.n_rt[0], k.key[0], k.eps, k, at);
    }
    return hipGetLastError();
}


}  // namespace amq
