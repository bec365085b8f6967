// amq_gemv.hip -- weight-streaming y = x . W^T for few rows (decode), gfx950.
//
// Replaces, for rows < 8/128, the reference's
//   VecQuant{2,3,4}MatMulKernelFaster_old (amq/kernel/AutoGPTQ/auto_gptq_kernel.cu:160-440)
//   gemv_kernel<2,Batch,256,128>          (amq/kernel/ft/quantization_new/gemv/gemv_cuda.cu:73-204)
// with one kernel family over the native AMQ-T16 layout (amq_common.cuh).
//
// Structure (every byte of W is read exactly once; see DESIGN.md for the measurements behind each choice):
//   * a workgroup owns whole row-tiles (16 output rows x all of K, one contiguous
//     byte range each) and walks rt = first, first + stride, ...; x is staged
//     (RMSNorm / SiLU*mul fused) ONCE per workgroup, not once per row-tile.
//   * its NW waves take the K/128 tiles of a row-tile round-robin.  Each wave keeps
//     U tile loads (16/12/8 B per lane, non-temporal) in flight in a register ring
//     that runs ACROSS row-tile boundaries and is primed before x is staged.
//     A slot is refilled only after its tile is consumed (straight-line pipeline,
//     counted vmcnt; the tail drains).
//   * the unpacked fp16x8 register block IS the MFMA B operand (layout chosen for
//     that): v_mfma_f32_16x16x32_f16 against x rows read from LDS with ds_read_b128.
//     W never touches LDS; M = 1..16 cost the same VALU work.
//   * MATH_EXACT  (default): scaled-subnormal unpack = the reference's two-rounding
//                 dequant, 12 VALU cycles per weight pair (amq_common.cuh).
//     MATH_DOT    (A/B only, M == 1): same weights, v_dot2c_f32_f16 + wavefront-shuffle
//                 reduction instead of MFMA.
//     MATH_LINEAR (opt-in): every field is shifted to one mantissa position and fed to the MFMA as the fp16
//                 subnormal q*2^(SH-24) (gfx950 MFMA honours fp16 subnormals -- measured); scale / zero are applied
//                 per group in fp32:  y += s*(2^(24-SH)*sum(x q) - z*sum_g(x)), with the group sums of x taken once in
//                 the staging pass.  4 VALU cycles per pair; results are the real-valued dequant (no per-weight fp16
//                 roundings), ~3e-4 of the output rms away from the reference's rounded-weight result.
//   * per row-tile, fixed-order cross-wave sum through double-buffered LDS and one
//     barrier: deterministic, no atomics.
//   * several linears that share x (q/k/v, gate/up) with different bit-widths
//     run as segments of ONE launch; a workgroup serves one segment.
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

constexpr int XPAD = 8;              // halves of padding per staged x row (16 B)

// Kernel argument block.  Everything a wave needs before its first weight load ("hot") sits at
// static offsets in the first 168 bytes, structure-of-arrays over the segments, so the compiler
// fetches it with ONE batch of scalar loads (one kernarg round trip, measured ~0.3 us each, instead of the
// three dependent ones of an array-of-structs with a dynamic segment index); the epilogue-only fields follow.
struct GemvKArgs {
    const void* x; const void* x2; const void* gamma;
    int M, K, x_stride, nseg;
    float eps; int rpt, pad0_, pad1_;
    int wg_begin[GEMV_MAX_SEG];
    int n_rt[GEMV_MAX_SEG];
    int key[GEMV_MAX_SEG];                 // bits * 4 + mode
    const void* qweight[GEMV_MAX_SEG];
    const void* meta[GEMV_MAX_SEG];
    // ---- cold: epilogue only
    const void* bias[GEMV_MAX_SEG];
    const void* residual[GEMV_MAX_SEG];
    void* y[GEMV_MAX_SEG];
    int y_stride[GEMV_MAX_SEG];
#ifdef AMQ_STAMP
    unsigned long long* stamps;
#endif
};

struct SegOut { const _Float16* bias; const _Float16* residual; _Float16* y; int y_stride; };

// register-resident view of the launch-wide arguments (built from the preloaded kernel arguments)
struct GemvHot {
    const void* x; const void* x2; const void* gamma;
    int M, K, x_stride, rpt;
    float eps;
};

#ifdef AMQ_STAMP
// diagnostic build: slot i of this workgroup's 32-entry record <- 100 MHz realtime counter (comparable across CUs)
#define AMQ_STAMP_AT(a_, slot_)                                                                  \
    do {                                                                                         \
        if ((a_).stamps && (threadIdx.x & 63) == 0)                                              \
            (a_).stamps[(size_t)blockIdx.x * 128 + (slot_)] = __builtin_amdgcn_s_memrealtime();   \
    } while (0)
#else
#define AMQ_STAMP_AT(a_, slot_) do { } while (0)
#endif
enum { MATH_EXACT = 0, MATH_DOT = 1, MATH_LINEAR = 2 };

// whole-wave sum without LDS-crossbar shuffles (six dependent ds_bpermute round trips cost ~0.3 us on the prologue's
// critical path): DPP inside the four 16-lane rows, then four v_readlane; every lane gets the total (fixed order)
__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
}

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + __expf(-g)); }

// ---------------------------------------------------------------- staging
// Writes the (transformed) activations into LDS as fp16.
//   exact / dot : xl[m][xs]
//   linear      : the same xl[m][xs] plus xg[G][16] = per-group (128 k) sums of x per row (fp32; rows >= M are zero)
template <int PRO, int NW, bool LIN>
__device__ __forceinline__ void stage_x(const GemvHot& a, _Float16* xl, float* xg, float* red, int xs) {
    constexpr int THREADS = NW * 64;
    const int tid = threadIdx.x;
    const int K = a.K;
    const int chunks = K >> 3;      // 8 halves per chunk
    if (!LIN) {
        // Rows side by side: a row is staged by WPR = NW / M (a power of two, >= 1) waves, NW / WPR rows per round, ONE barrier
        // pair per round for the rows' sums of squares -- not one pair per row with the whole workgroup on each row in turn
        // (~0.5 us a row: 8 sequences took 3.05 ms a step where one takes 1.19).
        const int lane = tid & 63, wave = tid >> 6;
        int wpr = 1;
        while (wpr * 2 * a.M <= NW) wpr *= 2;
        const int rows_per_round = NW / wpr, sub = wave % wpr, stride = wpr * 64;
        for (int m0 = 0; m0 < a.M; m0 += rows_per_round) {
            const int m = m0 + wave / wpr;
            const bool on = m < a.M;
            const _Float16* xrow = (const _Float16*)a.x + (size_t)(on ? m : 0) * a.x_stride;
            _Float16* lrow = xl + (size_t)(on ? m : 0) * xs;
            float rstd = 1.0f;
            if (PRO == PRO_RMSNORM) {
                float ss = 0.f;
                if (on)
                    for (int c = sub * 64 + lane; c < chunks; c += stride) {
                        h8 v = *(const h8*)(xrow + 8 * c);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { float f = (float)v[i]; ss += f * f; }
                    }
                ss = wave_sum(ss);
                if (m0) __syncthreads();            // the previous round's readers of red[] are done
                if (lane == 0) red[wave] = ss;
                __syncthreads();
                float tot = 0.f;
                for (int i = 0; i < wpr; ++i) tot += red[wave - sub + i];
                rstd = rsqrtf(tot / (float)K + a.eps);
            }
            if (on)
                for (int c = sub * 64 + lane; c < chunks; c += stride) {
                    h8 v = *(const h8*)(xrow + 8 * c);
                    h8 r;
                    if (PRO == PRO_NONE) {
                        r = v;
                    } else if (PRO == PRO_SILU_MUL) {
                        const h8 u = *(const h8*)((const _Float16*)a.x2 + (size_t)m * a.x_stride + 8 * c);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { _Float16 s = (_Float16)silu_f((float)v[i]); r[i] = s * u[i]; }
                    } else {
                        const h8 gm = *(const h8*)((const _Float16*)a.gamma + 8 * c);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { _Float16 nrm = (_Float16)((float)v[i] * rstd); r[i] = gm[i] * nrm; }
                    }
                    *(h8*)(lrow + 8 * c) = r;
                }
        }
        return;         // (the caller's barrier publishes xl; red[] is next written only after that barrier)
    }
    if (LIN)
        for (int i = tid; i < (K >> 7) * 16; i += THREADS) xg[i] = 0.f;      // (rows written below are disjoint from these only by thread; ordered by the barrier after staging)
    if (LIN) __syncthreads();
    for (int m = 0; m < a.M; ++m) {
        const _Float16* xrow = (const _Float16*)a.x + (size_t)m * a.x_stride;
        _Float16* lrow = xl + (size_t)m * xs;
        float rstd = 1.0f;
        if (PRO == PRO_RMSNORM) {
            float ss = 0.f;
            for (int c = tid; c < chunks; c += THREADS) {
                h8 v = *(const h8*)(xrow + 8 * c);
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
            rstd = rsqrtf(tot / (float)K + a.eps);
        }
        for (int c = tid; c < chunks; c += THREADS) {
            h8 v = *(const h8*)(xrow + 8 * c);
            h8 r;
            if (PRO == PRO_NONE) {
                r = v;
            } else if (PRO == PRO_SILU_MUL) {
                // x = fp16(fp16(silu(gate)) * up)  -- HF LlamaMLP: act_fn(gate) * up
                const h8 u = *(const h8*)((const _Float16*)a.x2 + (size_t)m * a.x_stride + 8 * c);
#pragma unroll
                for (int i = 0; i < 8; ++i) { _Float16 s = (_Float16)silu_f((float)v[i]); r[i] = s * u[i]; }
            } else {
                // HF LlamaRMSNorm: weight * (x.float() * rstd).to(fp16)
                const h8 gm = *(const h8*)((const _Float16*)a.gamma + 8 * c);
#pragma unroll
                for (int i = 0; i < 8; ++i) { _Float16 nrm = (_Float16)((float)v[i] * rstd); r[i] = gm[i] * nrm; }
            }
            *(h8*)(lrow + 8 * c) = r;
            if (LIN) {
                float cs = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) cs += (float)r[i];
                cs += __shfl_xor(cs, 1); cs += __shfl_xor(cs, 2); cs += __shfl_xor(cs, 4); cs += __shfl_xor(cs, 8);
                if ((tid & 15) == 0) xg[(size_t)(c >> 4) * 16 + m] = cs;          // 16 chunks = one 128-k group; [G][16 rows]
            }
        }
    }
}

// Single-row (decode) staging, split in two so that the activation loads are the OLDEST entries of the wave's
// vector-memory queue: x_issue() runs before the weight ring is primed, x_finish() after it.  vmcnt waits are
// in issue order, so staging behind the primed weight tiles (the previous arrangement) made every workgroup wait
// for its first tiles -- 2-4 us under load (profiles/r01b_gemv_stamps.txt) -- before x could be written to LDS.
constexpr int XC_MAX = 4;            // 16-byte chunks of x per thread held in registers at most (K <= 32 * threads)
struct XRegs { h8 v[XC_MAX]; h8 w[XC_MAX]; };   // w: up (SiLU*mul) or gamma (RMSNorm)
// chunks actually held: two only in the 16-wave workgroups (one per CU, 128 VGPRs available); the 8-wave ones must
// stay under 80 VGPRs for three workgroups per CU, and K <= 4096 needs one chunk per thread there
template <int NW> struct XCfg { static constexpr int XC = NW == 16 ? 2 : 1; };
// (a third variant, XCH = 4 at 16 waves, covers 16384 < K <= 32768: the 70B down_proj, K = 28672)

// No branches around the loads (indices are clamped instead): the compiler can only emit a COUNTED vmcnt for the
// later uses when every path between a load and its use issues the same vector-memory operations.
template <int PRO, int NW, int XCH>
__device__ __forceinline__ void x_issue(const GemvHot& a, XRegs& xr) {
    constexpr int THREADS = NW * 64;
    const int last = (a.K >> 3) - 1;
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        int c = (int)threadIdx.x + i * THREADS;
        c = c < last ? c : last;                                  // clamp: every lane loads, tail lanes discard
        xr.v[i] = *(const h8*)((const _Float16*)a.x + 8 * c);
        if (PRO == PRO_SILU_MUL) xr.w[i] = *(const h8*)((const _Float16*)a.x2 + 8 * c);
        if (PRO == PRO_RMSNORM) xr.w[i] = *(const h8*)((const _Float16*)a.gamma + 8 * c);
    }
}

template <int PRO, int NW, int XCH, bool LIN>
__device__ __forceinline__ void x_finish(const GemvHot& a, const XRegs& xr, _Float16* xl, float* xg, float* red) {
    constexpr int THREADS = NW * 64;
    const int tid = threadIdx.x;
    const int chunks = a.K >> 3;
    float rstd = 1.0f;
    if (PRO == PRO_RMSNORM) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            if (tid + i * THREADS < chunks) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { float f = (float)xr.v[i][e]; ss += f * f; }
            }
        }
        ss = wave_sum(ss);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[w];
        rstd = rsqrtf(tot / (float)a.K + a.eps);
    }
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + i * THREADS;
        float cs = 0.f;
        if (c < chunks) {
            h8 r;
            if (PRO == PRO_NONE) {
                r = xr.v[i];
            } else if (PRO == PRO_SILU_MUL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { _Float16 sg = (_Float16)silu_f((float)xr.v[i][e]); r[e] = sg * xr.w[i][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { _Float16 nrm = (_Float16)((float)xr.v[i][e] * rstd); r[e] = xr.w[i][e] * nrm; }
            }
            *(h8*)(xl + 8 * c) = r;
            if (LIN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) cs += (float)r[e];
            }
        }
        if (LIN) {
            // 16 chunks = one 128-k group = one 16-lane DPP row (threads per pass are a multiple of 16); row 0 of xg[G][16]
            cs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cs), 0xB1, 0xF, 0xF, false));
            cs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cs), 0x4E, 0xF, 0xF, false));
            cs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cs), 0x141, 0xF, 0xF, false));
            cs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cs), 0x140, 0xF, 0xF, false));
            if ((tid & 15) == 0 && c < chunks) xg[(size_t)(c >> 4) * 16] = cs;
        }
    }
    // (the caller's barrier publishes xl; red[] is next written only after that barrier)
}

// ---------------------------------------------------------------- epilogue
__device__ __forceinline__ void store_out(const SegOut& s, int m, int n, float acc) {
    _Float16 y = (_Float16)acc;                                   // fp16(matmul)
    if (s.bias) y = y + s.bias[n];                                // out + bias      (fp16 add)
    if (s.residual) y = s.residual[(size_t)m * s.y_stride + n] + y;  // residual + out
    s.y[(size_t)m * s.y_stride + n] = y;
}

// shift+mask unpack for MATH_LINEAR: every pair at ONE mantissa position, out[4t+p] = packed fp16 subnormals q * 2^(SH-24)
template <int BITS> struct LinCfg;
template <> struct LinCfg<4> { static constexpr int SH = 6; };
template <> struct LinCfg<3> { static constexpr int SH = 7; };
template <> struct LinCfg<2> { static constexpr int SH = 8; };

template <int BITS>
__device__ __forceinline__ void unpack_lane_sub(const uint32_t* w, h2* out) {
    constexpr int SH = LinCfg<BITS>::SH;
    constexpr uint32_t fm = (1u << BITS) - 1u;
    constexpr uint32_t msk = (fm << SH) | ((fm << SH) << 16);
    if (BITS == 4) {            // fields at bits 0,4,8,12 of each half
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t u = w[t];
            out[4 * t + 0] = as_h2((u << 6) & msk);
            out[4 * t + 1] = as_h2((u << 2) & msk);
            out[4 * t + 2] = as_h2((u >> 2) & msk);
            out[4 * t + 3] = as_h2((u >> 6) & msk);
        }
    } else if (BITS == 2) {     // fields at bits 2i of each half
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const uint32_t u = w[d];
            out[8 * d + 0] = as_h2((u << 8) & msk);
            out[8 * d + 1] = as_h2((u << 6) & msk);
            out[8 * d + 2] = as_h2((u << 4) & msk);
            out[8 * d + 3] = as_h2((u << 2) & msk);
            out[8 * d + 4] = as_h2(u & msk);
            out[8 * d + 5] = as_h2((u >> 2) & msk);
            out[8 * d + 6] = as_h2((u >> 4) & msk);
            out[8 * d + 7] = as_h2((u >> 6) & msk);
        }
    } else {                    // fields at bits 3i of each half; pair 15 in bits 15 / 31 of the three dwords
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t u = w[d];
            out[5 * d + 0] = as_h2((u << 7) & msk);
            out[5 * d + 1] = as_h2((u << 4) & msk);
            out[5 * d + 2] = as_h2((u << 1) & msk);
            out[5 * d + 3] = as_h2((u >> 2) & msk);
            out[5 * d + 4] = as_h2((u >> 5) & msk);
        }
        out[15] = as_h2(((w[0] >> 8) & 0x00800080u) | ((w[1] >> 7) & 0x01000100u) | ((w[2] >> 6) & 0x02000200u));
    }
}

// ---------------------------------------------------------------- body
// RS: floats per wave in the cross-wave sum buffer red[2][NW][RS]: 256 = 16 x-rows x 16 columns; 128 for launches of at most 8 rows
// (several sequences decoded together), whose staged x is what limits the workgroups per CU
// SC1: the outputs are agent-scope (write-through) stores -- for a consumer INSIDE the same launch (gemv_qkv_attn_kernel)
// GP: (scale, zero) pairs per (row, tile) = 128 / group (amq_common.cuh); 2 / 4 are served by the exact-math body only
template <int BITS, int MODE, int PRO, int NW, int U, int MATH, int XCH, int RS = 256, bool SC1 = false, int GP = 1>
__device__ __forceinline__ void gemv_body(const GemvHot& a, const GemvKArgs& blk, int sidx, const void* qweight, const void* meta_base,
                                          int seg_n_rt, int local, _Float16* lds_x, const _Float16* xl, float* xg,
                                          float* red, int xs, bool fastx, const XRegs& xr) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = a.K >> 7;
    const int r = lane & 15, o = lane >> 4;
    const int nt = (G - wave + NW - 1) / NW;                      // tiles of one row-tile owned by this wave: g = wave + i*NW
    const int rt0 = local * a.rpt;                                // this workgroup's row-tiles: rt0 .. rt0 + n_my - 1 (contiguous bytes)
    const int n_my = (seg_n_rt - rt0) < a.rpt ? (seg_n_rt - rt0) : a.rpt;
    const int total = n_my * nt;
    const uint32_t* qw = (const uint32_t*)qweight;
    static_assert(GP == 1 || MATH == MATH_EXACT, "groups finer than 128: exact math only");
    const h2* mt = (const h2*)meta_base + r * GP;

#ifdef AMQ_ABL_NOMETA      /* ablation: no scale/zero traffic */
#define AMQ_META_LOAD(slot, tile_) meta[slot] = as_h2(0x40003c00u + (uint32_t)(tile_ & 1))
#else
#define AMQ_META_LOAD(slot, tile_)                                                               \
    do {                                                                                         \
        if constexpr (GP == 1) meta[slot] = as_h2(AMQ_STREAM_LOAD((const uint32_t*)(mt + (tile_) * 16)));   \
        else metag[slot] = load_meta_g<GP>(mt + (tile_) * (16 * GP));                            \
    } while (0)
#endif
#ifdef AMQ_ABL_NOXLDS      /* ablation: A operand from registers instead of LDS */
#define AMQ_XREAD(p) ((h8){(_Float16)1, (_Float16)2, (_Float16)-1, (_Float16)0.5f, (_Float16)1, (_Float16)-2, (_Float16)1, (_Float16)3} + (h8)(_Float16)(float)(kbase & 1))
#else
#define AMQ_XREAD(p) (*(const h8*)(p))
#endif
    LanePayload<BITS> pay[U];
    h2 meta[U];
    [[maybe_unused]] MetaG<GP> metag[U];                           // (GP > 1: `meta` is unused)
    int ii = 0, ij = 0;                                           // issue cursor (tile, row-tile)
#ifdef AMQ_ABL_NOLOAD      /* ablation build: no weight traffic, compute on whatever is in the registers */
#define AMQ_ISSUE(slot)                                                                          \
    do {                                                                                         \
        _Pragma("unroll") for (int d_ = 0; d_ < BITS; ++d_) asm volatile("" : "+v"(pay[slot].w[d_]));   \
        asm volatile("" : "+v"(meta[slot]));                                                     \
        if (++ii == nt) { ii = 0; ++ij; }                                                        \
    } while (0)
#else
#define AMQ_ISSUE_AT(slot, clamp_)                                                               \
    do {                                                                                         \
        size_t tile_ = (size_t)(rt0 + ij) * G + (wave + ii * NW);                                \
        if (clamp_) tile_ = tile_ < last_tile ? tile_ : last_tile;                               \
        pay[slot] = load_payload<BITS>(qw + tile_ * (64 * BITS), lane);                           \
        AMQ_META_LOAD(slot, tile_);                                                              \
        if (++ii == nt) { ii = 0; ++ij; }                                                        \
    } while (0)
#define AMQ_ISSUE(slot) AMQ_ISSUE_AT(slot, false)
#endif

    // prime the ring: unconditional (a wave with fewer than U tiles re-reads the workgroup's last tile and never
    // consumes it) so that the staging code below sees a fixed number of younger loads -> counted vmcnt for x
    const size_t last_tile = (size_t)(rt0 + n_my) * G - 1;
#ifdef AMQ_ABL_NOLOAD
#pragma unroll
    for (int u = 0; u < U; ++u) AMQ_ISSUE(u);
#else
#pragma unroll
    for (int u = 0; u < U; ++u) AMQ_ISSUE_AT(u, true);
#endif
    if (wave == 0) AMQ_STAMP_AT(blk, 1);

    // epilogue-only fields: fetched behind the primed ring (their latency hides under the first tiles)
    SegOut so;
    so.bias = (const _Float16*)blk.bias[sidx];
    so.residual = (const _Float16*)blk.residual[sidx];
    so.y = (_Float16*)blk.y[sidx];
    so.y_stride = blk.y_stride[sidx];
#ifndef AMQ_ABL_NOSTAGE
    if (fastx) x_finish<PRO, NW, XCH, MATH == MATH_LINEAR>(a, xr, lds_x, xg, red);
    else stage_x<PRO, NW, MATH == MATH_LINEAR>(a, lds_x, xg, red, xs);
#endif
    __syncthreads();
    if (wave == 0) AMQ_STAMP_AT(blk, 2);

    // bias / residual of the row-tile being accumulated, fetched a whole row-tile ahead: loaded inside the epilogue
    // they are the youngest entries of the vector-memory queue and the wait for them drains the weight ring
    _Float16 pf_bias = (_Float16)0.f, pf_res = (_Float16)0.f;
    const int e_m = (int)threadIdx.x >> 4, e_c = (int)threadIdx.x & 15;
    const bool e_on = (int)threadIdx.x < a.M * 16;
#define AMQ_EPI_PREFETCH(rt_)                                                                    \
    do {                                                                                         \
        if (e_on) {                                                                              \
            if (so.bias) pf_bias = so.bias[(rt_) * 16 + e_c];                                    \
            if (so.residual) pf_res = so.residual[(size_t)e_m * so.y_stride + (rt_) * 16 + e_c]; \
        }                                                                                        \
    } while (0)
    AMQ_EPI_PREFETCH(rt0);

    float acc1[4] = {0.f, 0.f, 0.f, 0.f};
    f4 accm = (f4){0.f, 0.f, 0.f, 0.f};
    const int mrow = r < a.M ? r : a.M - 1;                       // A rows >= M: any finite data, result unused
    const _Float16* xrow = xl + (size_t)mrow * xs + 8 * o;
    int ci = 0, cj = 0, par = 0;                                  // compute cursor, red[] parity

    int plevel_ = 3;
#define AMQ_SETPRIO_LEVEL()                                                                      \
    do {                                                                                         \
        if (plevel_ == 3) __builtin_amdgcn_s_setprio(3);                                         \
        else if (plevel_ == 2) __builtin_amdgcn_s_setprio(2);                                    \
        else if (plevel_ == 1) __builtin_amdgcn_s_setprio(1);                                    \
        else __builtin_amdgcn_s_setprio(0);                                                      \
    } while (0)
    // end of a row-tile for this wave: publish partials, one barrier, fixed-order sum by the first M*16 threads
#define AMQ_FINISH()                                                                             \
    do {                                                                                         \
        float* rp_ = red + par * (NW * RS);                                                      \
        if (MATH == MATH_DOT) {                                                                  \
            float v_ = (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);                                \
            v_ += __shfl_xor(v_, 16);                                                            \
            v_ += __shfl_xor(v_, 32);                                                            \
            if (lane < 16) rp_[wave * RS + lane] = v_;                                           \
            acc1[0] = acc1[1] = acc1[2] = acc1[3] = 0.f;                                         \
        } else {                                                                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                     \
                if (4 * o + i_ < a.M) rp_[wave * RS + (4 * o + i_) * 16 + r] = accm[i_];         \
            accm = (f4){0.f, 0.f, 0.f, 0.f};                                                     \
        }                                                                                        \
        __syncthreads();                                                                         \
        const int rt_ = rt0 + cj;                                                                \
        if (e_on) {                                           /* M * 16 <= 256 <= threads */     \
            float tot_ = 0.f;                                                                    \
            _Pragma("unroll") for (int w_ = 0; w_ < NW; ++w_) tot_ += rp_[w_ * RS + threadIdx.x];  \
            _Float16 y_ = (_Float16)tot_;                     /* fp16(matmul) */                 \
            if (so.bias) y_ = y_ + pf_bias;                   /* out + bias      (fp16 add) */   \
            if (so.residual) y_ = pf_res + y_;                /* residual + out */               \
            if (SC1) __hip_atomic_store((unsigned short*)(so.y + (size_t)e_m * so.y_stride + rt_ * 16 + e_c),          \
                                        __builtin_bit_cast(unsigned short, y_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            else so.y[(size_t)e_m * so.y_stride + rt_ * 16 + e_c] = y_;                          \
        }                                                                                        \
        if (cj + 1 < n_my) AMQ_EPI_PREFETCH(rt_ + 1);                                            \
        par ^= 1;                                                                                \
    } while (0)

    // consume one tile out of ring slot `slot`
#ifdef AMQ_ABL_NOCOMPUTE   /* ablation build: keep the loads and the row-tile protocol, drop unpack / LDS reads / MFMA */
#define AMQ_COMPUTE(slot)                                                                        \
    do {                                                                                         \
        uint32_t x_ = 0;                                                                         \
        _Pragma("unroll") for (int d_ = 0; d_ < BITS; ++d_) x_ ^= pay[slot].w[d_];               \
        accm[0] += (float)(x_ & 1u) + (float)meta[slot].x;                                       \
    } while (0)
#else
#define AMQ_COMPUTE(slot)                                                                        \
    do {                                                                                         \
        const int g_ = wave + ci * NW;                                                           \
        const int kbase = g_ << 7;                                                               \
        h2 wv[16];                                                                               \
        if (MATH == MATH_LINEAR) unpack_lane_sub<BITS>(pay[slot].w, wv);                         \
        else if constexpr (GP == 1 && MODE == MODE_FMA1) dequant_lane_fma1<BITS>(pay[slot].w, meta[slot], wv);   /* reference-format weights, one op per pair */ \
        else if constexpr (GP == 1) dequant_lane_sd<BITS, MODE>(pay[slot].w, meta[slot], wv);    \
        else dequant_lane_sd_g<BITS, MODE, GP>(pay[slot].w, metag[slot], wv);                    \
        if (MATH == MATH_DOT) {                                                                  \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                      \
                const h8 xv = *(const h8*)(xl + kbase + 8 * o + 32 * t);                         \
                _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                  \
                    h2 xp = {xv[2 * p], xv[2 * p + 1]};                                          \
                    acc1[t] = __builtin_amdgcn_fdot2(wv[4 * t + p], xp, acc1[t], false);         \
                }                                                                                \
            }                                                                                    \
        } else {                                                                                 \
            f4 c_ = (MATH == MATH_LINEAR) ? (f4){0.f, 0.f, 0.f, 0.f} : accm;                     \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                      \
                h8 b;                                                                            \
                _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                  \
                    b[2 * p] = wv[4 * t + p].x; b[2 * p + 1] = wv[4 * t + p].y;                  \
                }                                                                                \
                const h8 av = AMQ_XREAD(xrow + kbase + 32 * t);                                  \
                c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b, c_, 0, 0, 0);                 \
            }                                                                                    \
            if (MATH == MATH_LINEAR) {                                                           \
                /* rows m = 4*o + i of this lane's column r: y += s*(2^(24-SH)*S - z*X_g)  (HQQ) */ \
                /*                                            y += s*2^(24-SH)*S + c*X_g    (FMA) */ \
                const float sf = (float)meta[slot].x, zf = (float)meta[slot].y;                  \
                const float s24 = sf * (float)(1 << (24 - LinCfg<BITS>::SH));                    \
                const float zx = (MODE == MODE_HQQ) ? -(sf * zf) : zf;                           \
                const f4 xs4 = *(const f4*)(xg + g_ * 16 + 4 * o);                               \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                 \
                    accm[i_] = __builtin_fmaf(s24, c_[i_], __builtin_fmaf(zx, xs4[i_], accm[i_])); \
            } else {                                                                             \
                accm = c_;                                                                       \
            }                                                                                    \
        }                                                                                        \
    } while (0)
#endif
    // end of this wave's share of a row-tile?  (runs AFTER the slot has been re-issued: the ring stays full
    // while the wave sits in the row-tile barrier)
#ifdef AMQ_ABL_NOFINISH    /* ablation: no row-tile barrier / reduction / store (timing only, results wrong) */
#define AMQ_ROWEND() do { if (++ci == nt) { ci = 0; ++cj; } } while (0)
#else
#define AMQ_ROWEND()                                                                             \
    do {                                                                                         \
        if (++ci == nt) { AMQ_FINISH(); ci = 0; ++cj; }                                          \
    } while (0)
#endif

    if (nt == 0) {                                                // K < 128 * NW: this wave owns no tile
        for (int j = 0; j < n_my; ++j) { AMQ_FINISH(); ++cj; }
        return;
    }

    // Software pipeline over this wave's tile stream.  A slot is refilled only after its
    // tile has been fully consumed (sched_barrier keeps the compiler from hoisting the load
    // into temporaries + a vmcnt(0)/v_mov rotation -- what a naive ring compiles to); the
    // main loop refills unconditionally so its waits stay counted, the tail drains.
    int idx = 0;
#ifdef AMQ_STAMP
    // diagnostic build: shader-clock cycles this wave spends (a) waiting for its next tile, (b) in unpack + MFMA,
    // (c) in the row-tile epilogue incl. its barrier.  The explicit wait is the one the compiler would insert itself.
    unsigned long long c_wait = 0, c_math = 0, c_row = 0;
    constexpr int OPS_PER_TILE = (BITS == 3) ? 4 : 2;
#define AMQ_T() __builtin_amdgcn_s_memtime()
    AMQ_STAMP_AT(blk, 96 + wave);                                  // realtime: about to wait for the first tile
    bool first_ = true;
    for (; idx + 2 * U <= total; idx += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long t0 = AMQ_T();
            __builtin_amdgcn_s_waitcnt(0x0F70 | ((U - 1) * OPS_PER_TILE));      /* vmcnt((U-1)*ops) [hi bits 0], lgkm/exp untouched */
            if (first_) { AMQ_STAMP_AT(blk, 112 + wave); first_ = false; }       // realtime: first tile has arrived
            const unsigned long long t1 = AMQ_T();
            AMQ_COMPUTE(u);
            __builtin_amdgcn_sched_barrier(0);
            AMQ_ISSUE(u);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t2 = AMQ_T();
            AMQ_ROWEND();
            const unsigned long long t3 = AMQ_T();
            c_wait += t1 - t0; c_math += t2 - t1; c_row += t3 - t2;
        }
    }
    if (blk.stamps && lane == 0) {
        blk.stamps[(size_t)blockIdx.x * 128 + 32 + wave * 4 + 0] = c_wait;
        blk.stamps[(size_t)blockIdx.x * 128 + 32 + wave * 4 + 1] = c_math;
        blk.stamps[(size_t)blockIdx.x * 128 + 32 + wave * 4 + 2] = c_row;
        blk.stamps[(size_t)blockIdx.x * 128 + 32 + wave * 4 + 3] = (unsigned long long)idx;
    }
#else
#ifndef AMQ_NO_PRIO_PROGRESS
    // Issue priority falls with the wave's progress (quartiles of its tile count).  The SIMD arbiter is oldest-first: a CU's
    // three workgroups -- and the waves of one workgroup that share a SIMD -- otherwise complete one after the other, and
    // the SIMD runs at its two-wave efficiency (335-394 cycles per tile) instead of the six-wave one (242); with laggards
    // preferred all resident waves stay interleaved (profiles/r01b_gemv_prio.txt: 2-10% per launch).
    const int q1_ = total >> 2, q2_ = total >> 1, q3_ = q1_ + q2_;
#endif
    for (; idx + 2 * U <= total; idx += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            AMQ_COMPUTE(u);
            __builtin_amdgcn_sched_barrier(0);
            AMQ_ISSUE(u);
            __builtin_amdgcn_sched_barrier(0);
            AMQ_ROWEND();
        }
#ifndef AMQ_NO_PRIO_PROGRESS
        plevel_ = idx + U >= q3_ ? 0 : idx + U >= q2_ ? 1 : idx + U >= q1_ ? 2 : 3;
        AMQ_SETPRIO_LEVEL();
#endif
    }
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (idx + u < total) {
            AMQ_COMPUTE(u);
            __builtin_amdgcn_sched_barrier(0);
            if (idx + u + U < total) AMQ_ISSUE(u);
            AMQ_ROWEND();
        }
    }
    idx += U;
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (idx + u < total) { AMQ_COMPUTE(u); AMQ_ROWEND(); }
#ifdef AMQ_ABL_NOFINISH
    cj = 0;
    AMQ_FINISH();                                                 // keep the accumulators alive: one epilogue at the end
#endif
    AMQ_STAMP_AT(blk, 8 + wave);                                    // per-wave end of stream
#undef AMQ_ISSUE
#undef AMQ_ISSUE_AT
#undef AMQ_COMPUTE
#undef AMQ_ROWEND
#undef AMQ_EPI_PREFETCH
#undef AMQ_SETPRIO_LEVEL
#undef AMQ_FINISH
}

// minimum waves per SIMD for the register allocator (A/B builds only).  The product kernels need <= 80 VGPRs so that three
// 8-wave workgroups fit a CU (6 waves per SIMD); they get there without a bound (76-78) by holding one activation chunk
// per thread (XCfg).  Forcing 64 (8 waves per SIMD) spills 27-38 registers into the main loop.
#ifndef AMQ_LB_WAVES
#define AMQ_LB_WAVES(NW_) 1
#endif
// Kernel-argument preload (gfx950): the first 14 dwords of explicit arguments are delivered in SGPRs by the command
// processor at wave launch (`-mllvm -amdgpu-kernarg-preload-count=14`, csrc/Makefile), so they cost no memory round
// trip.  They carry everything a single-segment launch -- and segment 0 of a grouped one -- needs to issue its
// activation and weight loads; the other segments take ONE clause of static-offset scalar loads from the block.
struct GemvPre {            // not a kernel parameter type: just names the 14 dwords
    const void* x; const void* xw; const void* qw0; const void* mt0;
    int K, m_nseg, rpt, n_rt0, key0; float eps;
};

// (groups of 64 on 8-wave workgroups: the allocator is held to the three-workgroups-per-CU budget, 80 VGPRs -- left alone it takes 90-96 for the second
//  meta pair and the launch runs at five waves per SIMD: 7B avg-3 decode 732 -> 773 tokens/s with the bound (5 spilled dwords); at groups of 32 the
//  bound costs more than the sixth wave brings, 709 -> 689, so those keep the free allocation.  profiles/r04_fine_groups.txt)
#ifndef AMQ_LB_WAVES_G
#define AMQ_LB_WAVES_G(NW_, GP_) (((GP_) == 2 && (NW_) == 8) ? 6 : AMQ_LB_WAVES(NW_))
#endif
template <int PRO, int NW, int U, int MATH, int XCH, int RS = 256, int GP = 1>
__global__ __launch_bounds__(NW * 64, AMQ_LB_WAVES_G(NW, GP)) void gemv_kernel(const void* p_x, const void* p_xw, const void* p_qw0,
                                                                     const void* p_mt0, int p_K, int p_m_nseg, int p_rpt,
                                                                     int p_n_rt0, int p_key0, float p_eps, GemvKArgs blk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifndef AMQ_NO_PRIO_PROGRESS
    __builtin_amdgcn_s_setprio(3);          // prologue and first quarter of the stream at top priority (see the main loop)
#endif
    GemvHot a;
    a.x = p_x; a.x2 = p_xw; a.gamma = p_xw;
    a.K = p_K; a.M = p_m_nseg & 0xFFFF; a.rpt = p_rpt; a.eps = p_eps;
    const int nseg = p_m_nseg >> 16;
    const bool slow_x = a.M != 1 || (a.K >> 3) > XCH * NW * 64;   // generic staging path
    a.x_stride = slow_x ? blk.x_stride : a.K;
    const int xs = a.K + XPAD;
    _Float16* xl = (_Float16*)smem;
    const size_t xbytes = ((size_t)a.M * xs * 2 + 15) & ~(size_t)15;
    float* xg = (float*)(smem + xbytes);                                        // [G][16] (linear math only)
    const size_t xgbytes = (MATH == MATH_LINEAR) ? (size_t)(a.K >> 7) * 64 : 0;
    float* red = (float*)(smem + xbytes + xgbytes);                             // [2][NW][16][16]

    // decode fast path: one activation row whose chunks fit the per-thread registers -> its loads leave FIRST, before the per-segment arguments'
    // kernarg round trip below (everything they need arrives preloaded): staging x -- arrival, norm, LDS, barrier -- is the critical path of a
    // launch's prologue (issuing the first weight tile ahead of them instead measured 2-3 % slower)
    XRegs xr;
    const bool fastx = !slow_x;
    if (fastx) x_issue<PRO, NW, XCH>(a, xr);

    const int bid = (int)blockIdx.x;
    int sidx = 0;
    int wgb = 0, nrt = p_n_rt0, key = p_key0;
    const void* qwp = p_qw0;
    const void* mtp = p_mt0;
    if (nseg > 1) {
        // All hot per-segment arguments are forced into SGPRs here, by one clause of scalar loads and a single wait: left
        // to itself the compiler sinks each s_load next to its first use, which makes 3-4 DEPENDENT kernarg round trips
        // (cold at every launch, ~0.3 us each) before the first weight load can be issued.
        asm volatile("" ::"s"(blk.wg_begin[1]), "s"(blk.wg_begin[2]), "s"(blk.wg_begin[3]), "s"(blk.n_rt[1]),
                     "s"(blk.n_rt[2]), "s"(blk.n_rt[3]), "s"(blk.key[1]), "s"(blk.key[2]), "s"(blk.key[3]),
                     "s"(blk.qweight[1]), "s"(blk.qweight[2]), "s"(blk.qweight[3]), "s"(blk.meta[1]), "s"(blk.meta[2]),
                     "s"(blk.meta[3]));
#pragma unroll
        for (int i = 1; i < GEMV_MAX_SEG; ++i) {
            const bool take = i < nseg && bid >= blk.wg_begin[i];
            sidx = take ? i : sidx;
            wgb = take ? blk.wg_begin[i] : wgb;
            nrt = take ? blk.n_rt[i] : nrt;
            key = take ? blk.key[i] : key;
            qwp = take ? blk.qweight[i] : qwp;
            mtp = take ? blk.meta[i] : mtp;
        }
    }
    const int local = bid - wgb;
#ifdef AMQ_STAMP
    if (threadIdx.x == 0 && blk.stamps) {
        blk.stamps[(size_t)blockIdx.x * 128 + 0] = __builtin_amdgcn_s_memrealtime();
        blk.stamps[(size_t)blockIdx.x * 128 + 5] = __builtin_amdgcn_s_memtime();
        blk.stamps[(size_t)blockIdx.x * 128 + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                                (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
    }
#endif

    const _Float16* xuse = xl;
    constexpr bool HAS_FMA1 = MATH == MATH_EXACT && GP == 1;      // (launch_gemv maps MODE_FMA1 to MODE_FMA for the kernels without those bodies)
    bool done = false;
    if constexpr (HAS_FMA1) {
        if (key == 4 * 4 + MODE_FMA1) { gemv_body<4, MODE_FMA1, PRO, NW, U, MATH, XCH, RS, false, GP>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, fastx, xr); done = true; }
        else if (key == 3 * 4 + MODE_FMA1) { gemv_body<3, MODE_FMA1, PRO, NW, U, MATH, XCH, RS, false, GP>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, fastx, xr); done = true; }
        else if (key == 2 * 4 + MODE_FMA1) { gemv_body<2, MODE_FMA1, PRO, NW, U, MATH, XCH, RS, false, GP>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, fastx, xr); done = true; }
    }
    if (!done) switch (key) {
        case 4 * 4 + MODE_HQQ: gemv_body<4, MODE_HQQ, PRO, NW, U, MATH, XCH, RS, false, GP>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, fastx, xr); break;
        case 3 * 4 + MODE_HQQ: gemv_body<3, MODE_HQQ, PRO, NW, U, MATH, XCH, RS, false, GP>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, fastx, xr); break;
        case 2 * 4 + MODE_HQQ: gemv_body<2, MODE_HQQ, PRO, NW, U, MATH, XCH, RS, false, GP>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, fastx, xr); break;
        case 4 * 4 + MODE_FMA: gemv_body<4, MODE_FMA, PRO, NW, U, MATH, XCH, RS, false, GP>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, fastx, xr); break;
        case 3 * 4 + MODE_FMA: gemv_body<3, MODE_FMA, PRO, NW, U, MATH, XCH, RS, false, GP>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, fastx, xr); break;
        default:               gemv_body<2, MODE_FMA, PRO, NW, U, MATH, XCH, RS, false, GP>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, fastx, xr); break;
    }
    if (threadIdx.x < 64) AMQ_STAMP_AT(blk, 4);
#ifdef AMQ_STAMP
    if (threadIdx.x == 0 && blk.stamps) blk.stamps[(size_t)blockIdx.x * 128 + 6] = __builtin_amdgcn_s_memtime();
#endif
}

#ifdef AMQ_AB_ROUTES     /* an A/B route (slower than the two launches it replaces): compiled into libamq_hip_ab.so only -- make ab */
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
        case 4 * 4 + MODE_HQQ: gemv_body<4, MODE_HQQ, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, true, xr); break;
        case 3 * 4 + MODE_HQQ: gemv_body<3, MODE_HQQ, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, true, xr); break;
        case 2 * 4 + MODE_HQQ: gemv_body<2, MODE_HQQ, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, true, xr); break;
        case 4 * 4 + MODE_FMA: gemv_body<4, MODE_FMA, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, true, xr); break;
        case 3 * 4 + MODE_FMA: gemv_body<3, MODE_FMA, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, true, xr); break;
        default:               gemv_body<2, MODE_FMA, PRO, NW, U, MATH, XCH, 256, true>(a, blk, sidx, qwp, mtp, nrt, local, xl, xl, xg, red, xs, true, xr); break;
    }
    // ---- publish this workgroup's row-tiles: drain (the storing threads sit in wave 0), barrier, one lane adds to the tickets
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int rt0 = local * a.rpt;
    const int n_my = (nrt - rt0) < a.rpt ? (nrt - rt0) : a.rpt;
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
        k.wg_begin[i] = s.wg_begin; k.n_rt[i] = s.n_rt; k.key[i] = s.bits * 4 + (s.mode == MODE_FMA1 ? (int)MODE_FMA : s.mode);
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

#endif  // AMQ_AB_ROUTES

#ifdef AMQ_STAMP
unsigned long long* g_stamp_ptr = nullptr;
extern "C" int amq_debug_set_stamps(void* p) { g_stamp_ptr = (unsigned long long*)p; return 0; }
#endif

// nw: waves per workgroup the launch will use (sizes the double-buffered cross-wave sum); <= 1: the largest (16), a safe bound
size_t gemv_lds_bytes(int M, int K, int nw) {
    const size_t xbytes = (((size_t)M * (K + XPAD) * 2) + 15) & ~(size_t)15;
    const size_t xg = (size_t)(K >> 7) * 64;
    const size_t red = (size_t)2 * (nw > 1 ? nw : 16) * 16 * 16 * 4;
    return xbytes + xg + red;
}

int gemv_pick_waves(int total_rt, int K) {
    // measured on MI355X (tools/sweep1.sh, profiles/r01b_gemv_sweep.txt): 8 waves x 2 tiles in flight is the best or
    // within 3% of it whenever a wave gets >= 4 tiles of a row-tile or there is more than one workgroup per CU; one
    // 16-wave workgroup per CU wins (3-6%) only for long rows (K >= 8192: down_proj) on <= 256 row-tiles; for K = 4096
    // on 256 row-tiles (o_proj) 8 waves are ~10% faster (16-wave workgroups dispatch later); tiny K: 4 waves.
    // K > 4096 always takes 16 waves: the register-held activation staging of the decode prologue covers
    // K <= 8 * XC * threads (4096 for 8 waves, 16384 for 16), and one 16-wave workgroup per CU is within 5% of the
    // 8-wave grids on every shape measured.  (Exception made in launch_gemv: single-row launches with 4096 < K <= 8192
    // go to 8-wave workgroups that stage two chunks per thread, two workgroups per CU.)
    const int G = K >> 7;
    if (G < 16) return 4;
    if (G > 32) return 16;
    return 8;
}

template <int PRO, int NW, int U, int MATH, int XCH = XCfg<NW>::XC, int RS = 256, int GP = 1>
static hipError_t launch_one(const GemvKArgs& a, int total_wg, size_t lds, hipStream_t st) {
    auto kern = gemv_kernel<PRO, NW, U, MATH, XCH, RS, GP>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const void* xw = PRO == PRO_SILU_MUL ? a.x2 : a.gamma;
    hipLaunchKernelGGL(kern, dim3(total_wg), dim3(NW * 64), lds, st, a.x, xw, a.qweight[0], a.meta[0], a.K,
                       a.M | (a.nseg << 16), a.rpt, a.n_rt[0], a.key[0], a.eps, a);
    return hipGetLastError();
}

template <int PRO, int NW>
static hipError_t launch_nw(const GemvKArgs& a, int flags, int depth, int total_wg, size_t lds, hipStream_t st) {
    const int u = depth ? depth : 2;
    if ((flags & GEMV_FLAG_DOT) && a.M == 1) return launch_one<PRO, NW, 2, MATH_DOT>(a, total_wg, lds, st);
    if (flags & GEMV_FLAG_LINEAR) {
        if (u == 4) return launch_one<PRO, NW, 4, MATH_LINEAR>(a, total_wg, lds, st);
        return launch_one<PRO, NW, 2, MATH_LINEAR>(a, total_wg, lds, st);
    }
    if (u == 4) return launch_one<PRO, NW, 4, MATH_EXACT>(a, total_wg, lds, st);
    if (NW == 8 && (flags & GEMV_FLAG_RS128)) return launch_one<PRO, 8, 2, MATH_EXACT, XCfg<8>::XC, 128>(a, total_wg, lds, st);
    if (NW == 16 && a.M == 1 && (a.K >> 3) > XCfg<16>::XC * 1024 && (a.K >> 3) <= 4 * 1024)
        return launch_one<PRO, 16, 2, MATH_EXACT, 4>(a, total_wg, lds, st);        // 16384 < K <= 32768 (70B down_proj)
    if (NW == 8 && a.M == 1 && (a.K >> 3) > 512 && (a.K >> 3) <= 1024)
        return launch_one<PRO, 8, 2, MATH_EXACT, 2>(a, total_wg, lds, st);         // 4096 < K <= 8192 on 8 waves (two x chunks per thread)
    return launch_one<PRO, NW, 2, MATH_EXACT>(a, total_wg, lds, st);
}

// groups of 64 / 32 (GP = 2 / 4 meta pairs per tile): the default geometry of the exact-math body (two tile loads in flight, generic x staging
// beyond its register-held chunks)
template <int PRO, int GP>
static hipError_t launch_pro_g(const GemvKArgs& a, int nw, int total_wg, size_t lds, hipStream_t st) {
    if (nw == 4) return launch_one<PRO, 4, 2, MATH_EXACT, XCfg<4>::XC, 256, GP>(a, total_wg, lds, st);
    if (nw == 16) return launch_one<PRO, 16, 2, MATH_EXACT, XCfg<16>::XC, 256, GP>(a, total_wg, lds, st);
    return launch_one<PRO, 8, 2, MATH_EXACT, XCfg<8>::XC, 256, GP>(a, total_wg, lds, st);
}

template <int PRO>
static hipError_t launch_pro(const GemvKArgs& a, int flags, int depth, int nw, int total_wg, size_t lds, hipStream_t st) {
    if (nw == 4) return launch_nw<PRO, 4>(a, flags, depth, total_wg, lds, st);
    if (nw == 16) return launch_nw<PRO, 16>(a, flags, depth, total_wg, lds, st);
    return launch_nw<PRO, 8>(a, flags, depth, total_wg, lds, st);
}

// Fills the per-segment workgroup ranges and launches.  rpt = row-tiles per workgroup.
hipError_t launch_gemv(GemvArgs& a, hipStream_t st) {
    StreamDevice sd_(st);                                  // kernel attributes are per device: the stream's, not the current one
    int total_rt = 0;
    for (int i = 0; i < a.nseg; ++i) { a.seg[i].n_rt = a.seg[i].N / 16; total_rt += a.seg[i].n_rt; }
    int nw = a.force_waves ? a.force_waves : gemv_pick_waves(total_rt, a.K);
    // several rows (batched decode): the staged x grows with M, and once fewer than three 8-wave workgroups fit a CU's LDS one
    // 16-wave workgroup keeps more waves on the weight stream (7B, 8 sequences: 2.76 -> 2.15 ms a step; 4 sequences still fit three)
    // with at most 8 rows the cross-wave sum needs half its buffer (RS = 128): at 5 rows three 8-wave workgroups fit a CU again
    // (7B: 2.20 -> 2.01 ms a step); TWO 8-wave workgroups (6 - 8 rows) measured slower than one 16-wave workgroup (2.28 / 2.34 vs
    // 2.10 / 2.19 ms), so those keep the 16-wave form (profiles/r02_decode_batch.txt)
    const int gp = a.gp > 1 ? a.gp : 1;                    // meta pairs per tile (groups of 64 / 32: 2 / 4)
    if (gp != 1 && gp != 2 && gp != 4) return hipErrorInvalidValue;
    if (gp > 1 && ((a.flags & (GEMV_FLAG_DOT | GEMV_FLAG_LINEAR)) || a.force_depth == 4)) return hipErrorInvalidValue;   // (the C ABI says so first)
    bool rs128 = false;
    if (!a.force_waves && a.M > 1 && nw == 8) {
        const bool plain = !(a.flags & (GEMV_FLAG_DOT | GEMV_FLAG_LINEAR)) && a.force_depth != 4 && gp == 1;
        const size_t lds128 = gemv_lds_bytes(a.M, a.K, 8) - 2 * 8 * 128 * 4;
        if (plain && a.M <= 8 && 3 * lds128 <= 160 * 1024) rs128 = true;
        else if (3 * gemv_lds_bytes(a.M, a.K, 8) > 160 * 1024) nw = 16;
    }
    // 4096 < K <= 8192 at one row (13B / 70B hidden sizes): 8-wave workgroups staging two x chunks per thread, two per CU
    // (~90 VGPRs), instead of one 16-wave workgroup -- 13B 464 -> 485 tokens/s, 70B 126.5 -> 133.  The same trade for
    // 8192 < K <= 16384 (four chunks per thread) loses (13B 484 -> 467; 7B's K = 11008 with three chunks 808 -> 773), as do
    // three workgroups per CU (449 / 122).
    const bool mid_k = gp == 1 && !a.force_waves && nw == 16 && a.M == 1 && (a.K >> 3) > 512 && (a.K >> 3) <= 1024 &&
                       !(a.flags & (GEMV_FLAG_DOT | GEMV_FLAG_LINEAR)) && a.force_depth != 4;    // (only the exact-math body has the two-chunk variant)
    if (mid_k) nw = 8;
    // persistent-style grid: about 24 waves per CU (256 CUs) -- three 8-wave workgroups, but ONE 16-wave workgroup (two do
    // not fit the register file at 78 VGPRs, a second round of workgroups would run on an empty chip); a workgroup walks
    // `rpt` row-tiles
    int rpt = a.force_rpt;
    if (rpt <= 0) {
        const int target = mid_k ? 512 : nw == 16 ? 256 : 256 * 24 / nw;
        rpt = (total_rt + target - 1) / target;
        if (rpt < 1) rpt = 1;
    }
    int wg = 0, mask = 0;
    for (int i = 0; i < a.nseg; ++i) {
        a.seg[i].wg_begin = wg;
        a.seg[i].wg_count = (a.seg[i].n_rt + rpt - 1) / rpt;
        wg += a.seg[i].wg_count;
        mask |= 1 << a.seg[i].bits;
    }
    const bool lin = (a.flags & GEMV_FLAG_LINEAR) && !((a.flags & GEMV_FLAG_DOT) && a.M == 1);
    (void)lin; (void)mask;
    const size_t lds = gemv_lds_bytes(a.M, a.K, a.M > 1 ? nw : 16) - (rs128 ? 2 * 8 * 128 * 4 : 0);     // (one-row launches keep the allocation they were tuned with)
    GemvKArgs k{};
    k.x = a.x; k.x2 = a.x2; k.gamma = a.gamma;
    k.M = a.M; k.K = a.K; k.x_stride = a.x_stride; k.nseg = a.nseg;
    k.eps = a.eps; k.rpt = rpt;
    for (int i = 0; i < a.nseg; ++i) {
        const GemvSeg& s = a.seg[i];
        // (the one-op unpack of MODE_FMA1 exists in the exact-math, group-128 bodies; elsewhere such buffers run as MODE_FMA: same results)
        const bool has_fma1 = gp == 1 && !(a.flags & (GEMV_FLAG_DOT | GEMV_FLAG_LINEAR));
        k.wg_begin[i] = s.wg_begin; k.n_rt[i] = s.n_rt; k.key[i] = s.bits * 4 + (s.mode == MODE_FMA1 && !has_fma1 ? (int)MODE_FMA : s.mode);
        k.qweight[i] = s.qweight; k.meta[i] = s.meta;
        k.bias[i] = s.bias; k.residual[i] = s.residual; k.y[i] = s.y; k.y_stride[i] = s.y_stride;
    }
#ifdef AMQ_STAMP
    k.stamps = g_stamp_ptr;
#endif
    if (gp == 2) {
        switch (a.prologue) {
            case PRO_NONE: return launch_pro_g<PRO_NONE, 2>(k, nw, wg, lds, st);
            case PRO_RMSNORM: return launch_pro_g<PRO_RMSNORM, 2>(k, nw, wg, lds, st);
            default: return launch_pro_g<PRO_SILU_MUL, 2>(k, nw, wg, lds, st);
        }
    }
    if (gp == 4) {
        switch (a.prologue) {
            case PRO_NONE: return launch_pro_g<PRO_NONE, 4>(k, nw, wg, lds, st);
            case PRO_RMSNORM: return launch_pro_g<PRO_RMSNORM, 4>(k, nw, wg, lds, st);
            default: return launch_pro_g<PRO_SILU_MUL, 4>(k, nw, wg, lds, st);
        }
    }
    switch (a.prologue) {
        case PRO_NONE: return launch_pro<PRO_NONE>(k, a.flags | (rs128 ? GEMV_FLAG_RS128 : 0), a.force_depth, nw, wg, lds, st);
        case PRO_RMSNORM: return launch_pro<PRO_RMSNORM>(k, a.flags | (rs128 ? GEMV_FLAG_RS128 : 0), a.force_depth, nw, wg, lds, st);
        default: return launch_pro<PRO_SILU_MUL>(k, a.flags | (rs128 ? GEMV_FLAG_RS128 : 0), a.force_depth, nw, wg, lds, st);
    }
}

}  // namespace amq
