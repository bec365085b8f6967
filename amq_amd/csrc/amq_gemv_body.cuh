// amq_gemv_body.cuh -- weight-streaming y = x . W^T for few rows (decode), gfx950: the kernel family and its launch templates.
// Included by amq_gemv_pro{0,1,2}.hip / amq_gemv_fine.hip (one translation unit per fused prologue, so that the ~90 kernel bodies compile in
// parallel) and by the A/B route amq_gemv_qkvattn.hip; the host-side launch plan is amq_gemv.hip.
//
// Replaces, for rows < 8/128, the reference's
//   VecQuant{2,3,4}MatMulKernelFaster_old (amq/kernel/AutoGPTQ/auto_gptq_kernel.cu:160-440)
//   gemv_kernel<2,Batch,256,128>          (amq/kernel/ft/quantization_new/gemv/gemv_cuda.cu:73-204)
// with one kernel family over the native AMQ-T16 layout (amq_common.cuh).
//
// Structure (every byte of W is read exactly once; see DESIGN.md 3.2 and HISTORY.md for the measurements behind each choice):
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
//     MATH_GS     (opt-in, AMQ_MATH_GROUPSCALE): the first rounding exactly as MATH_EXACT at E = -9 -- a field is used where the packing left
//                 it, one shift per dword -- the scale applied once per (row, group) in fp32 after the tile's four MFMAs; 6.5 cycles per
//                 pair, ~3.2e-4 of the output rms from the reference result (amq_common.cuh dequant_lane_gs; one-rounding modes and the
//                 finer groups run their exact bodies under it).
//     MATH_LINEAR (opt-in): every field is shifted to one mantissa position and fed to the MFMA as the fp16
//                 subnormal q*2^(SH-24) (gfx950 MFMA honours fp16 subnormals -- measured); scale / zero are applied
//                 per group in fp32:  y += s*(2^(24-SH)*sum(x q) - z*sum_g(x)), with the group sums of x taken once in
//                 the staging pass.  4 VALU cycles per pair; results are the real-valued dequant (no per-weight fp16
//                 roundings), ~3e-4 of the output rms away from the reference's rounded-weight result.
//   * x staging: one row -- its loads leave first and are held in registers across the ring's priming (x_issue / x_finish); 2 .. 8 rows -- the
//     rows go straight into LDS by LDS-DMA ahead of the ring, transform applied in place (x_dma_rows / x_finish_dma; kernels RS = 64 / 128; PH = 2:
//     two K phases for rows that do not fit LDS whole); otherwise the generic stage_x.
//   * per row-tile, fixed-order cross-wave sum through double-buffered LDS and one
//     barrier: deterministic, no atomics.
//   * several linears that share x (q/k/v, gate/up) with different bit-widths
//     run as segments of ONE launch; a workgroup serves one segment.
#pragma once
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
    int n_rt[GEMV_MAX_SEG];                // gemv_split(row-tiles, workgroups) of the segment
    int key[GEMV_MAX_SEG];                 // bits * 4 + mode
    const void* qweight[GEMV_MAX_SEG];
    const void* meta[GEMV_MAX_SEG];
    // ---- cold: epilogue only
    const void* bias[GEMV_MAX_SEG];
    const void* residual[GEMV_MAX_SEG];
    void* y[GEMV_MAX_SEG];
    int y_stride[GEMV_MAX_SEG];
    float* sums_out;                       // (5 .. 8-row kernels, one segment) per-row-tile sums of squares of the rows written: [M][sums_stride], or null
    int sums_stride;                       // = N / 16
#ifdef AMQ_STAMP
    unsigned long long* stamps;
#endif
};

struct SegOut { const _Float16* bias; const _Float16* residual; _Float16* y; int y_stride; float* sums; int sums_stride; };

// sum over the 16 lanes of a DPP row (every lane of the row gets it; fixed order)
__device__ __forceinline__ float row16_total(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}

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
enum { MATH_EXACT = 0, MATH_DOT = 1, MATH_LINEAR = 2, MATH_GS = 3 };
// A/B build knob (-DAMQ_SD_E9=1): the exact two-rounding unpack over the group-scale body's field positions (E = -9: one shift per dword instead of one per
// pair or three per dword).  Measured: 7B 851 -> 866 tokens/s (+1.7 %), 70B 143.1 -> 146.9 (+2.6 %).  Not taken: the first rounding is then exact only for
// |q - z|, |z| >= 2^-5 (now 2^-9 .. 2^-11) -- ~0.6 % of weights (those within 3 % of a step of their zero) would differ from the oracle's by up to 2^-15 of a
// step, and the GEMV would stop being bit-identical to the GEMM families and to dequantize + fp16 GEMM on the same inputs (HISTORY.md R5).
#ifndef AMQ_SD_E9
#define AMQ_SD_E9 0
#endif

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

// mean of squares = tot / K, as LlamaRMSNorm's .mean(): a true division -- except that for K a power of two (hidden sizes 4096, 8192) the product with
// the exact reciprocal IS that quotient, bit for bit, at one instruction instead of ten per row (every workgroup of a fused-prologue launch pays them)
struct MeanDiv {
    float inv; bool pow2; float k;
    __device__ __forceinline__ explicit MeanDiv(int K)
        : inv(__builtin_bit_cast(float, (127 - __builtin_ctz((unsigned)K)) << 23)), pow2((K & (K - 1)) == 0), k((float)K) {}      // (inv = 2^-log2(K): exact when pow2, unused otherwise)
    __device__ __forceinline__ float operator()(float tot) const { return pow2 ? tot * inv : tot / k; }
};

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
                rstd = rsqrtf(MeanDiv(K)(tot) + a.eps);
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
            rstd = rsqrtf(MeanDiv(K)(tot) + a.eps);
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
constexpr int SUMS_PER_LANE = 8;     // PRO_RMSNORM_SUMS: partial sums of squares a lane fetches for its wave's row (K / 16 <= 512 partials: K <= 8192)
struct XRegs { union { h8 v[XC_MAX]; float f[4 * XC_MAX]; }; h8 w[XC_MAX]; };   // w: up (SiLU*mul) or gamma (RMSNorm); f: the row's partial sums of squares (PRO_RMSNORM_SUMS)
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
        rstd = rsqrtf(MeanDiv(a.K)(tot) + a.eps);
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

// rows a kernel's cross-wave sum buffer (RS floats per wave = rows x 16 columns) is laid out for: 256 -> any M <= 16 (generic staging beyond one
// row), 128 -> the 5 .. 8-row kernels, 64 -> the 2 .. 4-row kernels (the last two take their rows by LDS-DMA)
template <int RS> struct RowsCfg { static constexpr int MRMAX = RS == 64 ? 4 : RS == 128 ? 8 : 1; };

// Several rows (2 .. 8 sequences decoded together): the rows go from global memory STRAIGHT INTO LDS (global_load_lds_dwordx4, 1 KB per wave
// instruction, no registers), issued before the weight ring is primed like the one-row path's loads -- all of a launch's activation reads are
// independent and the oldest entries of the vector-memory queue.  After the ring is primed a wave waits for exactly its own transfers (counted
// vmcnt: the ring's loads are the only younger operations), reads ITS chunks back from LDS -- thread t owns column chunk t (+ i * THREADS) of every
// row, the bytes its own wave transferred, so no barrier is needed before the read-back -- applies the fused transform in place and the caller's
// barrier publishes x.  (The generic stage_x walks its chunks in dependent load -> use iterations BEHIND the primed ring and passes over x twice:
// +2.6 / +4.5 / +9 us per launch at 2 / 4 / 8 rows; holding the rows in registers instead costs 16 - 64 VGPRs across the ring's priming and pushes
// the kernel out of its three-workgroups-per-CU budget: profiles/r05_decode_batch.txt.)  Per row the sums run in the one-row path's order: a
// sequence's RMSNorm is the same bits alone or in a batch.  SiLU*mul: the gate rows are transferred, the up rows are plain loads issued behind the
// primed ring (they arrive with the first weight tiles) four rows at a time.
// (inline asm, not __builtin_amdgcn_global_load_lds: see amq_gemm_ring.hip)
__device__ __forceinline__ void gv_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" AMQ_MARK("gemv.xdma") :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// (k0, Ks: the columns staged -- all of K, or one K phase of a launch whose rows do not fit LDS whole: gemv_body's PH)
template <int PRO, int NW, int XCH>
__device__ __forceinline__ void x_dma_rows(const GemvHot& a, _Float16* xl, int xs, XRegs& xr, int k0, int Ks) {
    constexpr int THREADS = NW * 64;
    const int chunks = Ks >> 3;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)xl;
    for (int m = 0; m < a.M; ++m) {
        const _Float16* row = (const _Float16*)a.x + (size_t)m * a.x_stride + k0;
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int c = (int)threadIdx.x + i * THREADS;
            if (c < chunks) gv_glds16(row, 16u * (unsigned)c, lds0 + (unsigned)(m * xs) * 2u + (unsigned)(i * THREADS + wave * 64) * 16u);
        }
    }
    if (PRO == PRO_RMSNORM || PRO == PRO_RMSNORM_SUMS) {
        const int last = chunks - 1;
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            int c = (int)threadIdx.x + i * THREADS;
            c = c < last ? c : last;
            xr.w[i] = *(const h8*)((const _Float16*)a.gamma + 8 * c);
        }
    }
    if (PRO == PRO_RMSNORM_SUMS) {
        // the rows' sums of squares arrive as K / 16 partials per row (one per row-tile of the launch that wrote x: its epilogue, AMQ_FINISH): wave m
        // fetches row m's (SUMS_PER_LANE per lane, unconditional -- indices clamped, the surplus zeroed in x_finish_dma) into the registers the one-row
        // path keeps x in; they are older than the ring's loads, like gamma's
        const int P = a.K >> 4;
        const float* ss = (const float*)a.x2 + (size_t)(wave < a.M ? wave : a.M - 1) * P;
        const int lane = (int)threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < SUMS_PER_LANE; ++j) {
            const int idx = lane + 64 * j;
            xr.f[j] = ss[idx < P ? idx : P - 1];
        }
    }
}

// NRING: vector-memory operations the wave has issued since x_dma_rows (the primed ring's loads: payload + meta per tile, amq_common.cuh load_payload)
template <int PRO, int NW, int XCH, int NRING, int MR>
__device__ __forceinline__ void x_finish_dma(const GemvHot& a, const XRegs& xr, _Float16* xl, float* red, int xs, int k0, int Ks) {
    constexpr int THREADS = NW * 64;
    static_assert(NRING >= 0 && NRING <= 63, "counted wait (vmcnt is six bits on gfx9)");
    const int tid = threadIdx.x;
    const int chunks = Ks >> 3;
    // everything this wave issued before the ring's NRING loads has landed: its LDS-DMA transfers (and, RMSNorm, gamma's loads, issued right behind them).
    // Inline asm: the compiler neither sees the transfers nor may it move LDS reads above this wait ("memory").
    // (check_waits: on every path from the last transfer to here the wave has issued AT LEAST NRING vector-memory instructions -- the ring's 2 U
    //  loads, plus gamma's under a fused RMSNorm: those sit between the transfers and the ring and only make the wait stricter)
    AMQ_WAIT_VM("gemv.xrows", NRING, "from=gemv.xdma:>=%0");
    if (PRO == PRO_NONE) return;
    if (PRO == PRO_RMSNORM) {
        // all rows side by side (MR = the kernel's row bound, rows >= M predicated off by uniform branches): the rows' LDS reads, square sums and
        // wave reductions are independent chains; one row at a time they cost ~0.35 us each (profiles/r05_decode_batch.txt)
        h8 v[MR * XCH];
        float ss[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            ss[m] = 0.f;
            if (m < a.M) {
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const int c = tid + i * THREADS;
                    if (c < chunks) v[m * XCH + i] = *(const h8*)(xl + (size_t)m * xs + 8 * c);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < a.M) {
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    if (tid + i * THREADS < chunks) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { float f = (float)v[m * XCH + i][e]; ss[m] += f * f; }
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < a.M) {
                ss[m] = wave_sum(ss[m]);
                if ((tid & 63) == 0) red[m * NW + (tid >> 6)] = ss[m];
            }
        }
        __syncthreads();
        const MeanDiv mean(a.K);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < a.M) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += red[m * NW + w];
                const float rstd = rsqrtf(mean(tot) + a.eps);
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const int c = tid + i * THREADS;
                    if (c < chunks) {
                        h8 r;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { _Float16 nrm = (_Float16)((float)v[m * XCH + i][e] * rstd); r[e] = xr.w[i][e] * nrm; }
                        *(h8*)(xl + (size_t)m * xs + 8 * c) = r;
                    }
                }
            }
        }
        return;
    }
    if (PRO == PRO_RMSNORM_SUMS) {
        // rstd of row `wave` from its partials (fixed order: a lane's SUMS_PER_LANE in index order, then the wave's DPP tree), one barrier, then only the
        // transform gamma * fp16(x * rstd) on this thread's own chunks of every row -- no pass over x for the statistic, no per-row reduction
        const int P = a.K >> 4, lane = tid & 63, wave = tid >> 6;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < SUMS_PER_LANE; ++j) s += (lane + 64 * j < P) ? xr.f[j] : 0.f;
        s = wave_sum(s);
        if (lane == 0) red[wave] = rsqrtf(MeanDiv(a.K)(s) + a.eps);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < a.M) {
                const float rstd = red[m];
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const int c = tid + i * THREADS;
                    if (c < chunks) {
                        _Float16* p = xl + (size_t)m * xs + 8 * c;
                        const h8 v = *(const h8*)p;
                        h8 r;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { _Float16 nrm = (_Float16)((float)v[e] * rstd); r[e] = xr.w[i][e] * nrm; }
                        *(h8*)p = r;
                    }
                }
            }
        }
        return;
    }
    // SiLU * mul: x = fp16(fp16(silu(gate)) * up)
    const int last = chunks - 1;
    for (int m0 = 0; m0 < a.M; m0 += 4) {
        h8 up[4 * XCH];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = m0 + j < a.M ? m0 + j : a.M - 1;
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                int c = tid + i * THREADS;
                c = c < last ? c : last;
                up[j * XCH + i] = *(const h8*)((const _Float16*)a.x2 + (size_t)mm * a.x_stride + k0 + 8 * c);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (m0 + j < a.M) {
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const int c = tid + i * THREADS;
                    if (c < chunks) {
                        _Float16* p = xl + (size_t)(m0 + j) * xs + 8 * c;
                        const h8 g = *(const h8*)p;
                        h8 r;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { _Float16 sg = (_Float16)silu_f((float)g[e]); r[e] = sg * up[j * XCH + i][e]; }
                        *(h8*)p = r;
                    }
                }
            }
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
// PH: K phases (1, or 2 for launches of 7 - 8 rows whose x does not fit LDS whole -- the 7B down_proj, K = 11008): x is staged one K slice at a
//     time; a row-tile's tiles run slice by slice ("virtual row-tiles" of nt tiles each), its accumulators live across the slices, and between two
//     slices the workgroup restages (barrier, LDS-DMA of the next slice behind the still-full weight ring, transform, barrier)
template <int BITS, int MODE, int PRO, int NW, int U, int MATH, int XCH, int RS = 256, bool SC1 = false, int GP = 1, int PH = 1>
__device__ __forceinline__ void gemv_body(const GemvHot& a, const GemvKArgs& blk, int sidx, const void* qweight, const void* meta_base,
                                          int seg_split, int local, _Float16* lds_x, const _Float16* xl, float* xg,
                                          float* red, int xs, int xmode, const XRegs& xr) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = a.K >> 7;
    const int r = lane & 15, o = lane >> 4;
    static_assert(PH == 1 || (RS != 256 && PRO != PRO_RMSNORM && PRO != PRO_RMSNORM_SUMS && MATH != MATH_LINEAR && MATH != MATH_DOT), "K phases: the row kernels, no full-row statistic");
    static_assert(PRO != PRO_RMSNORM_SUMS || (RS != 256 && MATH == MATH_EXACT && GP == 1), "partial-sum RMSNorm: the 2 .. 8-row kernels");
    const int Gp = G / PH;                                        // tiles of one K phase of a row-tile
    const int nt = (Gp - wave + NW - 1) / NW;                     // tiles of one (row-tile, phase) owned by this wave: g = phase * Gp + wave + i*NW
    // this workgroup's row-tiles: rt0 .. rt0 + n_my - 1 (contiguous bytes).  seg_split = gemv_split(): the segment's first `rem` workgroups walk
    // base + 1 row-tiles, the others base -- no workgroup carries more than one row-tile more than any other
    const int sp_base = seg_split & GEMV_SPLIT_BASE_MASK, sp_rem = seg_split >> GEMV_SPLIT_BASE_BITS;
    const int rt0 = local * sp_base + (local < sp_rem ? local : sp_rem);
    const int n_my = sp_base + (local < sp_rem ? 1 : 0);
    const int total = n_my * PH * nt;
    const uint32_t* qw = (const uint32_t*)qweight;
    static_assert(GP == 1 || MATH == MATH_EXACT || MATH == MATH_GS, "groups finer than 128: exact math only");
    // group-scale arithmetic (amq_common.cuh) serves the two-rounding (HQQ) buffers at groups of 128; the one-rounding modes (reference-format
    // weights) and the finer groups keep their exact forms inside the same kernel
    constexpr bool GS = MATH == MATH_GS && MODE == MODE_HQQ && GP == 1;
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
        size_t tile_ = (size_t)(rt0 + ij / PH) * G + (ij % PH) * Gp + (wave + ii * NW);           \
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
    so.sums = (RS != 256 && PRO == PRO_NONE) ? blk.sums_out : nullptr;   // (the 2 .. 8-row kernels without a prologue -- o_proj, down_proj: what a PRO_RMSNORM_SUMS launch over y will read)
    so.sums_stride = blk.sums_stride;
#ifndef AMQ_ABL_NOSTAGE
    // xmode: 1 = one row held in registers (x_issue ran), 2 = rows on their way into LDS (x_dma_rows ran; the <= 8-row kernels, RS != 256), 0 = generic
    if (xmode == 1) x_finish<PRO, NW, XCH, MATH == MATH_LINEAR>(a, xr, lds_x, xg, red);
    else if (RS != 256 && xmode == 2) {
#ifdef AMQ_ABL_NOLOAD
        x_finish_dma<PRO, NW, XCH, 0, RowsCfg<RS>::MRMAX>(a, xr, lds_x, red, xs, 0, a.K / PH);
#else
        if constexpr (U * 2 <= 63) x_finish_dma<PRO, NW, XCH, U * 2, RowsCfg<RS>::MRMAX>(a, xr, lds_x, red, xs, 0, a.K / PH);
#endif
    }
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
        const int rt_ = rt0 + cj / PH;                                                           \
        if (e_on) {                                           /* M * 16 <= 256 <= threads */     \
            float tot_ = 0.f;                                                                    \
            _Pragma("unroll") for (int w_ = 0; w_ < NW; ++w_) tot_ += rp_[w_ * RS + threadIdx.x];  \
            _Float16 y_ = (_Float16)tot_;                     /* fp16(matmul) */                 \
            if (so.bias) y_ = y_ + pf_bias;                   /* out + bias      (fp16 add) */   \
            if (so.residual) y_ = pf_res + y_;                /* residual + out */               \
            if (SC1) __hip_atomic_store((unsigned short*)(so.y + (size_t)e_m * so.y_stride + rt_ * 16 + e_c),          \
                                        __builtin_bit_cast(unsigned short, y_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            else so.y[(size_t)e_m * so.y_stride + rt_ * 16 + e_c] = y_;                          \
            if constexpr (RS != 256 && PRO == PRO_NONE) {     /* sum of squares of this row-tile's 16 values of row e_m (one DPP row: fixed order) */ \
                if (so.sums) {                                                                   \
                    const float q_ = row16_total((float)y_ * (float)y_);                         \
                    if (e_c == 0) so.sums[(size_t)e_m * so.sums_stride + rt_] = q_;              \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
        if (cj / PH + 1 < n_my) AMQ_EPI_PREFETCH(rt_ + 1);                                       \
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
        else if constexpr (GS) dequant_lane_gs<BITS>(pay[slot].w, gs_zero(meta[slot]), wv);      /* first rounding only; the scale follows the MFMAs */ \
        else if constexpr (GP == 1 && MODE == MODE_FMA1) dequant_lane_fma1<BITS>(pay[slot].w, meta[slot], wv);   /* reference-format weights, one op per pair */ \
        else if constexpr (GP == 1 && MODE == MODE_HQQ && AMQ_SD_E9) {                           /* A/B: the fields where the packing left them (E = -9), then the multiply */ \
            dequant_lane_gs<BITS>(pay[slot].w, gs_zero(meta[slot]), wv);                         \
            const h2 sc9_ = bcast(meta[slot].x) * bcast((_Float16)512.0f);                       \
            _Pragma("unroll") for (int p_ = 0; p_ < 16; ++p_) wv[p_] = wv[p_] * sc9_;            \
        }                                                                                        \
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
            f4 c_ = (MATH == MATH_LINEAR || GS) ? (f4){0.f, 0.f, 0.f, 0.f} : accm;               \
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
            } else if constexpr (GS) {                                                           \
                /* this lane's column r, rows 4*o + i: y += (s 2^-E) * sum_k x_k RN16((q_k - z) 2^E) */ \
                const float sg_ = gs_scale(meta[slot]);                                          \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) accm[i_] = __builtin_fmaf(sg_, c_[i_], accm[i_]); \
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
        if (++ci == nt) {                                                                        \
            if (PH == 1 || cj % PH == PH - 1) AMQ_FINISH();                                      \
            if constexpr (PH > 1) { if (cj + 1 < n_my * PH) AMQ_RESTAGE((cj + 1) % PH); }        \
            ci = 0; ++cj;                                                                        \
        }                                                                                        \
    } while (0)
#endif
    // next K slice of x (PH > 1).  The slice's transfers are the YOUNGEST vector-memory operations of the wave here (the ring's loads were issued before
    // them), and vmcnt retires in issue order: only vmcnt(0) says they have landed.  (A counted wait that left the ring's 2 U operations "in flight"
    // -- the form of the initial staging, where the ring is primed AFTER the transfers -- let the last rows' transfers be the ones left in flight: a
    // rare wrong row-tile at 8 rows of K = 11008, caught by test_gemv_rows_staged_by_dma on one box in many.)
#define AMQ_RESTAGE_NRING 0
#define AMQ_RESTAGE(ph_)                                                                         \
    do {                                                                                         \
        __syncthreads();                                      /* every wave is done reading the slice in LDS */ \
        x_dma_rows<PRO, NW, XCH>(a, lds_x, xs, const_cast<XRegs&>(xr), (ph_) * (a.K / PH), a.K / PH);          \
        x_finish_dma<PRO, NW, XCH, AMQ_RESTAGE_NRING, RowsCfg<RS>::MRMAX>(a, xr, lds_x, red, xs, (ph_) * (a.K / PH), a.K / PH); \
        __syncthreads();                                                                         \
    } while (0)

    if (nt == 0) {                                                // K < 128 * NW: this wave owns no tile (never with PH > 1: launch_gemv)
        for (int j = 0; j < n_my; ++j) { AMQ_FINISH(); cj += PH; }
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
    constexpr int OPS_PER_TILE = 2;                                // payload (one load at every bit-width) + meta
#define AMQ_T() __builtin_amdgcn_s_memtime()
    AMQ_STAMP_AT(blk, 96 + wave);                                  // realtime: about to wait for the first tile
    bool first_ = true;
    for (; idx + 2 * U <= total; idx += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long t0 = AMQ_T();
            AMQ_WAIT_VM("gemv.stamp", (U - 1) * OPS_PER_TILE, "");               /* (stamp build only: the wait the compiler would insert itself) */
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
#undef AMQ_RESTAGE
#undef AMQ_RESTAGE_NRING
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
// (the group-scale bodies hold a tile's MFMA sum apart from the running accumulators: four registers more than the exact bodies, which sit at 78.
//  Held to the same three-workgroups-per-CU budget)
#ifndef AMQ_LB_WAVES_M
#define AMQ_LB_WAVES_M(NW_, GP_, MATH_, XCH_) (((MATH_) == MATH_GS && (NW_) == 8 && (GP_) == 1 && (XCH_) == 1) ? 6 : AMQ_LB_WAVES_G(NW_, GP_))
#endif
// (the 2 .. 4-row kernels on 8 waves: three workgroups per CU as at one row, rows held in registers during the prologue included;
//  the 5 .. 8-row kernels run two 8-wave workgroups or one 16-wave workgroup per CU: 128 registers)
#ifndef AMQ_RS64_WAVES
#define AMQ_RS64_WAVES 6
#endif
#ifndef AMQ_LB_WAVES_R
#define AMQ_LB_WAVES_R(NW_, GP_, MATH_, XCH_, RS_) (((RS_) == 64 && (NW_) == 8) ? AMQ_RS64_WAVES : ((RS_) == 128 && (NW_) == 8) ? 4 : (RS_) == 128 ? 1 : AMQ_LB_WAVES_M(NW_, GP_, MATH_, XCH_))
#endif
template <int PRO, int NW, int U, int MATH, int XCH, int RS = 256, int GP = 1, int PH = 1>
__global__ __launch_bounds__(NW * 64, AMQ_LB_WAVES_R(NW, GP, MATH, XCH, RS)) void gemv_kernel(const void* p_x, const void* p_xw, const void* p_qw0,
                                                                     const void* p_mt0, int p_K, int p_m_nseg, int p_rpt,
                                                                     int p_n_rt0, int p_key0, float p_eps, GemvKArgs blk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifndef AMQ_NO_PRIO_PROGRESS
    __builtin_amdgcn_s_setprio(3);          // prologue and first quarter of the stream at top priority (see the main loop)
#endif
    GemvHot a;
    a.x = p_x; a.x2 = p_xw; a.gamma = p_xw;
    if (PRO == PRO_RMSNORM_SUMS) a.x2 = blk.x2;                   // (gamma arrives preloaded; the partial sums' address is one scalar load from the argument block)
    a.K = p_K; a.M = p_m_nseg & 0xFF; a.rpt = p_rpt; a.eps = p_eps;
    const int nseg = p_m_nseg >> 16;
    const bool dense = (p_m_nseg >> 8) & 1;                       // x rows are K apart: the register-held paths need no argument-block fetch for the stride
    const int Kst = a.K / PH;                                     // columns of x in LDS at a time
    const bool fits = (Kst >> 3) <= XCH * NW * 64;
    int xmode = (a.M == 1 && fits && PH == 1) ? 1 : 0;
    if (RS != 256 && MATH != MATH_LINEAR && fits && dense && a.M >= 2 && a.M <= RowsCfg<RS>::MRMAX) xmode = 2;
    if (PH > 1 && xmode != 2) return;                             // (launch_gemv only sends what the phased form takes)
    if (PRO == PRO_RMSNORM_SUMS && xmode != 2) return;            // (... and the partial-sum prologue: 5 .. 8 dense rows that fit LDS)
    const bool slow_x = xmode == 0;                               // generic staging path
    a.x_stride = slow_x ? blk.x_stride : a.K;
    const int xs = Kst + XPAD;
    _Float16* xl = (_Float16*)smem;
    const size_t xbytes = ((size_t)a.M * xs * 2 + 15) & ~(size_t)15;
    float* xg = (float*)(smem + xbytes);                                        // [G][16] (linear math only)
    const size_t xgbytes = (MATH == MATH_LINEAR) ? (size_t)(a.K >> 7) * 64 : 0;
    float* red = (float*)(smem + xbytes + xgbytes);                             // [2][NW][16][16]

    // decode fast path: one activation row whose chunks fit the per-thread registers -> its loads leave FIRST, before the per-segment arguments'
    // kernarg round trip below (everything they need arrives preloaded): staging x -- arrival, norm, LDS, barrier -- is the critical path of a
    // launch's prologue (issuing the first weight tile ahead of them instead measured 2-3 % slower)
    XRegs xr;
    if (xmode == 1) x_issue<PRO, NW, XCH>(a, xr);
    else if (RS != 256 && xmode == 2) x_dma_rows<PRO, NW, XCH>(a, xl, xs, xr, 0, Kst);

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
    constexpr bool HAS_FMA1 = (MATH == MATH_EXACT || MATH == MATH_GS) && GP == 1;      // (launch_gemv maps MODE_FMA1 to MODE_FMA for the kernels without those bodies)
    bool done = false;
    if constexpr (HAS_FMA1) {
        if (key == 4 * 4 + MODE_FMA1) { gemv_body<4, MODE_FMA1, PRO, NW, U, MATH, XCH, RS, false, GP, PH>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, xmode, xr); done = true; }
        else if (key == 3 * 4 + MODE_FMA1) { gemv_body<3, MODE_FMA1, PRO, NW, U, MATH, XCH, RS, false, GP, PH>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, xmode, xr); done = true; }
        else if (key == 2 * 4 + MODE_FMA1) { gemv_body<2, MODE_FMA1, PRO, NW, U, MATH, XCH, RS, false, GP, PH>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, xmode, xr); done = true; }
    }
    if (!done) switch (key) {
        case 4 * 4 + MODE_HQQ: gemv_body<4, MODE_HQQ, PRO, NW, U, MATH, XCH, RS, false, GP, PH>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, xmode, xr); break;
        case 3 * 4 + MODE_HQQ: gemv_body<3, MODE_HQQ, PRO, NW, U, MATH, XCH, RS, false, GP, PH>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, xmode, xr); break;
        case 2 * 4 + MODE_HQQ: gemv_body<2, MODE_HQQ, PRO, NW, U, MATH, XCH, RS, false, GP, PH>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, xmode, xr); break;
        case 4 * 4 + MODE_FMA: gemv_body<4, MODE_FMA, PRO, NW, U, MATH, XCH, RS, false, GP, PH>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, xmode, xr); break;
        case 3 * 4 + MODE_FMA: gemv_body<3, MODE_FMA, PRO, NW, U, MATH, XCH, RS, false, GP, PH>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, xmode, xr); break;
        default:               gemv_body<2, MODE_FMA, PRO, NW, U, MATH, XCH, RS, false, GP, PH>(a, blk, sidx, qwp, mtp, nrt, local, xl, xuse, xg, red, xs, xmode, xr); break;
    }
    if (threadIdx.x < 64) AMQ_STAMP_AT(blk, 4);
#ifdef AMQ_STAMP
    if (threadIdx.x == 0 && blk.stamps) blk.stamps[(size_t)blockIdx.x * 128 + 6] = __builtin_amdgcn_s_memtime();
#endif
}


template <int PRO, int NW, int U, int MATH, int XCH = XCfg<NW>::XC, int RS = 256, int GP = 1, int PH = 1>
inline hipError_t launch_one(const GemvKArgs& a, int total_wg, size_t lds, hipStream_t st) {
    auto kern = gemv_kernel<PRO, NW, U, MATH, XCH, RS, GP, PH>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const void* xw = PRO == PRO_SILU_MUL ? a.x2 : a.gamma;       // (PRO_RMSNORM_SUMS: gamma; the kernel reads x2 = the partial sums from the block)
    hipLaunchKernelGGL(kern, dim3(total_wg), dim3(NW * 64), lds, st, a.x, xw, a.qweight[0], a.meta[0], a.K,
                       a.M | ((a.x_stride == a.K ? 1 : 0) << 8) | (a.nseg << 16), a.rpt, a.n_rt[0], a.key[0], a.eps, a);
    return hipGetLastError();
}

// the default geometry (two tile loads in flight per wave) in one of the two product arithmetics
template <int PRO, int NW, int MATH>
inline hipError_t launch_std(const GemvKArgs& a, int flags, int total_wg, size_t lds, hipStream_t st) {
    if (NW == 8 && (flags & GEMV_FLAG_RS64)) return launch_one<PRO, 8, 2, MATH, XCfg<8>::XC, 64>(a, total_wg, lds, st);
    if (NW == 8 && (flags & GEMV_FLAG_RS128)) return launch_one<PRO, 8, 2, MATH, XCfg<8>::XC, 128>(a, total_wg, lds, st);
    if (NW == 16 && (flags & GEMV_FLAG_RS64)) return launch_one<PRO, 16, 2, MATH, XCfg<16>::XC, 64>(a, total_wg, lds, st);
    if constexpr (NW == 16 && PRO != PRO_RMSNORM) {
        if (flags & GEMV_FLAG_PH2) return launch_one<PRO, 16, 2, MATH, 1, 128, 1, 2>(a, total_wg, lds, st);      // 7 - 8 rows of a long K: two K phases
    }
    if (NW == 16 && (flags & GEMV_FLAG_RS128)) return launch_one<PRO, 16, 2, MATH, XCfg<16>::XC, 128>(a, total_wg, lds, st);
    if (NW == 16 && a.M == 1 && (a.K >> 3) > XCfg<16>::XC * 1024 && (a.K >> 3) <= 4 * 1024)
        return launch_one<PRO, 16, 2, MATH, 4>(a, total_wg, lds, st);        // 16384 < K <= 32768 (70B down_proj)
    if (NW == 8 && a.M == 1 && (a.K >> 3) > 512 && (a.K >> 3) <= 1024)
        return launch_one<PRO, 8, 2, MATH, 2>(a, total_wg, lds, st);         // 4096 < K <= 8192 on 8 waves (two x chunks per thread)
    return launch_one<PRO, NW, 2, MATH>(a, total_wg, lds, st);
}

template <int PRO, int NW>
inline hipError_t launch_nw(const GemvKArgs& a, int flags, int depth, int total_wg, size_t lds, hipStream_t st) {
    const int u = depth ? depth : 2;
    if ((flags & GEMV_FLAG_DOT) && a.M == 1) return launch_one<PRO, NW, 2, MATH_DOT>(a, total_wg, lds, st);
    if (flags & GEMV_FLAG_LINEAR) {
        if (u == 4) return launch_one<PRO, NW, 4, MATH_LINEAR>(a, total_wg, lds, st);
        return launch_one<PRO, NW, 2, MATH_LINEAR>(a, total_wg, lds, st);
    }
    if (u == 4) return launch_one<PRO, NW, 4, MATH_EXACT>(a, total_wg, lds, st);
    if (flags & GEMV_FLAG_GS) return launch_std<PRO, NW, MATH_GS>(a, flags, total_wg, lds, st);
    return launch_std<PRO, NW, MATH_EXACT>(a, flags, total_wg, lds, st);
}

// (RMSNorm from partial sums, PRO_RMSNORM_SUMS: its two kernels -- the 5 .. 8-row geometry, RS = 128, default arithmetic -- are instantiated in
//  amq_gemv_pro3.hip alone; a non-template launcher here would instantiate them in every translation unit that includes this header)

// groups of 64 / 32 (GP = 2 / 4 meta pairs per tile): the default geometry of the exact-math body (two tile loads in flight, generic x staging
// beyond its register-held chunks)
template <int PRO, int GP>
hipError_t launch_pro_g(const GemvKArgs& a, int nw, int total_wg, size_t lds, hipStream_t st) {
    if (nw == 4) return launch_one<PRO, 4, 2, MATH_EXACT, XCfg<4>::XC, 256, GP>(a, total_wg, lds, st);
    if (nw == 16) return launch_one<PRO, 16, 2, MATH_EXACT, XCfg<16>::XC, 256, GP>(a, total_wg, lds, st);
    return launch_one<PRO, 8, 2, MATH_EXACT, XCfg<8>::XC, 256, GP>(a, total_wg, lds, st);
}

template <int PRO>
hipError_t launch_pro(const GemvKArgs& a, int flags, int depth, int nw, int total_wg, size_t lds, hipStream_t st) {
    if (nw == 4) return launch_nw<PRO, 4>(a, flags, depth, total_wg, lds, st);
    if (nw == 16) return launch_nw<PRO, 16>(a, flags, depth, total_wg, lds, st);
    return launch_nw<PRO, 8>(a, flags, depth, total_wg, lds, st);
}


}  // namespace amq
