// amq_gemm.hip -- y[M,N] = x[M,K] . W^T for many rows (prefill / batched), gfx950.
//
// Replaces the reference's tensor-core paths
//   gemm_w4a16_T1 / gemm_w4a16_T2 (amq/kernel/ft/quantization_new/gemm/gemm_cuda.cu:290-586, 746-927)
//   and GPTQLinear.forward's torch unpack + matmul branch for M >= 128
//   (hqq/backends/autogptq.py:245-283)
// for 2/3/4-bit alike, over the native AMQ-T16 layout.
//
// Structure: workgroup tile BM x (64*NSUB) (BM = 128 or 64, NSUB = 2 or 4), BK = 128 (one quant group), 4 waves side by
// side along N (wave tile BM x 16*NSUB).  x tiles are double-buffered through LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR staging, no ds_write pass -- the register-staged first version spent 30% of
// its time there, profiles/r01b_gemm_ablations.txt); the LDS image is lane-linear per wave-instruction (four whole
// 256-byte rows), so the bank swizzle is applied on the SOURCE address: 16-byte slot s of row r lives at slot s ^ (r & 15).
// The packed W tile never touches LDS -- each lane's 16/12/8-byte payload unpacks
// directly into the B operand of v_mfma_f32_16x16x32_f16 and is
// reused across all BM/16 row blocks, so the unpack VALU work is amortised BM/16x.
// fp32 accumulate.  Launches with too few tiles to fill the chip run split-K into fp32 slices of a caller-owned workspace
// followed by an ordered reduce (deterministic); a few dozen rows take gemm_skinny_kernel further down instead.
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

constexpr int GM_THREADS = 256;
constexpr int GM_LDA = 128;         // halves per staged x row: 256 B = 16 sixteen-byte slots, XOR-swizzled by the row (see above):
                                    // slot ((4t + o) ^ r) is distinct inside every ds_read_b128 lane group

// NSUB = 16-column sub-tiles per wave: every A fragment read from LDS feeds NSUB MFMAs (LDS read traffic per
// MFMA falls as 1/NSUB; accumulators grow as BM/16 * NSUB * 4 VGPRs).  Workgroup tile = BM x (4 waves * 16 * NSUB).
// GP: (scale, zero) pairs per (row, tile) = 128 / group (amq_common.cuh)
template <int BITS, int MODE, int BM, int NSUB, int GP = 1>
__global__ __launch_bounds__(GM_THREADS) void gemm_kernel(GemmArgs a) {
    constexpr int GM_BN = 4 * 16 * NSUB;
    constexpr int MBLK = BM / 16;              // 16-row blocks per wave tile
    constexpr int ACH = BM * 16 / GM_THREADS;  // LDS-DMA instructions per wave and x tile (= 16-byte chunks per thread)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* const abase = (_Float16*)smem;
    auto abuf = [&](int i) { return abase + (i & 1) * (BM * GM_LDA); };

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, o = lane >> 4;
    const int G = a.K >> 7;
    const int ntn = (a.N + GM_BN - 1) / GM_BN;
    // consecutive workgroups walk M for a fixed N panel: the packed W panel
    // (tiny) stays in L2 while x tiles stream
    const int bn = (int)blockIdx.x % ntn, bm = (int)blockIdx.x / ntn;
    // split-K (few-row launches that would leave most CUs idle): blockIdx.y owns K steps [k0, k1) and writes fp32
    // partials to its own workspace slice; splitk_reduce_kernel sums the slices in a fixed order (deterministic)
    const int sp = (int)blockIdx.y;
    const int k0 = (int)((long)sp * G / a.splits), k1 = (int)((long)(sp + 1) * G / a.splits);
    const int m0 = bm * BM, n0 = bn * GM_BN + wave * (16 * NSUB);

    const _Float16* x = (const _Float16*)a.x;
    const uint32_t* qw = (const uint32_t*)a.qweight;
    const h2* mt = (const h2*)a.meta;

    f4 acc[MBLK][NSUB];
#pragma unroll
    for (int i = 0; i < MBLK; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j) acc[i][j] = (f4){0, 0, 0, 0};

    LanePayload<BITS> pay[NSUB];
    h2 meta[NSUB];
    [[maybe_unused]] MetaG<GP> metag[NSUB];                // (GP > 1: `meta` is unused)
    // LDS-DMA of one BM x 128 x tile: wave-instruction i (0 .. BM/4-1) fills rows 4i .. 4i+3; wave w issues i = w + 4j.
    // Lane l supplies row 4i + (l >> 4), LDS slot l & 15  <-  global slot (l & 15) ^ (row & 15).
    // Rows past M re-read row M-1 (computed, never stored).
    auto issue_a = [&](int kt, _Float16* buf) {
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int i = wave + 4 * j;
            const int row = 4 * i + (lane >> 4);
            const int slot = (lane & 15) ^ (row & 15);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            const _Float16* src = x + (size_t)m * a.x_stride + kt * 128 + slot * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(buf + i * 512), 16, 0, 0);
        }
    };
    auto load_b = [&](int kt) {
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb) {
            // clamp column blocks past N (ragged N % 128): computed, never stored
            const int nblk = min((n0 >> 4) + nb, (a.N >> 4) - 1);
            const size_t tile = (size_t)nblk * G + kt;
            // plain (cached) loads: the W panel is re-read by every M block
            const uint32_t* p = qw + tile * 64 * BITS + lane * BITS;
#pragma unroll
            for (int d = 0; d < BITS; ++d) pay[nb].w[d] = p[d];
            if constexpr (GP == 1) meta[nb] = mt[tile * 16 + r];
            else {
                const h2* mp = mt + (tile * 16 + r) * GP;
#pragma unroll
                for (int s_ = 0; s_ < GP; ++s_) metag[nb].p[s_] = mp[s_];
            }
        }
    };
    auto unpack = [&](int nb, h2* out) {
        if constexpr (GP == 1) dequant_lane_sd<BITS, MODE>(pay[nb].w, meta[nb], out);
        else dequant_lane_sd_g<BITS, MODE, GP>(pay[nb].w, metag[nb], out);
    };

    issue_a(k0, abuf(0));
    load_b(k0);
    __syncthreads();            // (emits vmcnt(0): the DMA has landed before anyone reads)

#ifdef AMQ_GABL_NODEQ      /* ablation: unpack once, outside the K loop */
    h2 wv[NSUB][16];
#pragma unroll
    for (int nb = 0; nb < NSUB; ++nb) unpack(nb, wv[nb]);
#endif
    for (int kt = k0; kt < k1; ++kt) {
#ifdef AMQ_GABL_NOLOADA
        const _Float16* ab = abuf(0);
#else
        const _Float16* ab = abuf(kt - k0);
#endif
#ifndef AMQ_GABL_NODEQ
        h2 wv[NSUB][16];
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb) unpack(nb, wv[nb]);
#endif
#ifdef AMQ_GABL_NOLOADA    /* ablation: the x tile is staged once; the K loop re-reads the same LDS image */
        if (kt + 1 < k1) { load_b(kt + 1); }
#else
        if (kt + 1 < k1) { issue_a(kt + 1, abuf(kt + 1 - k0)); load_b(kt + 1); }      // next tile: DMA + packed W, in flight under the MFMAs
#endif
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            h8 b[NSUB];
#pragma unroll
            for (int nb = 0; nb < NSUB; ++nb)
#pragma unroll
                for (int p = 0; p < 4; ++p) { b[nb][2 * p] = wv[nb][4 * t + p].x; b[nb][2 * p + 1] = wv[nb][4 * t + p].y; }
#pragma unroll
            for (int mb = 0; mb < MBLK; ++mb) {
                const h8 av = *(const h8*)(ab + (mb * 16 + r) * GM_LDA + (((4 * t + o) ^ r) << 3));
#pragma unroll
                for (int nb = 0; nb < NSUB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b[nb], acc[mb][nb], 0, 0, 0);
            }
        }
#ifndef AMQ_GABL_NOBAR
        __syncthreads();        // vmcnt(0) + barrier: tile kt+1 has landed and nobody still reads tile kt's buffer
#endif
    }

    // epilogue: acc[mb][nb][i] = D[m = mb*16 + 4*o + i][n = nb*16 + r]
    const _Float16* bias = (const _Float16*)a.bias;
    const _Float16* res = (const _Float16*)a.residual;
    _Float16* y = (_Float16*)a.y;
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb)
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + mb * 16 + 4 * o + i;
                const int n = n0 + nb * 16 + r;
                if (m < a.M && n < a.N) {
                    if (a.splits > 1) {
                        a.ws[((size_t)sp * a.M + m) * a.N + n] = acc[mb][nb][i];
                    } else {
                        _Float16 v = (_Float16)acc[mb][nb][i];
                        if (bias) v = v + bias[n];
                        if (res) v = res[(size_t)m * a.y_stride + n] + v;
                        y[(size_t)m * a.y_stride + n] = v;
                    }
                }
            }
}

// y[m][n] = fp16(sum over splits, in order) (+ bias) (+ residual): 8 columns per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, const _Float16* bias, const _Float16* res, _Float16* y,
                                                           int M, int N, int y_stride, int splits) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int nchunk = N >> 3;
    if (idx >= (long)M * nchunk) return;
    const int m = (int)(idx / nchunk), n = (int)(idx % nchunk) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < splits; ++s) {
        const f4 lo = *(const f4*)(ws + ((size_t)s * M + m) * N + n);
        const f4 hi = *(const f4*)(ws + ((size_t)s * M + m) * N + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += lo[e]; acc[4 + e] += hi[e]; }
    }
    h8 o, rv;
    if (res) rv = *(const h8*)(res + (size_t)m * y_stride + n);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        _Float16 v = (_Float16)acc[e];
        if (bias) v = v + bias[n + e];
        if (res) v = rv[e] + v;
        o[e] = v;
    }
    *(h8*)(y + (size_t)m * y_stride + n) = o;
}

// The split-K reduce of a launch whose rows go straight into an RMSNorm (down_proj -> the next block's input norm on a short prompt pass): ONE launch
// per 64-row group instead of two.  One workgroup per row of the padded groups: the row's sums over the splits (splitk_reduce_kernel's expression and
// order), bias / residual, y stored; the sum of squares of the ROUNDED row taken from the registers that hold it, in rmsnorm_kernel's order (thread t:
// chunks t, t + 256, ...; wave sums; the four waves in order), then gamma * fp16(y * rstd) in fragment order (rmsnorm_kernel<true>'s layout; rows
// M .. 64 ceil(M / 64) - 1 as zeros).  Bit-identical to the two launches (tests/test_gpu_kernels.py::test_gemm_splitk_norm_xfrag_equals_two_launches).
// Rows of up to 8192 columns (four 8-column chunks per thread); wider rows keep the two launches.
constexpr int RN_CHUNKS = 4;
__global__ __launch_bounds__(256) void splitk_reduce_norm_kernel(const float* ws, const _Float16* bias, const _Float16* res, _Float16* y,
                                                                int M, int N, int y_stride, int splits, const _Float16* gamma, float eps,
                                                                _Float16* xf) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const bool live = m < M;
    const int nchunk = N >> 3;
    h8 row[RN_CHUNKS];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < RN_CHUNKS; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nchunk && live) {
            const int n = c * 8;
            // every split's partials requested before the first is added (gemm_pick_splits gives at most 8: one round trip instead of one per split)
            f4 lo[8], hi[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int sc = s < splits ? s : splits - 1;
                lo[s] = *(const f4*)(ws + ((size_t)sc * M + m) * N + n);
                hi[s] = *(const f4*)(ws + ((size_t)sc * M + m) * N + n + 4);
            }
            h8 o, rv;
            if (res) rv = *(const h8*)(res + (size_t)m * y_stride + n);
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s < splits) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[e] += lo[s][e]; acc[4 + e] += hi[s][e]; }
                }
            for (int s = 8; s < splits; ++s) {                      // (not reached with today's policy)
                const f4 l2 = *(const f4*)(ws + ((size_t)s * M + m) * N + n);
                const f4 h2_ = *(const f4*)(ws + ((size_t)s * M + m) * N + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] += l2[e]; acc[4 + e] += h2_[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 v = (_Float16)acc[e];
                if (bias) v = v + bias[n + e];
                if (res) v = rv[e] + v;
                o[e] = v;
            }
            *(h8*)(y + (size_t)m * y_stride + n) = o;
            row[j] = o;
#pragma unroll
            for (int i = 0; i < 8; ++i) { float f = (float)o[i]; ss += f * f; }
        } else {
            row[j] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)N + eps);
    const int G = N >> 7, gy = m >> 6, mb = (m & 63) >> 4, r = m & 15;
#pragma unroll
    for (int j = 0; j < RN_CHUNKS; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nchunk) {
            const h8 g = *(const h8*)(gamma + 8 * c);
            h8 o8;
#pragma unroll
            for (int i = 0; i < 8; ++i) { _Float16 nv = (_Float16)((float)row[j][i] * rstd); o8[i] = g[i] * nv; }
            if (!live) o8 = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            const int kt = c >> 4, t = (c >> 2) & 3, o = c & 3;     // column 8c = kt*128 + 32t + 8o
            *(h8*)(xf + ((((size_t)gy * G + kt) * 16 + mb * 4 + t) * 64 + o * 16 + r) * 8) = o8;
        }
    }
}

// ---- skinny GEMM: 9 .. 16*MB*gridDim.y rows -------------------------------------------------------------
// A few dozen rows are neither GEMV- nor GEMM-shaped: the packed weights are a few MB (one pass, HBM-trivial) and x
// (M x K fp16, <= 1 MB) lives in L2, so the launch is bound by latency and by how fast the CUs can pull x fragments.
// The tiled kernel above needs split-K to occupy the chip at these sizes (4 K steps per workgroup, a barrier and a full
// load latency each, fp32 partials written and re-read by a second launch: 15.6 us for 64 x 4096 x 4096).  Here a
// workgroup owns 16*NSUB output columns for ALL of K, GEMV style: wave w walks K tiles w, w+NWV, ... with a D-deep
// register ring and no barrier in the loop; each unpacked W tile feeds MB (x NSUB) MFMAs; the waves' fp32 accumulators
// are summed through LDS in wave order (deterministic, no workspace).  K tiles past the end are neutralised by a zero
// (scale, zero) pair instead of a branch, so every vmcnt wait stays a counted one.  Two ways of getting x:
//   XF = false  row-major x: fetched row-coalesced and transposed into MFMA A operands through a wave-private LDS scratch
//   XF = true   fragment-ordered x (launch_xfrag / rmsnorm_kernel<true>): every wave-load is a contiguous KiB that
//               already is an operand; LDS is used only for the final sum
// Measured (MI355X, 3-bit, us per launch, skinny | tiled+split-K): 4096x4096  M=16 6.2|12.2  32 8.0|13.4  64 12.8|15.6;
// 11008x4096  M=32 18.0|21.1  64 33.0|24.0;  4096x11008  M=32 18.0|19.8  64 29.9|20.9.
constexpr int GEMM_SKINNY_MAX = 32;      // rows up to which launch_gemm (GEMM_ROUTE_AUTO) always takes this kernel

// GP: (scale, zero) pairs per (row, tile) = 128 / group (amq_common.cuh); 2 / 4 only in the row-major form (XF = false)
template <int GP> struct SkinnyMeta { typedef MetaG<GP> type; };
template <> struct SkinnyMeta<1> { typedef h2 type; };
template <int BITS, int MB, int NSUB, int GP = 1>
struct SkinnyTile { h8 xr[MB * 4]; LanePayload<BITS> pay[NSUB]; typename SkinnyMeta<GP>::type meta[NSUB]; };

template <int BITS, int MODE, int MB, int NSUB, int D, int NWV, bool XF, int GP = 1>
__device__ __forceinline__ void skinny_body(const GemmArgs& a, int bx, unsigned char* smem) {
    // per-wave transpose scratch (16*MB rows x 256 B, XOR-swizzled like the tiled kernel's x tiles); reused for the
    // cross-wave sum after the K loop
    constexpr int SCR = MB * 16 * 128;             // halves per wave
    _Float16* const scratch = (_Float16*)smem;              // NWV * SCR halves
    static_assert(SCR * 2 >= MB * NSUB * 64 * 16, "accumulator exchange must fit the scratch");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, o = lane >> 4;
    const int G = a.K >> 7;
    const int nblk0 = bx * NSUB, nblk_last = (a.N >> 4) - 1;
    const int m_base = (int)blockIdx.y * (16 * MB);
    const uint32_t* qw = (const uint32_t*)a.qweight;
    const h2* mt = (const h2*)a.meta;
    const _Float16* x = (const _Float16*)a.x;
    _Float16* const my = scratch + wave * SCR;

    // x fragments are fetched row-coalesced (instruction j: rows 4j .. 4j+3, 256 contiguous bytes each; lane l holds
    // row 4j + (l >> 4), 16-byte chunk l & 15) and turned into MFMA A operands through the wave's scratch: fetching them
    // directly in operand layout (16 rows x 64 B per instruction) ran at ~12 B/clk per CU
    int xoff[MB * 4];
#pragma unroll
    for (int j = 0; j < MB * 4; ++j) {
        int m = m_base + 4 * j + o;
        m = m < a.M ? m : a.M - 1;                 // rows past M: computed, never stored
        xoff[j] = m * a.x_stride + r * 8;
    }
    size_t tile0[NSUB];
#pragma unroll
    for (int nb = 0; nb < NSUB; ++nb) tile0[nb] = (size_t)min(nblk0 + nb, nblk_last) * G;

    f4 acc[MB][NSUB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb) acc[mb][nb] = (f4){0, 0, 0, 0};

    SkinnyTile<BITS, MB, NSUB, GP> ring[D];
    auto load = [&](SkinnyTile<BITS, MB, NSUB, GP>& T, int kt) {
        const bool valid = kt < G;
        const int ktc = valid ? kt : G - 1;
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb) {
            const size_t tile = tile0[nb] + ktc;
            const uint32_t* p = qw + tile * 64 * BITS + lane * BITS;
#pragma unroll
            for (int d = 0; d < BITS; ++d) T.pay[nb].w[d] = p[d];
            if constexpr (GP == 1) {
                const h2 mv = mt[tile * 16 + r];
                T.meta[nb] = valid ? mv : (h2){(_Float16)0.f, (_Float16)0.f};
            } else {
                const h2* mp = mt + (tile * 16 + r) * GP;
#pragma unroll
                for (int s_ = 0; s_ < GP; ++s_) {
                    const h2 mv = mp[s_];
                    T.meta[nb].p[s_] = valid ? mv : (h2){(_Float16)0.f, (_Float16)0.f};
                }
            }
        }
        if (XF) {       // fragment-ordered x (amq_xfrag_f16): instruction j = mb*4 + t is one contiguous KiB, already an A operand
            const _Float16* xt = x + ((size_t)blockIdx.y * G + ktc) * (64 * 128) + lane * 8;
#pragma unroll
            for (int j = 0; j < MB * 4; ++j) T.xr[j] = *(const h8*)(xt + j * 512);
        } else {
#pragma unroll
            for (int j = 0; j < MB * 4; ++j) T.xr[j] = *(const h8*)(x + (size_t)xoff[j] + ktc * 128);
        }
    };
    auto compute = [&](const SkinnyTile<BITS, MB, NSUB, GP>& T) {
        h2 wv[NSUB][16];
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb) {
            if constexpr (GP == 1) dequant_lane_sd<BITS, MODE>(T.pay[nb].w, T.meta[nb], wv[nb]);
            else dequant_lane_sd_g<BITS, MODE, GP>(T.pay[nb].w, T.meta[nb], wv[nb]);
        }
        if (!XF) {
#pragma unroll
            for (int j = 0; j < MB * 4; ++j) {
                const int row = 4 * j + o;
                *(h8*)(my + row * 128 + ((r ^ (row & 15)) << 3)) = T.xr[j];
            }
            __builtin_amdgcn_wave_barrier();       // same wave writes and reads: the LDS pipe keeps the order
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            h8 b[NSUB];
#pragma unroll
            for (int nb = 0; nb < NSUB; ++nb)
#pragma unroll
                for (int p = 0; p < 4; ++p) { b[nb][2 * p] = wv[nb][4 * t + p].x; b[nb][2 * p + 1] = wv[nb][4 * t + p].y; }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const h8 av = XF ? T.xr[mb * 4 + t] : *(const h8*)(my + (mb * 16 + r) * 128 + (((4 * t + o) ^ r) << 3));
#pragma unroll
                for (int nb = 0; nb < NSUB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b[nb], acc[mb][nb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

#pragma unroll
    for (int d = 0; d < D; ++d) load(ring[d], wave + NWV * d);
    const int nt = (G + NWV - 1) / NWV;                   // K tiles of wave 0 (the other waves' extra tile is neutralised)
    for (int j = 0; j < nt; j += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            compute(ring[d]);                      // tiles j + d >= nt carry zero meta
            load(ring[d], wave + NWV * (j + d + D));
        }
    }

    // cross-wave sum in wave order, then wave w finishes (mb, nb) pairs w, w + 4, ...
    __syncthreads();                               // every wave is done with its transpose scratch
    f4* const part = (f4*)scratch;                 // [NWV waves][MB * NSUB][64 lanes]
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb) part[(wave * (MB * NSUB) + mb * NSUB + nb) * 64 + lane] = acc[mb][nb];
    __syncthreads();
    const _Float16* bias = (const _Float16*)a.bias;
    const _Float16* res = (const _Float16*)a.residual;
    const _Float16* gate = (const _Float16*)a.gate;
    _Float16* y = (_Float16*)a.y;
    for (int idx = wave; idx < MB * NSUB; idx += NWV) {
        const int mb = idx / NSUB, nb = idx % NSUB;
        f4 s = part[idx * 64 + lane];
#pragma unroll
        for (int w = 1; w < NWV; ++w) {
            const f4 p = part[(w * (MB * NSUB) + idx) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] += p[i];
        }
        const int n = (nblk0 + nb) * 16 + r;
        if (nblk0 + nb > nblk_last) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_base + mb * 16 + 4 * o + i;
            if (m < a.M) {
                _Float16 v = (_Float16)s[i];
                if (bias) v = v + bias[n];
                if (gate) {                                    // act = fp16(silu(gate)) * up   (silu_mul_kernel's expression)
                    const float gf = (float)gate[(size_t)m * a.y_stride + n];
                    v = (_Float16)(gf / (1.0f + __expf(-gf))) * v;
                }
                if (res) v = res[(size_t)m * a.y_stride + n] + v;
                y[(size_t)m * a.y_stride + n] = v;
            }
        }
    }
}

template <int BITS, int MODE, int MB, int NSUB, int D, int NWV, bool XF, int GP = 1>
__global__ __launch_bounds__(NWV * 64) void gemm_skinny_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    skinny_body<BITS, MODE, MB, NSUB, D, NWV, XF, GP>(a, (int)blockIdx.x, smem);
}

// Several linears that consume the same fragment-ordered x (q/k/v, gate/up of a prompt pass), each with its own bit-width, as
// segments of ONE launch: a workgroup serves NSUB column blocks of one segment (three launches of ~11 us for the 7B q/k/v at
// 64 rows are bound by their fixed cost and by every workgroup pulling x through its CU's L1 once per 16 columns; one launch of
// 64-column workgroups pulls it once per 64).
struct SkinnySegs {
    int nseg;
    int wg_begin[GEMV_MAX_SEG];            // first workgroup (blockIdx.x) of each segment
    int N[GEMV_MAX_SEG], key[GEMV_MAX_SEG];    // key = bits * 2 + mode
    const void* qweight[GEMV_MAX_SEG]; const void* meta[GEMV_MAX_SEG]; const void* bias[GEMV_MAX_SEG];
    const void* residual[GEMV_MAX_SEG]; void* y[GEMV_MAX_SEG]; int y_stride[GEMV_MAX_SEG];
};

template <int NSUB>
__global__ __launch_bounds__(512) void gemm_skinny_grouped_kernel(const void* xf, int M, int K, SkinnySegs sg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int seg = 0;
#pragma unroll
    for (int i = 1; i < GEMV_MAX_SEG; ++i)
        if (i < sg.nseg && (int)blockIdx.x >= sg.wg_begin[i]) seg = i;
    GemmArgs a{xf, sg.qweight[seg], sg.meta[seg], sg.bias[seg], sg.y[seg], M, sg.N[seg], K, sg.key[seg] >> 1, sg.key[seg] & 1, K,
               sg.y_stride[seg], nullptr, 1, sg.residual[seg], nullptr};
    const int bx = (int)blockIdx.x - sg.wg_begin[seg];
    switch (sg.key[seg]) {
        case 4 * 2 + MODE_HQQ: skinny_body<4, MODE_HQQ, 4, NSUB, 2, 8, true>(a, bx, smem); break;
        case 3 * 2 + MODE_HQQ: skinny_body<3, MODE_HQQ, 4, NSUB, 2, 8, true>(a, bx, smem); break;
        case 2 * 2 + MODE_HQQ: skinny_body<2, MODE_HQQ, 4, NSUB, 2, 8, true>(a, bx, smem); break;
        case 4 * 2 + MODE_FMA: skinny_body<4, MODE_FMA, 4, NSUB, 2, 8, true>(a, bx, smem); break;
        case 3 * 2 + MODE_FMA: skinny_body<3, MODE_FMA, 4, NSUB, 2, 8, true>(a, bx, smem); break;
        default: skinny_body<2, MODE_FMA, 4, NSUB, 2, 8, true>(a, bx, smem); break;
    }
}

template <int NSUB>
static hipError_t skinny_grouped_launch(const void* xf, int M, int K, SkinnySegs& sg, int total_wg, hipStream_t st) {
    StreamDevice sd_(st);                                  // attributes / CU counts of the stream's device
    auto k = gemm_skinny_grouped_kernel<NSUB>;
    constexpr int LDS = 8 * NSUB * 4096;
    if (LDS > 64 * 1024) {
        static unsigned long long attr_done = 0;
        const hipError_t attr = ensure_dyn_lds(attr_done, (const void*)k, LDS);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(k, dim3(total_wg, (M + 63) / 64), dim3(512), LDS, st, xf, M, K, sg);
    return hipGetLastError();
}

#ifndef AMQ_FEWROW_STREAM
#define AMQ_FEWROW_STREAM 1
#endif
hipError_t launch_gemm_xfrag_grouped(const void* xf, int M, int K, const GemvSeg* segs, int nseg, hipStream_t st, int form, int nsub_forced) {
    if (form == 2) return launch_gemm_fewrow_stream_grouped(xf, M, K, segs, nseg, st, nsub_forced);
    long blocks = 0;
    for (int i = 0; i < nseg; ++i) blocks += segs[i].N >> 4;
    blocks *= (M + 63) / 64;
    const int nsub = blocks <= 320 ? 1 : blocks <= 640 ? 2 : 4;          // as skinny_launch_xf
#if AMQ_FEWROW_STREAM
    // The streaming form (amq_gemm_fewrow.hip: same results bit for bit, up to six column blocks per workgroup) where it saves a ROUND of the chip: a
    // workgroup of either kernel pays a cold start (3 - 5 us) and a cross-wave sum (2 - 3 us) around an issue-bound loop, so a launch costs about one
    // workgroup time per round (profiles/r05_prompt64.txt).  7B gate/up at 64 rows: 344 workgroups of four blocks = two rounds (30 us) against 230 of
    // six = one (23 us); q/k/v: 192 of four in either form (17.7 / 18.2 us) -- the older kernel kept.
    {
        StreamDevice sd_(st);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int ny = (M + 63) / 64;
        long wg_old = 0, wg_new = 0;
        int seg_blocks[GEMV_MAX_SEG];
        for (int i = 0; i < nseg; ++i) seg_blocks[i] = segs[i].N >> 4;
        const int nsub_new = fewrow_stream_nsub(seg_blocks, nseg, ny, cus);
        for (int i = 0; i < nseg; ++i) {
            wg_old += ((segs[i].N >> 4) + nsub - 1) / nsub;
            wg_new += ((segs[i].N >> 4) + nsub_new - 1) / nsub_new;
        }
        const long rounds_old = (wg_old * ny + cus - 1) / cus, rounds_new = (wg_new * ny + cus - 1) / cus;
        if (form == 0 && (rounds_new < rounds_old || AMQ_FEWROW_STREAM == 2)) return launch_gemm_fewrow_stream_grouped(xf, M, K, segs, nseg, st);      // (2: A/B builds, always)
    }
#endif
    SkinnySegs sg{};
    sg.nseg = nseg;
    int wg = 0;
    for (int i = 0; i < nseg; ++i) {
        sg.wg_begin[i] = wg;
        wg += ((segs[i].N >> 4) + nsub - 1) / nsub;
        sg.N[i] = segs[i].N; sg.key[i] = segs[i].bits * 2 + segs[i].mode;
        sg.qweight[i] = segs[i].qweight; sg.meta[i] = segs[i].meta; sg.bias[i] = segs[i].bias;
        sg.residual[i] = segs[i].residual; sg.y[i] = segs[i].y; sg.y_stride[i] = segs[i].y_stride;
    }
    if (nsub == 1) return skinny_grouped_launch<1>(xf, M, K, sg, wg, st);
    if (nsub == 2) return skinny_grouped_launch<2>(xf, M, K, sg, wg, st);
    return skinny_grouped_launch<4>(xf, M, K, sg, wg, st);
}

template <int BITS, int MODE, int MB, int GP = 1>
static hipError_t skinny_launch_mb(const GemmArgs& a, hipStream_t st) {
    StreamDevice sd_(st);                                  // attributes / CU counts of the stream's device
    // 8 waves split K (two per SIMD overlap each other's fetch / transpose / unpack / MFMA phases: 16.2 -> 12.8 us at
    // 64 x 4096 x 4096 against 4 waves with a 3-deep ring); one 16-column block per workgroup
    constexpr int NWV = 8;
    constexpr int LDS = NWV * MB * 16 * 128 * 2;
    const int nblk = a.N >> 4;
    const int ny = (a.M + 16 * MB - 1) / (16 * MB);
    auto k = gemm_skinny_kernel<BITS, MODE, MB, 1, 2, NWV, false, GP>;
    if (LDS > 64 * 1024) {
        static unsigned long long attr_done = 0;
        const hipError_t attr = ensure_dyn_lds(attr_done, (const void*)k, LDS);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(k, dim3(nblk, ny), dim3(NWV * 64), LDS, st, a);
    return hipGetLastError();
}

// fragment-ordered x (launch_xfrag / launch_rmsnorm_xfrag): 64-row groups (grid.y), the A operands are fetched as whole
// contiguous KiB straight into registers -- no transpose scratch, LDS only for the final cross-wave sum.  Every workgroup
// still streams its rows' whole x from L2 (M*K*2 bytes through one CU's L1), so the column blocks per workgroup grow with
// the launch.  Measured (3-bit, us, this | tiled + split-K): 4096^2 M = 64 / 128 / 256: 9.0 / 10.5 / 16.5 | 14.3 / 16.9 /
// 23.0; 11008x4096: 17.3 / 30.1 / 47.7 | 21.7 / 29.7 / 48.2; 4096x11008 (x = 1.4 MB per workgroup): 24.4 / 28.0 / 43.8 |
// 20.4 / 28.1 / 51.4.
template <int BITS, int MODE>
static hipError_t skinny_launch_xf(const GemmArgs& a, hipStream_t st) {
    StreamDevice sd_(st);                                  // attributes / CU counts of the stream's device
    const int nblk = a.N >> 4;
    const int ny = (a.M + 63) / 64;
    const long blocks = (long)nblk * ny;
#ifndef AMQ_XF_NSUB1_MAX
#define AMQ_XF_NSUB1_MAX 320
#endif
    if (blocks <= AMQ_XF_NSUB1_MAX) {
        hipLaunchKernelGGL((gemm_skinny_kernel<BITS, MODE, 4, 1, 2, 8, true>), dim3(nblk, ny), dim3(512), 8 * 4096, st, a);
    } else if (blocks <= 640) {
        hipLaunchKernelGGL((gemm_skinny_kernel<BITS, MODE, 4, 2, 2, 8, true>), dim3((nblk + 1) / 2, ny), dim3(512), 8 * 2 * 4096, st, a);
    } else {
        auto k = gemm_skinny_kernel<BITS, MODE, 4, 4, 2, 8, true>;
        static unsigned long long attr_done = 0;
        const hipError_t attr = ensure_dyn_lds(attr_done, (const void*)k, 8 * 4 * 4096);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL(k, dim3((nblk + 3) / 4, ny), dim3(512), 8 * 4 * 4096, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_gemm_xfrag(const GemmArgs& a, hipStream_t st) {
    if (a.mode == MODE_HQQ) {
        if (a.bits == 4) return skinny_launch_xf<4, MODE_HQQ>(a, st);
        if (a.bits == 3) return skinny_launch_xf<3, MODE_HQQ>(a, st);
        return skinny_launch_xf<2, MODE_HQQ>(a, st);
    }
    if (a.bits == 4) return skinny_launch_xf<4, MODE_FMA>(a, st);
    if (a.bits == 3) return skinny_launch_xf<3, MODE_FMA>(a, st);
    return skinny_launch_xf<2, MODE_FMA>(a, st);
}

// rows -> fragment order: xf[gy][kt][mb*4 + t][lane = 16*o + r][8] = x[gy*64 + mb*16 + r][kt*128 + 32*t + 8*o ..+8]
// (zero for rows >= M).  Source element (m, k) lives at src[m*stride_m + (k >> 7)*stride_kt + (k & 127)]: stride_kt = 128
// for row-major [M, K]; an attention output [heads, M, 128] is read in place with stride_m = 128, stride_kt = M*128.
__global__ __launch_bounds__(256) void xfrag_kernel(const _Float16* src, _Float16* xf, int M, int G, long stride_m, long stride_kt) {
    // one thread per 16-byte chunk of the destination; consecutive threads -> consecutive destination chunks
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(idx & 63), j = (int)((idx >> 6) & 15);
    const long tile = idx >> 10;                   // gy * G + kt
    const int kt = (int)(tile % G), gy = (int)(tile / G);
    const int r = lane & 15, o = lane >> 4, mb = j >> 2, t = j & 3;
    const int m = gy * 64 + mb * 16 + r;
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m < M) v = *(const h8*)(src + (size_t)m * stride_m + (size_t)kt * stride_kt + 32 * t + 8 * o);
    *(h8*)(xf + idx * 8) = v;
}

hipError_t launch_xfrag(const void* src, void* xf, int M, int K, long stride_m, long stride_kt, hipStream_t st) {
    const int G = K >> 7, ny = (M + 63) / 64;
    const long chunks = (long)ny * G * 1024;
    hipLaunchKernelGGL(xfrag_kernel, dim3((unsigned)(chunks / 256)), dim3(256), 0, st, (const _Float16*)src, (_Float16*)xf, M, G,
                       stride_m, stride_kt);
    return hipGetLastError();
}

template <int BITS, int MODE>
static hipError_t skinny_launch(const GemmArgs& a, hipStream_t st) {
    if (a.M <= 16) return skinny_launch_mb<BITS, MODE, 1>(a, st);
    if (a.M <= 32) return skinny_launch_mb<BITS, MODE, 2>(a, st);
    return skinny_launch_mb<BITS, MODE, 4>(a, st);
}

// Groups of 64 / 32 (two / four meta pairs per tile row): the few-row kernel is their only FUSED GEMM.  It serves any number of rows
// (grid.y blocks of 64 rows, each streaming the packed weights once), and up to FINE_SKINNY_MAX_ROWS that beats dequantize-once + the
// 256 x 256-tile fp16 GEMM, which leaves the chip idle at such sizes (7B shapes, 64 rows: ~100 us per linear; profiles/r04_fine_groups.txt)
constexpr int FINE_SKINNY_MAX_ROWS = 256;
bool gemm_fine_takes_skinny(int M) { return M <= FINE_SKINNY_MAX_ROWS; }
// ... and between the few-row kernel and launches that fill the chip with 256 x 256 tiles (where dequantize-once wins) the tiled kernel
// of this file reads the pairs too (the ring / wave-specialised kernels do not)
bool gemm_fine_takes_deq(int M, int N, int K) {
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    return M > FINE_SKINNY_MAX_ROWS && tiles >= 128 && K >= 256;
}

template <int BITS, int MODE, int GP>
static hipError_t skinny_launch_g(const GemmArgs& a, hipStream_t st) {
    if (a.M <= 16) return skinny_launch_mb<BITS, MODE, 1, GP>(a, st);
    if (a.M <= 32) return skinny_launch_mb<BITS, MODE, 2, GP>(a, st);
    return skinny_launch_mb<BITS, MODE, 4, GP>(a, st);
}
template <int GP>
static hipError_t skinny_launch_fine(const GemmArgs& a, hipStream_t st) {
    if (a.mode == MODE_HQQ) {
        if (a.bits == 4) return skinny_launch_g<4, MODE_HQQ, GP>(a, st);
        if (a.bits == 3) return skinny_launch_g<3, MODE_HQQ, GP>(a, st);
        return skinny_launch_g<2, MODE_HQQ, GP>(a, st);
    }
    if (a.bits == 4) return skinny_launch_g<4, MODE_FMA, GP>(a, st);
    if (a.bits == 3) return skinny_launch_g<3, MODE_FMA, GP>(a, st);
    return skinny_launch_g<2, MODE_FMA, GP>(a, st);
}

// up to g_gemm_skinny_max rows always; up to twice that while the column blocks fit one round of workgroups and K is
// short (measured above: at 64 rows the kernel wins for 4096x4096, loses for N = 11008 -- 688 workgroups, 2.7 rounds --
// and for K = 11008 -- 1.4 MB of x per workgroup)
static bool gemm_is_skinny(int M, int N, int K, int route) {
    if (route == GEMM_ROUTE_TILED || route == GEMM_ROUTE_RING || route == GEMM_ROUTE_RING128 || route == GEMM_ROUTE_WS) return false;
    if (route == GEMM_ROUTE_SKINNY) return M <= 64;
    return M <= GEMM_SKINNY_MAX || (M <= 2 * GEMM_SKINNY_MAX && M <= 64 && (N >> 4) <= 320 && K <= 6144);
}

// Split-K policy (profiles/r01c_gemm_split_sweep.txt, 3-bit, us at M = 64 / 128 / 256): up to 256 rows 64-column workgroups aiming at
// two per CU (4096^2: 14.3 / 16.9 / 23.0; 11008x4096: 21.7 / 29.7 / 48.2), beyond that 128-column workgroups aiming at one
// per CU (15.6 / 18.2 / 24.6; 23.8 / 31.2 / 47.6; better from 512 rows on).
static bool split_narrow(int M) { return M <= 256; }

int gemm_pick_splits(int M, int N, int K, int route) {
    const int bn = split_narrow(M) ? 64 : 128, target = split_narrow(M) ? 512 : 256;
    const long wg = (long)((M + 63) / 64) * ((N + bn - 1) / bn);      // 64-row tiles (what such launches use)
    const int G = K >> 7;
    if (gemm_is_skinny(M, N, K, route)) return 1;                        // gemm_skinny_kernel: no partials
    if (route == GEMM_ROUTE_RING || route == GEMM_ROUTE_RING128 || route == GEMM_ROUTE_WS || (route == GEMM_ROUTE_AUTO && gemm_takes_ring(M, N, K))) return 1;
    if (wg >= target * 3 / 4 || G < 4 || (N & 7)) return 1;
    int s = (int)((target + wg - 1) / wg);
    if (s > 8) s = 8;
    if (s > G / 2) s = G / 2;
    return s < 1 ? 1 : s;
}

template <int BITS, int MODE, int BM, int NSUB, int GP = 1>
static hipError_t gemm_launch_cfg(const GemmArgs& a, hipStream_t st) {
    constexpr int BN = 4 * 16 * NSUB;
    const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
    const size_t lds = 2 * BM * GM_LDA * 2;
    auto k = gemm_kernel<BITS, MODE, BM, NSUB, GP>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(ntm * ntn, a.splits), dim3(GM_THREADS), lds, st, a);
    return hipGetLastError();               // (split-K partials: summed by launch_splitk_reduce, behind the tiled dispatch in launch_gemm_nogate)
}

// the second launch of a split-K GEMM: partials summed in split order (+ bias, + residual); with `norm` (and rows of at most 8 * 256 * RN_CHUNKS columns,
// dense y) the RMSNorm into fragment order that the caller wants behind it is part of the same launch
static hipError_t launch_splitk_reduce(const GemmArgs& a, hipStream_t st, GemmNorm* norm) {
    if (norm && !norm->done && (a.N >> 3) <= 256 * RN_CHUNKS && (a.N & 127) == 0 && a.y_stride == a.N) {
        hipLaunchKernelGGL(splitk_reduce_norm_kernel, dim3(((a.M + 63) / 64) * 64), dim3(256), 0, st, (const float*)a.ws, (const _Float16*)a.bias,
                           (const _Float16*)a.residual, (_Float16*)a.y, a.M, a.N, a.y_stride, a.splits, (const _Float16*)norm->gamma, norm->eps,
                           (_Float16*)norm->xf);
        norm->done = true;
        return hipGetLastError();
    }
    const long items = (long)a.M * (a.N >> 3);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, (const float*)a.ws,
                       (const _Float16*)a.bias, (const _Float16*)a.residual, (_Float16*)a.y, a.M, a.N, a.y_stride, a.splits);
    return hipGetLastError();
}

template <int BITS, int MODE, int GP = 1>
static hipError_t gemm_launch_bm(const GemmArgs& a, hipStream_t st) {
    // 64-row tiles while 128-row tiles would leave the chip under-filled (< 1.5 workgroups per CU)
    const long wg128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (a.splits > 1 && split_narrow(a.M)) return gemm_launch_cfg<BITS, MODE, 64, 1, GP>(a, st);
    // 64 x 64 tiles while 64 x 128 ones give at most 1.5 workgroups per CU: two waves per SIMD instead of one cover this kernel's
    // per-K-step drain (4096^2 3-bit, TFLOP/s: M = 512 415 -> 431, 768 490 -> 530; at 1024 rows the wide tile wins, 629 vs 587)
    if (a.splits <= 1 && (long)((a.M + 63) / 64) * ((a.N + 127) / 128) <= 384) return gemm_launch_cfg<BITS, MODE, 64, 1, GP>(a, st);
    if (a.M <= 64 || wg128 < 384 || a.splits > 1) return gemm_launch_cfg<BITS, MODE, 64, 2, GP>(a, st);
    // measured (5120x5120, M = 4096 / 16384): NSUB = 2 -> 0.81 / 0.89-0.92 PFLOP/s; NSUB = 4 needs ~390 VGPRs
    // (one wave per SIMD) and dropped to 0.68 / 0.76 (variant removed)
    return gemm_launch_cfg<BITS, MODE, 128, 2, GP>(a, st);
}

template <int GP>
static hipError_t tiled_launch_fine(const GemmArgs& a, hipStream_t st) {
    if (a.mode == MODE_HQQ) {
        if (a.bits == 4) return gemm_launch_bm<4, MODE_HQQ, GP>(a, st);
        if (a.bits == 3) return gemm_launch_bm<3, MODE_HQQ, GP>(a, st);
        return gemm_launch_bm<2, MODE_HQQ, GP>(a, st);
    }
    if (a.bits == 4) return gemm_launch_bm<4, MODE_FMA, GP>(a, st);
    if (a.bits == 3) return gemm_launch_bm<3, MODE_FMA, GP>(a, st);
    return gemm_launch_bm<2, MODE_FMA, GP>(a, st);
}

// Many-row policy: the ring kernel (amq_gemm_ring.hip) or the wave-specialised kernel (amq_gemm_ws.hip) whenever one of their tile shapes fills the chip (gemm_many_rows_plan),
// else the kernels of this file (profiles/r02_gemm_routes.txt, 3-bit, TFLOP/s tiled | ring: 13824x5120 M = 1024 822 | 1060,
// 4096^2 M = 4096 876 | 1086, M = 2048 (128-row tiles) 814 | 910, M = 1024 615 | 531 -> stays tiled).
bool gemm_takes_ring(int M, int N, int K) { return K >= 256 && gemm_many_rows_plan(M, N) != 0; }

static hipError_t launch_gemm_nogate(const GemmArgs& a, hipStream_t st, int route, GemmNorm* norm);

// GEMM_ROUTE_AUTO takes the dequantize-once route (amq_gemm_f16.hip behind the dequantize kernel) where the launch is MFMA-bound:
// at least DEQ_MIN_ROWS rows and at least one full round of 256 x 256 tiles.  Below, the fused kernels win: they read 2-4 bit
// weights instead of writing and re-reading 16-bit ones, and they have smaller tiles for launches that do not fill the chip
// (profiles/r04_gemm_f16pp.txt, 7B / 13B shapes, dequantize-once over the best fused kernel: 0.63-0.91 at 1024 rows, 0.69-1.03 at 2048,
// 0.90-1.06 at 4096, 1.02-1.06 at 8192, 1.05-1.10 at 32768).
#ifndef AMQ_DEQ_MIN_ROWS
#define AMQ_DEQ_MIN_ROWS 6144
#endif
bool gemm_takes_deq(int M, int N, int K) {
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    return M >= AMQ_DEQ_MIN_ROWS && tiles >= 256 && K >= 256;
}

// the dequantize-once route: forced, or AUTO's choice -- either way only with a scratch for the fp16 weights and without split-K
static bool gemm_runs_deq(const GemmArgs& a, int route) {
    if (!a.w16 || a.splits > 1 || !gemm_f16w_ok(a.M, a.N, a.K, a.x_stride, a.y_stride)) return false;
    if (a.gp > 1) return route == GEMM_ROUTE_DEQ || (route == GEMM_ROUTE_AUTO && gemm_fine_takes_deq(a.M, a.N, a.K));   // groups of 64 / 32
    return route == GEMM_ROUTE_DEQ || (route == GEMM_ROUTE_AUTO && gemm_takes_deq(a.M, a.N, a.K));
}


// a.gate (y = fp16(silu(gate)) * fp16(x . W^T (+ bias)), no residual): formed in the epilogue by the ring and the few-row
// kernels; the tiled kernel (with or without split-K) is followed by the element-wise launch instead -- same expression, same bits.
bool gemm_gate_fused(const GemmArgs& a, int route) {
    const bool ring = (route == GEMM_ROUTE_RING || route == GEMM_ROUTE_RING128 || route == GEMM_ROUTE_WS ||
                       (route == GEMM_ROUTE_AUTO && a.splits <= 1 && gemm_takes_ring(a.M, a.N, a.K))) && gemm_ring_ok(a);
    if (a.gp > 1)                                          // groups of 64 / 32: the few-row kernel and the dequantize-once route apply it, the tiled kernel does not
        return gemm_runs_deq(a, route) || ((route == GEMM_ROUTE_AUTO || route == GEMM_ROUTE_SKINNY) && gemm_fine_takes_skinny(a.M));
    return ring || gemm_is_skinny(a.M, a.N, a.K, route) || gemm_runs_deq(a, route);
}

hipError_t launch_gemm(const GemmArgs& a, hipStream_t st, int route, GemmNorm* norm) {
    StreamDevice sd_(st);                                  // ONE guard for the plan (CU counts), the kernel attributes and the launch itself
    if (norm) norm->done = false;
    hipError_t e;
    if (!a.gate || gemm_gate_fused(a, route)) {
        e = launch_gemm_nogate(a, st, route, norm);
    } else {
        GemmArgs b = a;
        b.gate = nullptr;
        e = launch_gemm_nogate(b, st, route, nullptr);
        if (e == hipSuccess) e = launch_silu_mul(a.gate, a.y, a.y, (long)a.M * a.N, st);       // (contiguous y and gate: checked by the caller)
    }
    if (e != hipSuccess || !norm || norm->done) return e;
    norm->done = true;                                     // no split-K reduce to carry it: the norm as its own launch (dense y: checked by the caller)
    return launch_rmsnorm_xfrag(a.y, norm->gamma, norm->xf, a.M, a.N, norm->eps, st);
}

static hipError_t launch_gemm_tiled(const GemmArgs& a, hipStream_t st) {
    if (a.gp > 1) return a.gp == 2 ? tiled_launch_fine<2>(a, st) : tiled_launch_fine<4>(a, st);
    if (a.mode == MODE_HQQ) {
        if (a.bits == 4) return gemm_launch_bm<4, MODE_HQQ>(a, st);
        if (a.bits == 3) return gemm_launch_bm<3, MODE_HQQ>(a, st);
        return gemm_launch_bm<2, MODE_HQQ>(a, st);
    }
    if (a.bits == 4) return gemm_launch_bm<4, MODE_FMA>(a, st);
    if (a.bits == 3) return gemm_launch_bm<3, MODE_FMA>(a, st);
    return gemm_launch_bm<2, MODE_FMA>(a, st);
}

static hipError_t launch_gemm_tiled_reduced(const GemmArgs& a, hipStream_t st, GemmNorm* norm) {
    const hipError_t e = launch_gemm_tiled(a, st);
    if (e != hipSuccess || a.splits <= 1) return e;
    return launch_splitk_reduce(a, st, norm);
}

static hipError_t launch_gemm_nogate(const GemmArgs& a, hipStream_t st, int route, GemmNorm* norm) {
    if (gemm_runs_deq(a, route)) {
        // MFMA-bound launches: the exact fp16 weights once (amq_dequantize_f16's kernel), then a GEMM with no unpack in its loop
        if (hipError_t e = launch_dequantize(a.bits, a.mode, a.qweight, a.meta, a.N, a.K, a.w16, st, a.gp > 1 ? a.gp : 1)) return e;
        return launch_gemm_f16w(a.x, a.w16, a.bias, a.residual, a.gate, a.y, a.M, a.N, a.K, a.x_stride, a.y_stride, st);
    }
    if (a.gp > 1) {
        if ((route == GEMM_ROUTE_AUTO || route == GEMM_ROUTE_SKINNY) && gemm_fine_takes_skinny(a.M))
            return a.gp == 2 ? skinny_launch_fine<2>(a, st) : skinny_launch_fine<4>(a, st);
        if (route == GEMM_ROUTE_AUTO || route == GEMM_ROUTE_TILED) return launch_gemm_tiled_reduced(a, st, norm);
        return hipErrorInvalidValue;                       // (amq_capi.hip refuses such a call with its reason before it gets here)
    }
    if (route == GEMM_ROUTE_RING && gemm_ring_ok(a)) return launch_gemm_ring(a, st);
    if (route == GEMM_ROUTE_RING128 && gemm_ring_ok(a)) return launch_gemm_ring(a, st, 128);
    if (route == GEMM_ROUTE_WS && gemm_ring_ok(a)) return launch_gemm_ws(a, st);
    if (route == GEMM_ROUTE_AUTO && a.splits <= 1 && gemm_takes_ring(a.M, a.N, a.K) && gemm_ring_ok(a)) {
        const int plan = gemm_many_rows_plan(a.M, a.N);          // 256 / 128-row ring tiles, or the 256 x 128 wave-specialised tile
        return plan < 0 ? launch_gemm_ws(a, st) : launch_gemm_ring(a, st, plan);
    }
    if (gemm_is_skinny(a.M, a.N, a.K, route)) {
        if (a.mode == MODE_HQQ) {
            if (a.bits == 4) return skinny_launch<4, MODE_HQQ>(a, st);
            if (a.bits == 3) return skinny_launch<3, MODE_HQQ>(a, st);
            return skinny_launch<2, MODE_HQQ>(a, st);
        }
        if (a.bits == 4) return skinny_launch<4, MODE_FMA>(a, st);
        if (a.bits == 3) return skinny_launch<3, MODE_FMA>(a, st);
        return skinny_launch<2, MODE_FMA>(a, st);
    }
    return launch_gemm_tiled_reduced(a, st, norm);
}

}  // namespace amq
