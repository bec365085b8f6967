// amq_gemm_fewrow.hip -- few-row GEMM over fragment-ordered x, streaming form (round 5), gfx950.
//
// Serves the grouped q/k/v and gate/up launches of a short prompt pass (16 < rows <= 384; the reference's default protocol runs 64:
// amq/amq_speed_benchmark.py:107-109, amq/utils/speed.py:61-71) in place of gemm_skinny_grouped_kernel (amq_gemm.hip), bit for bit.
// Replaces, like it, the FT path's gemm_4bit for few rows (amq/kernel/ft/quantization_new/gemm/gemm_cuda.cu:929-1033).
//
// Where such a launch's time goes (stamps: tools/stamp_fewrow.py, profiles/r05_prompt64.txt; 7B gate/up at 64 rows, 230 workgroups of six blocks, 23 us):
// 4.9 us until the primed loads have landed (a cold start: 208 KB per workgroup requested at once, 48 MB over the chip), a main loop of 11.4 us that is bound
// by the SIMD's issue port (per block and step 16 unpack instructions + 4 MFMAs: the exact two-rounding unpack is amortised over 64 rows only), 3.3 us
// of cross-wave sum and stores.  (x itself is not the limit: 256 workgroups each stream the same 512 KB out of the L2s at 53 - 62 B/clk per CU,
// tools/ubench/l2_intake_dma.hip -- an earlier reading of these launches as intake-bound was wrong.)  The fixed parts are per workgroup ROUND:
// gemm_skinny_grouped_kernel holds a K tile's x fragments for all 64 rows twice over (2 x 64 VGPRs) and can afford four 16-column blocks per workgroup --
// the 7B gate/up launch is 344 workgroups = TWO rounds.  Here the x fragments are taken one MFMA step at a time (4 KiB per wave and step) through a four-slot
// ring that is refilled the moment a step's MFMAs have issued -- a quarter of the registers -- which leaves room for up to SIX column blocks per workgroup:
// gate/up runs as 230 workgroups, ONE round.
//   * 8 waves split K (wave w: K tiles w, w + 8, ...); per tile and step t the wave unpacks 4 register pairs of each column block straight
//     into the MFMA B operand (dequant_pair_sd: the exact two-rounding arithmetic of every other kernel) and issues 4 x NSUB MFMAs.
//   * packed weights and (scale, zero) of the NEXT tile are requested at the start of the current one (two-slot ring).
//   * per-wave partial sums accumulate in the same (tile, step) order as gemm_skinny_kernel's and are added across waves in wave order:
//     results are bit-identical to it (tests/test_gpu_kernels.py::test_gemm_xfrag_grouped_equals_single_launches).
#include "amq_common.cuh"
#include "amq_kernels.h"
#include <type_traits>

namespace amq {

#ifdef AMQ_FS_STAMP                /* diagnostic build: per-workgroup phase stamps (100 MHz realtime counter), tools/stamp_fewrow.py */
unsigned long long* g_fs_stamp_ptr = nullptr;
extern "C" int amq_debug_set_fs_stamps(void* p) { g_fs_stamp_ptr = (unsigned long long*)p; return 0; }
#define FS_STAMP(a_, slot_) do { if ((a_).ws && lane == 0) ((unsigned long long*)(a_).ws)[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + (slot_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FS_STAMP(a_, slot_) do { } while (0)
#endif

// XS: x steps in flight per wave (4 = one K tile ahead, 8 = two: what the registers allow up to four column blocks)
template <int BITS, int MODE, int NSUB, int XS>
__device__ __forceinline__ void fewrow_stream_body(const GemmArgs& a, int bx, unsigned char* smem) {
    constexpr int NWV = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, o = lane >> 4;
    const int G = a.K >> 7;
    const int nblk0 = bx * NSUB, nblk_last = (a.N >> 4) - 1;
    const int m_base = (int)blockIdx.y * 64;
    const uint32_t* qw = (const uint32_t*)a.qweight;
    const h2* mt = (const h2*)a.meta;
    const _Float16* xg = (const _Float16*)a.x + (size_t)blockIdx.y * G * (64 * 128) + lane * 8;     // this row group's fragments, this lane's 16 bytes

    size_t tile0[NSUB];
#pragma unroll
    for (int nb = 0; nb < NSUB; ++nb) tile0[nb] = (size_t)min(nblk0 + nb, nblk_last) * G;

    f4 acc[4][NSUB];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb) acc[mb][nb] = (f4){0, 0, 0, 0};

    static_assert(XS == 4 || XS == 8, "x ring: one or two K tiles of steps");
    h8 xr[XS][4];                                  // [ring slot][row block mb]: the A operands of one MFMA step, refilled per step
    LanePayload<BITS> pay[2][NSUB];
    h2 meta[2][NSUB];
    auto xload = [&](int slot, int t, int kt) {    // fragment j = mb * 4 + t of K tile kt (clamped: tiles past the end re-read the last one, unused)
        const int ktc = kt < G ? kt : G - 1;
        const _Float16* xt = xg + (size_t)ktc * (64 * 128) + t * 512;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) xr[slot][mb] = *(const h8*)(xt + mb * 2048);
    };
    auto wload = [&](int ws, int kt) {
        const bool valid = kt < G;
        const int ktc = valid ? kt : G - 1;
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb) {
            const size_t tile = tile0[nb] + ktc;
            const uint32_t* p = qw + tile * 64 * BITS + lane * BITS;
#pragma unroll
            for (int d = 0; d < BITS; ++d) pay[ws][nb].w[d] = p[d];
            const h2 mv = mt[tile * 16 + r];
            meta[ws][nb] = valid ? mv : (h2){(_Float16)0.f, (_Float16)0.f};      // a tile past the end contributes exact zeros
        }
    };
    auto step = [&](int ws, int xs_, auto tc) {    // MFMA step t of the tile in weight slot ws, A operands in x slot xs_
        constexpr int t = decltype(tc)::value;
#pragma unroll
        for (int nb = 0; nb < NSUB; ++nb) {
#ifdef AMQ_FS_ABL_NOUNPACK         /* timing-only ablation: the payload dwords go to the MFMA as they are */
            const h2 w0 = as_h2(pay[ws][nb].w[0] + t), w1 = as_h2(pay[ws][nb].w[1 % BITS]), w2 = as_h2(pay[ws][nb].w[0] ^ 0x11u), w3 = meta[ws][nb];
#else
            const SdMeta m = sd_meta<BITS, MODE>(meta[ws][nb]);
            const h2 w0 = dequant_pair_sd<BITS, MODE, 4 * t + 0>(pay[ws][nb].w, m), w1 = dequant_pair_sd<BITS, MODE, 4 * t + 1>(pay[ws][nb].w, m);
            const h2 w2 = dequant_pair_sd<BITS, MODE, 4 * t + 2>(pay[ws][nb].w, m), w3 = dequant_pair_sd<BITS, MODE, 4 * t + 3>(pay[ws][nb].w, m);
#endif
            const h8 b = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, w3.x, w3.y};
#ifdef AMQ_FS_ABL_NOMFMA           /* timing-only ablation: one MFMA per column block and step instead of four */
            acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xr[xs_][0] + xr[xs_][1] + xr[xs_][2] + xr[xs_][3], b, acc[0][nb], 0, 0, 0);
#else
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xr[xs_][mb], b, acc[mb][nb], 0, 0, 0);
#endif
        }
#ifdef AMQ_FS_INTERLEAVE           /* A/B: a block's four MFMAs issue one per four unpack instructions of the NEXT block (the default order is 16 unpack, 4 MFMAs per block) */
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
#pragma unroll
        for (int nb = 0; nb + 1 < NSUB; ++nb)
#pragma unroll
            for (int i_ = 0; i_ < 4; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#endif
    };

    // prime: weights of this wave's first two tiles, x steps of the first (XS = 8: first two) tiles
    constexpr int TA = XS / 4;                     // tiles of x the ring runs ahead
    FS_STAMP(a, wave);
#ifdef AMQ_FS_STAMP
    if (wave == 0 && lane == 0 && a.ws) ((unsigned long long*)a.ws)[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + 40] =
        ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    wload(0, wave);
#pragma unroll
    for (int s_ = 0; s_ < XS; ++s_) xload(s_, s_ & 3, wave + NWV * (s_ >> 2));
    wload(1, wave + NWV);
    const int nt = (G + NWV - 1) / NWV;            // K tiles of wave 0 (the other waves' extra tile carries zero meta)
    for (int i = 0; i < nt; i += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#ifdef AMQ_FS_STAMP
            if (i == 0 && d == 0) { AMQ_WAIT_VM("fewrow.primed", 0, ""); FS_STAMP(a, 8 + wave); }      // everything primed has landed
#endif
            const int kt_next = wave + NWV * (i + d + TA);
            const int s0 = (XS == 8 ? d * 4 : 0);  // this tile's first x slot (the loop is unrolled over d: compile-time)
            step(d, s0 + 0, std::integral_constant<int, 0>{}); xload(s0 + 0, 0, kt_next);      // a slot is refilled as soon as its MFMAs have issued
            step(d, s0 + 1, std::integral_constant<int, 1>{}); xload(s0 + 1, 1, kt_next);
            step(d, s0 + 2, std::integral_constant<int, 2>{}); xload(s0 + 2, 2, kt_next);
            step(d, s0 + 3, std::integral_constant<int, 3>{}); xload(s0 + 3, 3, kt_next);
            wload(d, wave + NWV * (i + d + 2));
        }
    }

    FS_STAMP(a, 16 + wave);
    // cross-wave sum in wave order, at most 12 accumulators (96 KB) per pass; wave w finishes accumulators w, w + 8, ... of the pass
    constexpr int NA = 4 * NSUB, PASS = NA < 12 ? NA : 12;
    f4* const part = (f4*)smem;                    // [NWV][PASS][64 lanes]
    const _Float16* bias = (const _Float16*)a.bias;
    const _Float16* res = (const _Float16*)a.residual;
    _Float16* y = (_Float16*)a.y;
#pragma unroll
    for (int p0 = 0; p0 < NA; p0 += PASS) {
        if (p0) __syncthreads();                   // the previous pass's readers are done
#pragma unroll
        for (int q = 0; q < PASS; ++q)
            if (p0 + q < NA) part[(wave * PASS + q) * 64 + lane] = acc[(p0 + q) / NSUB][(p0 + q) % NSUB];
        __syncthreads();
        if (p0 == 0) FS_STAMP(a, 32 + wave);
        for (int q = wave; q < PASS && p0 + q < NA; q += NWV) {
            const int idx = p0 + q, mb = idx / NSUB, nb = idx % NSUB;
            f4 s = part[q * 64 + lane];
#pragma unroll
            for (int w = 1; w < NWV; ++w) {
                const f4 p = part[(w * PASS + q) * 64 + lane];
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += p[i];
            }
            if (nblk0 + nb > nblk_last) continue;
            const int n = (nblk0 + nb) * 16 + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m_base + mb * 16 + 4 * o + i;
                if (m < a.M) {
                    _Float16 v = (_Float16)s[i];
                    if (bias) v = v + bias[n];
                    if (res) v = res[(size_t)m * a.y_stride + n] + v;
                    y[(size_t)m * a.y_stride + n] = v;
                }
            }
        }
    }
    FS_STAMP(a, 24 + wave);
}

struct FewrowSegs {
    int nseg;
    int wg_begin[GEMV_MAX_SEG];            // first workgroup (blockIdx.x) of each segment
    int N[GEMV_MAX_SEG], key[GEMV_MAX_SEG];    // key = bits * 2 + (mode != MODE_HQQ)
    const void* qweight[GEMV_MAX_SEG]; const void* meta[GEMV_MAX_SEG]; const void* bias[GEMV_MAX_SEG];
    const void* residual[GEMV_MAX_SEG]; void* y[GEMV_MAX_SEG]; int y_stride[GEMV_MAX_SEG];
#ifdef AMQ_FS_STAMP
    unsigned long long* stamps;
#endif
};

template <int NSUB>
__global__ __launch_bounds__(512, 2) void gemm_fewrow_stream_kernel(const void* xf, int M, int K, FewrowSegs sg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XS = NSUB <= 3 ? 8 : 4;          // x steps in flight: two K tiles where the accumulators leave room (<= 256 VGPRs, no scratch)
    int seg = 0;
#pragma unroll
    for (int i = 1; i < GEMV_MAX_SEG; ++i)
        if (i < sg.nseg && (int)blockIdx.x >= sg.wg_begin[i]) seg = i;
    GemmArgs a{xf, sg.qweight[seg], sg.meta[seg], sg.bias[seg], sg.y[seg], M, sg.N[seg], K, sg.key[seg] >> 1, sg.key[seg] & 1, K,
               sg.y_stride[seg], nullptr, 1, sg.residual[seg], nullptr};
#ifdef AMQ_FS_STAMP
    a.ws = (float*)sg.stamps;
#endif
    const int bx = (int)blockIdx.x - sg.wg_begin[seg];
    switch (sg.key[seg]) {
        case 4 * 2 + MODE_HQQ: fewrow_stream_body<4, MODE_HQQ, NSUB, XS>(a, bx, smem); break;
        case 3 * 2 + MODE_HQQ: fewrow_stream_body<3, MODE_HQQ, NSUB, XS>(a, bx, smem); break;
        case 2 * 2 + MODE_HQQ: fewrow_stream_body<2, MODE_HQQ, NSUB, XS>(a, bx, smem); break;
        case 4 * 2 + MODE_FMA: fewrow_stream_body<4, MODE_FMA, NSUB, XS>(a, bx, smem); break;
        case 3 * 2 + MODE_FMA: fewrow_stream_body<3, MODE_FMA, NSUB, XS>(a, bx, smem); break;
        default: fewrow_stream_body<2, MODE_FMA, NSUB, XS>(a, bx, smem); break;
    }
}

template <int NSUB>
static hipError_t fewrow_stream_launch(const void* xf, int M, int K, FewrowSegs& sg, int total_wg, hipStream_t st) {
    auto k = gemm_fewrow_stream_kernel<NSUB>;
    constexpr int NA = 4 * NSUB, PASS = NA < 12 ? NA : 12;
    constexpr int LDS = 8 * PASS * 1024;
    if (LDS > 64 * 1024) {
        static unsigned long long attr_done = 0;
        const hipError_t attr = ensure_dyn_lds(attr_done, (const void*)k, LDS);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(k, dim3(total_wg, (M + 63) / 64), dim3(512), LDS, st, xf, M, K, sg);
    return hipGetLastError();
}

// column blocks per workgroup: the fewest (1, 2, 3, 4 or 6) that put the launch's workgroups into the fewest rounds of the chip's CUs (one workgroup per CU
// at a time: ~200 VGPRs), counted per segment as they are launched -- 7B q/k/v at 64 rows is 3 x 86 = 258 workgroups of three blocks, two over one round
// (14 us for 256 of them, 25 for the launch: profiles/r05_prompt64.txt), and 192 of four
int fewrow_stream_nsub(const int* seg_blocks, int nseg, int row_groups, int cus) {
    static const int cand[5] = {1, 2, 3, 4, 6};
    int best = 6;
    long best_rounds = -1;
    for (int c = 4; c >= 0; --c) {
        long wg = 0;
        for (int i = 0; i < nseg; ++i) wg += (seg_blocks[i] + cand[c] - 1) / cand[c];
        const long rounds = (wg * (row_groups > 0 ? row_groups : 1) + cus - 1) / cus;
        if (best_rounds < 0 || rounds <= best_rounds) { best_rounds = rounds; best = cand[c]; }
    }
    return best;
}

hipError_t launch_gemm_fewrow_stream_grouped(const void* xf, int M, int K, const GemvSeg* segs, int nseg, hipStream_t st, int nsub_forced) {
    StreamDevice sd_(st);                                  // attributes / CU counts of the stream's device
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int seg_blocks[GEMV_MAX_SEG];
    for (int i = 0; i < nseg; ++i) seg_blocks[i] = segs[i].N >> 4;
    const int nsub = nsub_forced > 0 ? nsub_forced : fewrow_stream_nsub(seg_blocks, nseg, (M + 63) / 64, cus);
    FewrowSegs sg{};
    sg.nseg = nseg;
    int wg = 0;
    for (int i = 0; i < nseg; ++i) {
        sg.wg_begin[i] = wg;
        wg += ((segs[i].N >> 4) + nsub - 1) / nsub;
        sg.N[i] = segs[i].N; sg.key[i] = segs[i].bits * 2 + (segs[i].mode == MODE_HQQ ? (int)MODE_HQQ : (int)MODE_FMA);
        sg.qweight[i] = segs[i].qweight; sg.meta[i] = segs[i].meta; sg.bias[i] = segs[i].bias;
        sg.residual[i] = segs[i].residual; sg.y[i] = segs[i].y; sg.y_stride[i] = segs[i].y_stride;
    }
    for (int i = nseg; i < GEMV_MAX_SEG; ++i) sg.wg_begin[i] = 0x7fffffff;
#ifdef AMQ_FS_STAMP
    sg.stamps = g_fs_stamp_ptr;
#endif
    switch (nsub) {
        case 1: return fewrow_stream_launch<1>(xf, M, K, sg, wg, st);
        case 2: return fewrow_stream_launch<2>(xf, M, K, sg, wg, st);
        case 3: return fewrow_stream_launch<3>(xf, M, K, sg, wg, st);
        case 4: return fewrow_stream_launch<4>(xf, M, K, sg, wg, st);
        default: return fewrow_stream_launch<6>(xf, M, K, sg, wg, st);
    }
}

}  // namespace amq
