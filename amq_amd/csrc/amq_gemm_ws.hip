// amq_gemm_ws.hip -- the many-row GEMM with WAVE SPECIALISATION (round 3), gfx950.
// Slower than the ring kernel at equal fill (profiles/r03_gemm_ws_negative.txt), faster where its half-size tile fills the chip better:
// launch_gemm takes it for those launches (gemm_many_rows_plan, amq_gemm_ring.hip).
//
// Same job as amq_gemm_ring.hip (y[M,N] = x[M,K] . W^T, 2/3/4-bit AMQ-T16 weights, replaces gemm_w4a16_T2,
// amq/kernel/ft/quantization_new/gemm/gemm_cuda.cu:746-927), different division of labour.  In the ring kernel all 8 waves
// carry DMA issue + exact unpack + MFMA in one in-order stream each (1.39 VALU per MFMA; profiles/r02_gemm_ring_pmc.txt:
// matrix pipe 70 % busy).  Here a workgroup is 4 CONSUMER waves (0..3, one per SIMD) and 4 PRODUCER waves (4..7):
//
//   * producers issue every LDS-DMA (x half-tiles of 256 rows x 64 k into a 3-slot ring; the packed tiles + (scale, zero) of
//     their own two 16-column blocks into wave-private 2-slot rings), unpack those tiles exactly (dequant_pair_sd: the
//     arithmetic of every other kernel) one half-tile AHEAD and ds_write the fp16 fragments into a 2-slot ring laid out
//     like the x image (128-byte rows = 64 k of one output column, 16-byte chunk c at position c ^ (col & 7));
//   * consumers run ds_read_b128 + v_mfma_f32_16x16x32_f16 only: wave c owns rows [128 (c >> 1), +128) x columns
//     [64 (c & 1), +64) of the 256 x 128 output tile = 32 accumulator tiles (128 registers), 12 operand reads per 32 MFMAs
//     (the ring kernel: 16 per 32).  W is the MFMA A operand, x the B operand (y^T fragments: 8-byte row-segment stores).
//   * ONE raw s_barrier per half-tile.  B_h: the producers have seen x(h) land (counted vmcnt) and have written W16(h)
//     (lgkmcnt(0)); the consumers have received every operand of half-tile h - 1.  A consumer arrives at B_(h+1) at the START
//     of row block 6 of half-tile h (of 8) -- all its reads of h are back by then -- requests the first operands of h + 1
//     right behind the barrier, two row blocks (16 MFMAs) before their first use.
//   LDS: 3 x 32 KiB (x) + 2 x 16 KiB (fp16 W) + 2 x 8 KiB (packed W) + 2 x 1 KiB (meta) = 149,504 B: one workgroup per CU.
#include "amq_common.cuh"
#include "amq_kernels.h"

#include <utility>

namespace amq {
namespace {

template <class F, int... I>
__device__ __forceinline__ void ws_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void ws_for(F&& f) { ws_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int WS_THREADS = 512;
constexpr int WS_BM = 256, WS_BN = 128;
constexpr int WS_NA = 3;                           // x ring slots
constexpr int WS_ABYTES = WS_BM * 128;             // one x half-tile (256 rows x 64 k fp16)
constexpr int WS_W16 = WS_BN * 128;                // one fp16 W half-tile (128 columns x 64 k)
constexpr int WS_PREG = 2048;                      // a producer's packed-W region in a slot (two tiles, <= 2 x 1024 B)
constexpr int WS_PSLOT = 4 * WS_PREG;
constexpr int WS_MSLOT = 4 * 256;
constexpr int WS_LDS = WS_NA * WS_ABYTES + 2 * WS_W16 + 2 * WS_PSLOT + 2 * WS_MSLOT;

// (see amq_gemm_ring.hip: asm, not the builtin; M0 set and left)
__device__ __forceinline__ void ws_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void ws_glds4(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
#define WS_FENCE() asm volatile("" ::: "memory")

#ifdef AMQ_WS_CYCLES               /* diagnostic build: shader cycles of every workgroup's consumer wave 0 over its K loop (tools/ws_cycles.py) */
__device__ unsigned long long ws_cycles[8192];
__device__ unsigned long long ws_pcycles[8192][4];     // producer wave 4: sum over half-tiles of (issue + stores done, x landed, barrier passed) [+ count]
#endif

template <int BITS, int MODE>
__global__ __launch_bounds__(WS_THREADS) void gemm_ws_kernel(GemmArgs a, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned char* const a_ring = smem;
    unsigned char* const w16 = smem + WS_NA * WS_ABYTES;
    unsigned char* const p_ring = w16 + 2 * WS_W16;
    unsigned char* const m_ring = p_ring + 2 * WS_PSLOT;
    constexpr int TB = 256 * BITS;                 // bytes of one packed 16 x 128 tile
    constexpr int NWI = BITS == 2 ? 1 : 2;         // LDS-DMA instructions per producer and group for the packed W
    constexpr int NXI = 8;                         // ... per producer and x half-tile

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r = lane & 15, o = lane >> 4;
    const int G = a.K >> 7, NH = 2 * G;

    int bm, bn;
    {   // bijective XCD remap + bands of 4 row-tiles (amq_gemm_ring.hip)
        const int T = ntm * ntn, b = (int)blockIdx.x;
        const int q = T >> 3, rem = T & 7, xcd = b & 7;
        const int v = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
        constexpr int GM = 4;
        const int width = GM * ntn, first = (v / width) * GM;
        const int gs = (ntm - first) < GM ? (ntm - first) : GM;
        bm = first + (v % width) % gs;
        bn = (v % width) / gs;
    }
    const int m0 = bm * WS_BM, n0 = bn * WS_BN;
    const int nblk_last = (a.N >> 4) - 1;
    const int cx = (r >> 1) & 7;
    const int aoff0 = r * 128 + (((0 + o) ^ cx) << 4);       // this lane's chunk of k-step 0 / 1 inside a 16-row (16-column) block image
    const int aoff1 = r * 128 + (((4 + o) ^ cx) << 4);
    // the fp16 W image swizzles by (column & 7): conflict-free for the consumers' ds_read_b128 (16-lane groups over two rows) AND for
    // the producers' ds_write_b128 (8-lane groups r = 0..7 / 8..15 of one chunk: 8 distinct positions; with (col >> 1) & 7 two-way)
    const int woff0 = r * 128 + (((0 + o) ^ (r & 7)) << 4);
    const int woff1 = r * 128 + (((4 + o) ^ (r & 7)) << 4);

    if (wave >= 4) {
        // ================================================================ producer
        const int p = wave - 4;
        const int cb0 = (n0 >> 4) + 2 * p;                   // this producer's two 16-column blocks
        unsigned aoff[NXI];
#pragma unroll
        for (int j = 0; j < NXI; ++j) {                      // DMA instruction i = p + 4 j fills rows 8 i .. 8 i + 7 of the half-tile image
            const int row = 8 * (p + 4 * j) + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            aoff[j] = ((unsigned)m * (unsigned)a.x_stride + chunk * 8) * 2u;
        }
        const unsigned char* const xbase = (const unsigned char*)a.x;
        const unsigned char* const qbase = (const unsigned char*)a.qweight;
        const unsigned char* const mbase = (const unsigned char*)a.meta;
        unsigned woff[NWI];
        if (BITS == 2) {
            const int cb = min(cb0 + (lane >> 5), nblk_last);
            woff[0] = (unsigned)cb * (unsigned)G * TB + (lane & 31) * 16;
        } else if (BITS == 3) {
#pragma unroll
            for (int j = 0; j < NWI; ++j) {
                int b = 1024 * j + 16 * lane;
                b = b < 2 * TB ? b : b - 512;
                const int cb = min(cb0 + b / TB, nblk_last);
                woff[j] = (unsigned)cb * (unsigned)G * TB + b % TB;
            }
        } else {
#pragma unroll
            for (int nb = 0; nb < NWI; ++nb) {
                const int cb = min(cb0 + nb, nblk_last);
                woff[nb] = (unsigned)cb * (unsigned)G * TB + lane * (4 * BITS);
            }
        }
        unsigned moff;
        {
            const int cb = min(cb0 + ((lane >> 4) & 1), nblk_last);
            moff = ((unsigned)cb * (unsigned)G * 16 + r) * 4;
        }
        const unsigned lds_a = lds0 + p * 1024, lds_p = lds0 + WS_NA * WS_ABYTES + 2 * WS_W16 + p * WS_PREG,
                       lds_m = lds0 + WS_NA * WS_ABYTES + 2 * WS_W16 + 2 * WS_PSLOT + p * 256;
        auto issue_a = [&](int h, int slot) {
            const int hc = h < NH ? h : NH - 1;              // past the end: harmless re-read into a consumed slot (uniform counts)
#pragma unroll
            for (int j = 0; j < NXI; ++j) ws_glds16(xbase + hc * 128, aoff[j], lds_a + slot * WS_ABYTES + j * 4096);
        };
        auto issue_w = [&](int g, int slot) {
            const int gc = g < G ? g : G - 1;
#pragma unroll
            for (int nb = 0; nb < NWI; ++nb) ws_glds16(qbase + (size_t)gc * TB, woff[nb], lds_p + slot * WS_PSLOT + nb * 1024);
            ws_glds4(mbase + (size_t)gc * 64, moff, lds_m + slot * WS_MSLOT);
        };
        uint32_t pw[2][BITS];
        SdMeta pm[2];
        auto read_packed = [&](int slot) {
            const unsigned char* wb = p_ring + slot * WS_PSLOT + p * WS_PREG + lane * (4 * BITS);
            const unsigned char* mb = m_ring + slot * WS_MSLOT + p * 256 + r * 4;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                if (BITS == 4) {
                    const u4 v = *(const u4*)(wb + nb * TB);
                    pw[nb][0] = v.x; pw[nb][1] = v.y; pw[nb][2] = v.z; pw[nb][3] = v.w;
                } else if (BITS == 2) {
                    const u2 v = *(const u2*)(wb + nb * TB);
                    pw[nb][0] = v.x; pw[nb][1] = v.y;
                } else {
#pragma unroll
                    for (int d = 0; d < 3; ++d) pw[nb][d] = *(const uint32_t*)(wb + nb * TB + 4 * d);
                }
                pm[nb] = sd_meta<BITS, MODE>(as_h2(*(const uint32_t*)(mb + nb * 64)));
            }
        };
        // (prologue) unpack half NS (0 / 1) of the group held in pw / pm and store its four fragments into fp16-W slot `slot`
        unsigned char* const wdst = w16 + (2 * p) * 2048;
        auto unpack_store = [&](auto ns_c, int slot) {
            constexpr int NS = decltype(ns_c)::value;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                h8 f[2];
                ws_for<8>([&](auto pp_c) {
                    constexpr int pp = decltype(pp_c)::value;
                    const h2 v = dequant_pair_sd<BITS, MODE, 8 * NS + pp>(pw[nb], pm[nb]);
                    f[pp >> 2][2 * (pp & 3)] = v.x;
                    f[pp >> 2][2 * (pp & 3) + 1] = v.y;
                });
                *(h8*)(wdst + slot * WS_W16 + nb * 2048 + woff0) = f[0];
                *(h8*)(wdst + slot * WS_W16 + nb * 2048 + woff1) = f[1];
            }
        };

        // One half-tile of producer work, behind barrier B_h (x(h - 1) and W16(h - 1) are consumed).  S = h & 1:
        //   S = 0 (h = 2g):     DMA pieces W(g + 1) [NWI + 1], x(2g + 2) [8];  unpack the SECOND half of group g  -> W16 slot 1
        //   S = 1 (h = 2g + 1): DMA pieces x(2g + 3) [8];                       unpack the FIRST half of group g + 1 -> W16 slot 0
        // The producer is the critical path (stamps, tools/ws_cycles.py): a DMA piece costs its issuer ~60 cycles (630 per half-tile), the
        // unpack arithmetic runs at HALF rate beside the consumer's MFMA stream (an MFMA holds the SIMD's vector issue for 8 of its 16
        // cycles: ~500 cycles for 70 instructions) and four ds_write_b128 from one wave take ~250 cycles to complete.  One wave issues
        // in order, so these add; what can overlap is the stores' completion: one fragment (4 pairs) is unpacked and stored behind
        // each of the first four pieces and the remaining pieces are issued on top of the stores in flight.
#ifdef AMQ_WS_CYCLES
        unsigned long long pc_[3] = {0, 0, 0};
#endif
        auto produce = [&](auto s_c, int g, int xslot) {
            constexpr int S = decltype(s_c)::value;
#ifdef AMQ_WS_CYCLES
            const unsigned long long ta_ = __builtin_amdgcn_s_memtime();
#endif
            constexpr int NS = 1 - S;
            constexpr int NDW = S == 0 ? NWI + 1 : 0;
            constexpr int NP = NDW + NXI;
            const int h = 2 * g + 2 + S;
            const int hc = h < NH ? h : NH - 1;              // past the end: harmless re-reads into consumed slots (uniform counts)
            const int gc = g + 1 < G ? g + 1 : G - 1;
            const unsigned char* const xsrc = xbase + hc * 128;
            const unsigned char* const qsrc = qbase + (size_t)gc * TB;
            const unsigned adst = lds_a + xslot * WS_ABYTES, pdst = lds_p + ((g + 1) & 1) * WS_PSLOT;
            if constexpr (S == 1) {
                AMQ_WAIT_VM("ws.w", NXI, "from=ws.p0:0");                    // packed W(g + 1) landed (x(2g + 2), the half before's NXI youngest pieces, may be in flight)
                read_packed((g + 1) & 1);
            }
            h8 f[2][2];
            ws_for<NP>([&](auto k_c) {
                constexpr int k = decltype(k_c)::value;
                if constexpr (k < NDW - 1) ws_glds16(qsrc, woff[k < NWI ? k : 0], pdst + k * 1024);
                else if constexpr (k == NDW - 1) ws_glds4(mbase + (size_t)gc * 64, moff, lds_m + ((g + 1) & 1) * WS_MSLOT);
                else {
                    constexpr int j = k - NDW;
#ifdef AMQ_WS_ABL_NOXDMA           /* timing-only ablation: the x image is not refreshed (a 4-byte DMA keeps the vmcnt counts) */
                    ws_glds4(mbase, moff, lds_m);
#else
                    ws_glds16(xsrc, aoff[j < NXI ? j : 0], adst + j * 4096);
#endif
                }
                if constexpr (k < 4) {                       // one fragment (4 weight pairs) behind each of the first four pieces, stored at once:
                    constexpr int nb = k >> 1, tp = k & 1;   // the stores complete under the remaining pieces' issue time
                    ws_for<4>([&](auto e_c) {
                        constexpr int pp = 4 * tp + decltype(e_c)::value;
#ifdef AMQ_WS_ABL_NODEQ            /* timing-only ablation: no unpack arithmetic */
                        const h2 v = {(_Float16)1, (_Float16)1};
#else
                        const h2 v = dequant_pair_sd<BITS, MODE, 8 * NS + pp>(pw[nb], pm[nb]);
#endif
                        f[nb][tp][2 * (pp & 3)] = v.x;
                        f[nb][tp][2 * (pp & 3) + 1] = v.y;
                    });
#ifndef AMQ_WS_ABL_NOSTORE         /* timing-only ablation: the fp16 W image is written by the prologue only (the unpack arithmetic goes too) */
                    *(h8*)(wdst + NS * WS_W16 + nb * 2048 + (tp ? woff1 : woff0)) = f[nb][tp];   // half-tile 2g + 1 + S lives in slot 1 - S = NS
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#ifdef AMQ_WS_CYCLES
            AMQ_WAIT_LGKM0("ws.cycles");
            const unsigned long long tb_ = __builtin_amdgcn_s_memtime();
#endif
            // x(h - 1) landed: all but this half-tile's own pieces; the fragment stores are in the LDS
            // (check_waits: this half-tile issued exactly NP pieces since the previous half's wait)
            if constexpr (S == 0) AMQ_WAIT_VM_LGKM0("ws.p0", NP, "from=ws.p1:%0 from=ws.pro2:%0");
            else AMQ_WAIT_VM_LGKM0("ws.p1", NP, "from=ws.p0:%0");
#ifdef AMQ_WS_CYCLES
            const unsigned long long tc_ = __builtin_amdgcn_s_memtime();
#endif
#ifndef AMQ_WS_ABL_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            WS_FENCE();
#ifdef AMQ_WS_CYCLES
            const unsigned long long td_ = __builtin_amdgcn_s_memtime();
            pc_[0] += tb_ - ta_; pc_[1] += tc_ - tb_; pc_[2] += td_ - tc_;
#endif
        };

        // DMA issue order: W(0) A(0) A(1) | W(1) A(2) | A(3) | W(2) A(4) | A(5) | ...
        issue_w(0, 0);
        issue_a(0, 0);
        issue_a(1, 1);
        AMQ_WAIT_VM("ws.pro", 2 * NXI, "from=entry:%1", "n"(NWI + 1 + 2 * NXI));   // W(0) landed
        read_packed(0);
        unpack_store(std::integral_constant<int, 0>{}, 0);
        AMQ_WAIT_VM_LGKM0("ws.pro2", NXI, "from=ws.pro:0");                  // x(0) landed, W16(0) written
        __builtin_amdgcn_s_barrier();                                         // B_0
        WS_FENCE();
        int sa = 0;
        for (int g = 0; g < G; ++g) {
            const int s1 = sa + 1 < WS_NA ? sa + 1 : sa + 1 - WS_NA;
            const int s2 = s1 + 1 < WS_NA ? s1 + 1 : s1 + 1 - WS_NA;
            produce(std::integral_constant<int, 0>{}, g, s2);                // ... B_(2g + 1)
            produce(std::integral_constant<int, 1>{}, g, sa);                // ... B_(2g + 2)
            sa = s2;
        }
        AMQ_WAIT_VM("ws.exit", 0, "");                                       // the trailing (clamped) DMAs must not outlive the workgroup
#ifdef AMQ_WS_CYCLES
        if (p == 0 && lane == 0 && blockIdx.x < 8192) { ws_pcycles[blockIdx.x][0] = pc_[0]; ws_pcycles[blockIdx.x][1] = pc_[1]; ws_pcycles[blockIdx.x][2] = pc_[2]; }
#endif
        return;
    }

    // ==================================================================== consumer
    const int rbase = 8 * (wave >> 1), cbase = 4 * (wave & 1);                // first row block / column block of this wave
    f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* const abase = a_ring + rbase * 2048;
    const unsigned char* const wbase = w16 + cbase * 2048;
    h8 xf[4][2];                                                              // x operands of 4 row blocks in flight
    struct WF { h8 f[4][2]; };                                                // W operands of one half-tile: [column block][k-step]
    auto load_w = [&](WF& w, int ws) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            w.f[cb][0] = *(const h8*)(wbase + ws * WS_W16 + cb * 2048 + woff0);
            w.f[cb][1] = *(const h8*)(wbase + ws * WS_W16 + cb * 2048 + woff1);
        }
    };
    auto half = [&](int slot, const WF& wc, WF& wn, int nslot, int nws) {
        const unsigned char* const ab = abase + slot * WS_ABYTES;
        const unsigned char* const nb_ = abase + nslot * WS_ABYTES;
        ws_for<8>([&](auto rb_c) {
            constexpr int rb = decltype(rb_c)::value;
            // x operands run 3 row blocks ahead (4-slot register ring): the LDS is busy with the DMA's and the producers' writes
            if constexpr (rb + 3 < 8) {
                xf[(rb + 3) & 3][0] = *(const h8*)(ab + (rb + 3) * 2048 + aoff0);
                xf[(rb + 3) & 3][1] = *(const h8*)(ab + (rb + 3) * 2048 + aoff1);
            }
            if constexpr (rb == 6) {
                // every operand of this half-tile has been received (the last read was issued at step 4): arrive at B_(h+1), then
                // request the next half-tile's W operands and first x operands, two row blocks (16 MFMAs) before their use
                AMQ_WAIT_LGKM0("ws.consumer");
#ifndef AMQ_WS_ABL_NOBAR           /* timing-only ablation: consumers do not wait for the producers */
                __builtin_amdgcn_s_barrier();
#endif
                WS_FENCE();
                load_w(wn, nws);
                xf[0][0] = *(const h8*)(nb_ + aoff0);
                xf[0][1] = *(const h8*)(nb_ + aoff1);
                xf[1][0] = *(const h8*)(nb_ + 2048 + aoff0);
                xf[1][1] = *(const h8*)(nb_ + 2048 + aoff1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (rb == 7) {
                xf[2][0] = *(const h8*)(nb_ + 4096 + aoff0);
                xf[2][1] = *(const h8*)(nb_ + 4096 + aoff1);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc.f[cb][ks], xf[rb & 3][ks], acc[rb][cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

#ifndef AMQ_WS_PRIO
#define AMQ_WS_PRIO 2
#endif
    __builtin_amdgcn_s_setprio(AMQ_WS_PRIO);                                  // the matrix pipe's only feeder on this SIMD
    WF wA, wB;
    __builtin_amdgcn_s_barrier();                                             // B_0
    WS_FENCE();
#ifdef AMQ_WS_CYCLES
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
    load_w(wA, 0);
    xf[0][0] = *(const h8*)(abase + aoff0);
    xf[0][1] = *(const h8*)(abase + aoff1);
    xf[1][0] = *(const h8*)(abase + 2048 + aoff0);
    xf[1][1] = *(const h8*)(abase + 2048 + aoff1);
    xf[2][0] = *(const h8*)(abase + 4096 + aoff0);
    xf[2][1] = *(const h8*)(abase + 4096 + aoff1);
    __builtin_amdgcn_sched_barrier(0);
    {
        int sa = 0;
        for (int g = 0; g < G; ++g) {
            const int s1 = sa + 1 < WS_NA ? sa + 1 : sa + 1 - WS_NA;
            const int s2 = s1 + 1 < WS_NA ? s1 + 1 : s1 + 1 - WS_NA;
            half(sa, wA, wB, s1, 1);
            half(s1, wB, wA, s2, 0);
            sa = s2;
        }
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef AMQ_WS_CYCLES
    if (wave == 0 && lane == 0 && blockIdx.x < 8192) ws_cycles[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0_;
#endif

    // ---- epilogue: acc[rb][cb][i] = y[m0 + 16 (rbase + rb) + r][n0 + 16 (cbase + cb) + 4 o + i]
    const _Float16* bias = (const _Float16*)a.bias;
    const _Float16* res = (const _Float16*)a.residual;
    const _Float16* gate = (const _Float16*)a.gate;
    _Float16* y = (_Float16*)a.y;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int gcb = (n0 >> 4) + cbase + cb;
        if (gcb > nblk_last) continue;
        const int n = gcb * 16 + 4 * o;
        h4 bv = {0, 0, 0, 0};
        if (bias) bv = *(const h4*)(bias + n);
        h4 rv[8];
        const _Float16* const side = res ? res : gate;
        if (side) {
#pragma unroll
            for (int rb = 0; rb < 8; ++rb) {
                int m = m0 + 16 * (rbase + rb) + r;
                m = m < a.M ? m : a.M - 1;
                rv[rb] = *(const h4*)(side + (size_t)m * a.y_stride + n);
            }
        }
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
            const int m = m0 + 16 * (rbase + rb) + r;
            h4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (_Float16)acc[rb][cb][i];
            if (bias) v = v + bv;
            if (res) v = rv[rb] + v;
            else if (gate) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gf = (float)rv[rb][i];
                    v[i] = (_Float16)(gf / (1.0f + __expf(-gf))) * v[i];
                }
            }
            if (m < a.M) *(h4*)(y + (size_t)m * a.y_stride + n) = v;
        }
    }
}

template <int BITS, int MODE>
hipError_t ws_launch(const GemmArgs& a, hipStream_t st) {
    StreamDevice sd_(st);                                  // attributes / CU counts of the stream's device
    auto k = gemm_ws_kernel<BITS, MODE>;
    static unsigned long long attr_done = 0;
    const hipError_t attr = ensure_dyn_lds(attr_done, (const void*)k, WS_LDS);
    if (attr != hipSuccess) return attr;
    const int ntm = (a.M + WS_BM - 1) / WS_BM, ntn = (a.N + WS_BN - 1) / WS_BN;
    hipLaunchKernelGGL(k, dim3(ntm * ntn), dim3(WS_THREADS), WS_LDS, st, a, ntm, ntn);
    return hipGetLastError();
}

}  // namespace

#ifdef AMQ_WS_CYCLES
extern "C" int amq_debug_ws_cycles(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ws_cycles), sizeof(unsigned long long) * (size_t)(n < 8192 ? n : 8192));
}
extern "C" int amq_debug_ws_pcycles(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ws_pcycles), 4 * sizeof(unsigned long long) * (size_t)(n < 8192 ? n : 8192));
}
#endif

// (the shape conditions of the ring kernel: gemm_ring_ok)
hipError_t launch_gemm_ws(const GemmArgs& a, hipStream_t st) {
    if (a.mode == MODE_HQQ) {
        if (a.bits == 4) return ws_launch<4, MODE_HQQ>(a, st);
        if (a.bits == 3) return ws_launch<3, MODE_HQQ>(a, st);
        return ws_launch<2, MODE_HQQ>(a, st);
    }
    if (a.bits == 4) return ws_launch<4, MODE_FMA>(a, st);
    if (a.bits == 3) return ws_launch<3, MODE_FMA>(a, st);
    return ws_launch<2, MODE_FMA>(a, st);
}

}  // namespace amq
