// amq_gemm_ring.hip -- y[M,N] = x[M,K] . W^T for MANY rows (batched prefill, BASELINE.json configs[3]), gfx950.
//
// Replaces gemm_w4a16_T2 (amq/kernel/ft/quantization_new/gemm/gemm_cuda.cu:746-927, dispatch :1005-1030) and the
// reference's torch unpack + matmul branch (hqq/backends/autogptq.py:245-283) in the MFMA-bound regime, for 2/3/4 bit.
//
// Why a second many-row kernel: gemm_kernel (amq_gemm.hip) double-buffers its x tile and drains the whole pipeline at
// every K step (`__syncthreads` = vmcnt(0) + barrier): profiles/r01_gemm_config4_pmc.txt shows the matrix pipe 41-56 %
// busy and a quarter of the wave cycles parked at that wait.  Here nothing in the K loop ever waits for vmcnt(0):
//
//   * workgroup = 8 waves, output tile 256 x 256; wave w owns ALL 256 rows x columns [32w, 32w + 32): every weight is
//     unpacked exactly once per workgroup (no redundant dequantization across waves, 256-row reuse of each unpacked
//     tile = 1/2 the VALU work per MFMA of the 128-row kernel) and needs no LDS image of dequantized W at all.
//   * EVERY global read is an LDS-DMA (global_load_lds): x half-tiles (256 rows x 64 k, 32 KB) into a 3-slot ring,
//     the packed 2/3/4-bit W tiles of the wave's own 32 columns (raw bytes, 8-16 KB per 128-k group and workgroup) and
//     their (scale, zero) into 2-slot rings.  No VGPR-destination load exists in the loop, so hipcc inserts no vmcnt
//     wait of its own; the waits are hand-counted: `vmcnt(4)` / `vmcnt(4 + W)` leave the next half-tile (and the next
//     group's packed W) in flight across a RAW `s_barrier` -- one barrier per 64 MFMAs per wave.
//   * LDS image of an x half-tile: 128-byte rows, 16-byte chunk c of row R at chunk position c ^ ((R >> 1) & 7).
//     The image is lane-linear per DMA instruction (8 whole rows), the swizzle is applied on the SOURCE address; the
//     ds_read_b128 of an MFMA operand (16 rows x 4 chunks) is bank-conflict-free (checked exhaustively against the
//     gfx950 lane groups, tools/lds_conflicts.py).
//   * MFMA roles are swapped against gemm_kernel: the unpacked W registers are the A operand, the x fragment the B
//     operand, so the accumulator holds y^T fragments -- four CONSECUTIVE output columns of one row per lane, stored as
//     one 8-byte store (the un-swapped form needs four 2-byte stores).  The operand register layouts are identical
//     for both roles of v_mfma_f32_16x16x32_f16, so the swap costs nothing.
//   * tile order: bijective XCD remap (blocks b, b + 8 share an L2) + bands of 4 row-tiles, so the 32 workgroups
//     resident on one XCD work on a 1024-row x 2048-column super-tile and share their x / W panels in that L2.
#include "amq_common.cuh"
#include "amq_kernels.h"

#include <utility>

namespace amq {

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int RG_THREADS = 512;
constexpr int RG_BN = 256;                        // (rows per tile: template parameter BM = 256, or 128 for launches of 50-150 big tiles)
constexpr int RG_NA = 3;                          // x ring slots (half-tiles of 64 k)
constexpr int RG_WSLOT = 16 * 1024;               // packed W of one 128-k group: 16 column blocks x 64 lanes x 4*BITS B <= 16 KiB
constexpr int RG_MSLOT = 8 * 256;                 // (scale, zero): 8 waves x 64 lanes x 4 B (upper 32 lanes: duplicates)
constexpr int rg_lds(int bm) { return RG_NA * bm * 128 + 2 * RG_WSLOT + 2 * RG_MSLOT; }   // 135,168 B at 256 rows: one workgroup per CU

// LDS-DMA as inline asm, not __builtin_amdgcn_global_load_lds: with the builtin in a kernel hipcc (ROCm 7.2) turns EVERY
// ds_read wait into `s_waitcnt lgkmcnt(0)` (it books the DMA as a flat access that may return out of order with LDS
// reads), which drains the operand look-ahead at every step; an asm DMA is invisible to that pass, the ds_read waits
// stay counted, and the DMA's own completion is hand-counted anyway (vmcnt).  Source = 64-bit scalar base + 32-bit lane
// offset (no 64-bit vector add per piece); M0 (the LDS destination, wave-uniform byte address) is set and LEFT: it is
// compiler-reserved, but nothing the compiler emits for these kernels reads it (gfx9+ LDS instructions do not), and saving /
// restoring it cost 2 scalar instructions per piece (hipcc warns about the clobber: -Wno-inline-asm for this file).
__device__ __forceinline__ void rg_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void rg_glds4(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
#ifdef AMQ_RING_ABL_NOBAR          /* timing-only ablation: results are wrong */
#define RG_BARRIER() do { } while (0)
#else
#define RG_BARRIER() __builtin_amdgcn_s_barrier()
#endif
// all but the youngest x half-tile (NXI DMA instructions of this wave) have landed.  (check_waits: the first half of a group issues exactly
// NWI + 1 + NXI pieces -- packed W and meta of the next group FIRST, then the x half-tile -- the second half exactly NXI: the NXI youngest
// operations at either wait are one x half-tile's)
#define RG_WAIT_X0() AMQ_WAIT_VM("ring.x0", NXI, "from=ring.x1:%1 from=ring.pro:0", "n"(NXI))
#define RG_WAIT_X1() AMQ_WAIT_VM("ring.x1", NXI, "from=ring.x0:%1", "n"(NWI + 1 + NXI))
template <int BITS, int MODE, int BM>
__global__ __launch_bounds__(RG_THREADS) void gemm_ring_kernel(GemmArgs a, int ntm, int ntn) {
    constexpr int RG_BM = BM;
    constexpr int RG_ABYTES = BM * 64 * 2;            // one x half-tile: 32 KiB (16 KiB at 128 rows)
    constexpr int NRB = BM / 16;                      // row blocks = steps of a half-tile
    constexpr int NXI = BM / 64;                      // LDS-DMA instructions per wave and x half-tile
    constexpr int PPS = BM == 256 ? 2 : 3;            // weight pairs unpacked per step (16 pairs over steps 2 ..)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address of smem
    unsigned char* const a_ring = smem;
    unsigned char* const w_ring = smem + RG_NA * RG_ABYTES;
    unsigned char* const m_ring = w_ring + 2 * RG_WSLOT;
    constexpr int TB = 256 * BITS;                // bytes of one packed 16 x 128 tile
    constexpr int NWI = BITS == 2 ? 1 : 2;        // LDS-DMA instructions per wave and group for the packed W
    constexpr int WREG = BITS == 3 ? 2048 : 2 * TB;   // bytes of a wave's packed-W region in a ring slot (3-bit: 1536 used + pad)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r = lane & 15, o = lane >> 4;
    const int G = a.K >> 7, NH = 2 * G;

    // ---- which tile
    int bm, bn;
    {
        const int T = ntm * ntn, b = (int)blockIdx.x;
        const int q = T >> 3, rem = T & 7, xcd = b & 7;
        const int v = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
#ifndef AMQ_RING_GM
#define AMQ_RING_GM 4
#endif
        constexpr int GM = AMQ_RING_GM;
        const int width = GM * ntn, first = (v / width) * GM;
        const int gs = (ntm - first) < GM ? (ntm - first) : GM;
        bm = first + (v % width) % gs;
        bn = (v % width) / gs;
    }
    const int m0 = bm * RG_BM, n0 = bn * RG_BN;
    const int nblk_last = (a.N >> 4) - 1;
    const int cb0 = (n0 >> 4) + 2 * wave;                  // this wave's two 16-column blocks: cb0, cb0 + 1

    // ---- DMA sources
    // x: instruction i (0..31) of a half-tile fills rows 8i .. 8i+7 (128 B each); wave w issues i = w + 8j.
    // lane l -> row 8i + (l >> 3), LDS chunk position l & 7  <-  global chunk (l & 7) ^ ((row >> 1) & 7)
    // (every DMA source = wave-uniform 64-bit base in SGPRs + a 32-bit byte offset per lane: no 64-bit vector adds in the loop;
    // gemm_ring_ok checks that x, the packed weights and the meta each span < 4 GiB)
    unsigned aoff[NXI];
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        const int row = 8 * (wave + 8 * j) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;                         // rows past M: computed, never stored
        aoff[j] = ((unsigned)m * (unsigned)a.x_stride + chunk * 8) * 2u;
    }
    const unsigned char* const xbase = (const unsigned char*)a.x;
    const unsigned char* const qbase = (const unsigned char*)a.qweight;
    const unsigned char* const mbase = (const unsigned char*)a.meta;
    // packed W + meta of the wave's own column blocks (blocks past N are clamped: computed, never stored)
    unsigned woff[NWI];
    if (BITS == 2) {                                       // one instruction: lanes 0-31 tile cb0, lanes 32-63 tile cb0 + 1
        const int cb = min(cb0 + (lane >> 5), nblk_last);
        woff[0] = (unsigned)cb * (unsigned)G * TB + (lane & 31) * 16;
    } else if (BITS == 3) {
        // two 768-byte tiles = 1536 contiguous LDS bytes, moved as raw bytes by two 16-byte-per-lane instructions (the
        // 12-byte form of the DMA does not lay lanes out 12 bytes apart): byte b = 1024 j + 16 lane of the image comes from
        // tile b / 768, offset b % 768; the last 32 lanes of j = 1 fall into the region's pad and re-read valid bytes
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
    const unsigned lds_a = lds0 + wave * 1024, lds_w = lds0 + RG_NA * RG_ABYTES + wave * WREG,
                   lds_m = lds0 + RG_NA * RG_ABYTES + 2 * RG_WSLOT + wave * 256;

    auto issue_a = [&](int h, int slot) {                  // 4 DMA instructions
        const int hc = h < NH ? h : NH - 1;                // past the end: harmless re-read into a consumed slot (keeps the counts uniform)
#pragma unroll
        for (int j = 0; j < NXI; ++j) rg_glds16(xbase + hc * 128, aoff[j], lds_a + slot * RG_ABYTES + j * 8192);
    };
    auto issue_w = [&](int g, int slot) {                  // NWI + 1 DMA instructions
        const int gc = g < G ? g : G - 1;
#pragma unroll
        for (int nb = 0; nb < NWI; ++nb) rg_glds16(qbase + (size_t)gc * TB, woff[nb], lds_w + slot * RG_WSLOT + nb * 1024);
        rg_glds4(mbase + (size_t)gc * 64, moff, lds_m + slot * RG_MSLOT);
    };

    f4 acc[NRB][2];
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    // operand read address of this lane inside a half-tile image: row 16*rb + r, chunk (4*tp + o) ^ ((r >> 1) & 7)
    const int cx = (r >> 1) & 7;
    const int aoff0 = r * 128 + (((0 + o) ^ cx) << 4);
    const int aoff1 = r * 128 + (((4 + o) ^ cx) << 4);

    // Unpacked weights of ONE half-tile (two 32-deep k-steps) for the wave's two column blocks = the MFMA A operands
    // [column block][k-step].  Two sets: while the MFMAs of half-tile h read one, the VALU fills the other for h + 1 in
    // their shadow (the matrix pipe leaves half of the issue cycles free) -- unpacking at the top of the half it
    // belongs to left every SIMD idle for ~500 cycles per group, both of its waves being in the same phase.
    struct WHalf { h8 f[2][2]; };
    struct WPacked { uint32_t w[2][BITS]; uint32_t mt[2]; SdMeta m[2]; };
    auto read_packed = [&](int slot, WPacked& pk) {         // LDS -> registers: packed words + (scale, zero) of both column blocks
        const unsigned char* wb = w_ring + slot * RG_WSLOT + wave * WREG + lane * (4 * BITS);
        const unsigned char* mb = m_ring + slot * RG_MSLOT + wave * 256 + r * 4;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            if (BITS == 4) {
                const u4 v = *(const u4*)(wb + nb * TB);
                pk.w[nb][0] = v.x; pk.w[nb][1] = v.y; pk.w[nb][2] = v.z; pk.w[nb][3] = v.w;
            } else if (BITS == 2) {
                const u2 v = *(const u2*)(wb + nb * TB);
                pk.w[nb][0] = v.x; pk.w[nb][1] = v.y;
            } else {
#pragma unroll
                for (int d = 0; d < 3; ++d) pk.w[nb][d] = *(const uint32_t*)(wb + nb * TB + 4 * d);
            }
            pk.mt[nb] = *(const uint32_t*)(mb + nb * 64);
        }
    };

    // One half-tile = 16 steps (row blocks); a step is one scheduling region (sched_barrier at its end), so the compiler
    // keeps what is put into it together and SIInsertWaitcnts emits COUNTED lgkmcnt waits:
    //   * the x operands of row block rb + 2 are requested (3-deep register ring: with one step of look-ahead both waves of
    //     a SIMD sat out the LDS latency, ~150 cycles, in front of every 64 cycles of MFMA),
    //   * one of the half's LDS-DMA instructions is issued (their issue cost, 60-180 cycles each, hides under MFMAs
    //     instead of piling up behind the barrier),
    //   * 4 MFMAs,
    //   * a slice of the NEXT half's unpack runs in the MFMAs' shadow (step 0: packed words LDS -> registers, step 1:
    //     meta scaling, steps 2-9: two weight pairs each).
    // S: 0 = first half of group g (DMA: packed W of g + 1, then x half-tile 2g + 2; unpack: second half of g),
    //    1 = second half (DMA: x half-tile 2g + 3; unpack: first half of g + 1, landed since this half's wait).
#ifndef AMQ_RING_RD
#define AMQ_RING_RD 3
#endif
    constexpr int RD = AMQ_RING_RD;                         // depth of the x operand register ring (row blocks in flight + 1)
    auto compute_half = [&](auto s_c, int slot, const WHalf& wc, int nslot, WHalf& wn, int dma_h, int dma_slot, int dma_g, int dma_wslot) {
        constexpr int S = decltype(s_c)::value;
        constexpr int NS = 1 - S;                           // which half of its group the NEXT half-tile is
        constexpr int NDW = S == 0 ? NWI + 1 : 0;           // W + meta DMA instructions of this half
        const unsigned char* ab = a_ring + slot * RG_ABYTES;
        const int hc = dma_h < NH ? dma_h : NH - 1;         // past the end: harmless re-read into a consumed slot (keeps the counts uniform)
        const int gc = dma_g < G ? dma_g : G - 1;
        const unsigned char* const xsrc = xbase + hc * 128;
        const unsigned char* const qsrc = qbase + (size_t)gc * TB;
        const unsigned adst = lds_a + dma_slot * RG_ABYTES, wdst = lds_w + dma_wslot * RG_WSLOT;
        h8 fx[RD][2];
        WPacked pk;
#pragma unroll
        for (int d = 0; d < RD - 1; ++d) {
            fx[d][0] = *(const h8*)(ab + d * 2048 + aoff0);
            fx[d][1] = *(const h8*)(ab + d * 2048 + aoff1);
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<NRB>([&](auto rb_c) {
            constexpr int rb = decltype(rb_c)::value;
#ifndef AMQ_RING_ABL_NOLDSX        /* timing-only ablation: operands are not re-read per row block */
            if constexpr (rb + RD - 1 < NRB) {
                fx[(rb + RD - 1) % RD][0] = *(const h8*)(ab + (rb + RD - 1) * 2048 + aoff0);
                fx[(rb + RD - 1) % RD][1] = *(const h8*)(ab + (rb + RD - 1) * 2048 + aoff1);
            }
#else
            if constexpr (rb == 0) {
                fx[RD - 1][0] = *(const h8*)(ab + (RD - 1) * 2048 + aoff0);
                fx[RD - 1][1] = *(const h8*)(ab + (RD - 1) * 2048 + aoff1);
            }
#endif
            if constexpr (rb < NDW) {                       // packed W first: it must have landed one half-tile before x(2g + 2) is needed
                if constexpr (rb == NDW - 1) rg_glds4(mbase + (size_t)gc * 64, moff, lds_m + dma_wslot * RG_MSLOT);
                else rg_glds16(qsrc, woff[rb], wdst + rb * 1024);
            } else if constexpr (rb < NDW + NXI) {
                constexpr int j = rb - NDW;
#ifdef AMQ_RING_ABL_NOXDMA         /* timing-only ablation: the x image is never refreshed (a 4-byte DMA keeps the vmcnt counts) */
                rg_glds4(mbase + (size_t)gc * 64, moff, lds_m + dma_wslot * RG_MSLOT);
#else
                rg_glds16(xsrc, aoff[j], adst + j * 8192);
#endif
            }
#ifdef AMQ_RING_PRIO
            __builtin_amdgcn_s_setprio(AMQ_RING_PRIO);
#endif
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc.f[nb][0], fx[rb % RD][0], acc[rb][nb], 0, 0, 0);
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc.f[nb][1], fx[rb % RD][1], acc[rb][nb], 0, 0, 0);
            }
#ifdef AMQ_RING_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            if constexpr (rb == 0) read_packed(nslot, pk);
            if constexpr (rb == 1) {
                pk.m[0] = sd_meta<BITS, MODE>(as_h2(pk.mt[0]));
                pk.m[1] = sd_meta<BITS, MODE>(as_h2(pk.mt[1]));
            }
#ifdef AMQ_RING_ABL_NODEQ          /* timing-only ablation: packed words reinterpreted, no unpack arithmetic */
            if constexpr (rb == 2) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const h2 v = as_h2((pk.w[nb][(tp * 4 + p) % BITS] & 0x03ff03ffu) | 0x3c003c00u);
                            wn.f[nb][tp][2 * p] = v.x; wn.f[nb][tp][2 * p + 1] = v.y;
                        }
            }
            if constexpr (false) {
#else
            if constexpr (rb >= 2 && PPS * (rb - 2) < 16) {
#endif
                static_for<PPS>([&](auto e_c) {
                    constexpr int idx = PPS * (rb - 2) + decltype(e_c)::value;       // 0..15: [column block][pair of the half]
                    if constexpr (idx < 16) {
                        constexpr int nb = idx >> 3, pp = idx & 7;
                        const h2 v = dequant_pair_sd<BITS, MODE, 8 * NS + pp>(pk.w[nb], pk.m[nb]);
                        wn.f[nb][pp >> 2][2 * (pp & 3)] = v.x;
                        wn.f[nb][pp >> 2][2 * (pp & 3) + 1] = v.y;
                    }
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- pipeline.  DMA issue order per wave: W(0) A(0) A(1) | W(1) A(2) | A(3) | W(2) A(4) | A(5) | ...  Every wait is
    // vmcnt(4): all but the youngest x half-tile have landed -- in particular the packed W of the NEXT group, which the
    // second half of a group already unpacks (it is wave-private: its own wait is all it needs, no barrier).
    issue_w(0, 0);
    issue_a(0, 0);
    issue_a(1, 1);
    WHalf w0, w1;
    AMQ_WAIT_VM("ring.pro", 2 * NXI, "from=entry:%1", "n"(NWI + 1 + 2 * NXI));      // W(0) landed (two x half-tiles may be in flight)
    {
        WPacked pk;
        read_packed(0, pk);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            h2 wv[16];
            dequant_lane_sd<BITS, MODE>(pk.w[nb], as_h2(pk.mt[nb]), wv);   // (only the first half's pairs are live)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                for (int p = 0; p < 4; ++p) { w0.f[nb][tp][2 * p] = wv[4 * tp + p].x; w0.f[nb][tp][2 * p + 1] = wv[4 * tp + p].y; }
        }
    }
    int sa = 0;                                            // ring slot of half-tile 2g
    for (int g = 0; g < G; ++g) {
        const int s1 = sa + 1 < RG_NA ? sa + 1 : sa + 1 - RG_NA;
        const int s2 = s1 + 1 < RG_NA ? s1 + 1 : s1 + 1 - RG_NA;
        // half-tile 2g.  sched_barrier: the previous half-tile's MFMAs (register-only, so not held by the asm's memory
        // clobber) must all be issued BEFORE this wave arrives at the barrier -- an MFMA issues only once its ds_read
        // operands have returned, so "every wave has passed compute_half" makes the vacated slot safe to refill (WAR).
        __builtin_amdgcn_sched_barrier(0);
        RG_WAIT_X0();
        RG_BARRIER();                                      // every wave's pieces landed; every wave is done reading half-tile 2g - 1
        asm volatile("" ::: "memory");                     // (compiler fence: no DMA issue / LDS read may move above the barrier)
        compute_half(std::integral_constant<int, 0>{}, sa, w0, g & 1, w1, 2 * g + 2, s2, g + 1, (g + 1) & 1);
        // half-tile 2g + 1
        RG_WAIT_X1();
        RG_BARRIER();
        asm volatile("" ::: "memory");
        compute_half(std::integral_constant<int, 1>{}, s1, w1, (g + 1) & 1, w0, 2 * g + 3, sa, 0, 0);
        sa = s2;
    }
    AMQ_WAIT_VM("ring.exit", 0, "");                       // the trailing (clamped) DMAs must not land after the workgroup has gone

    // ---- epilogue: acc[rb][nb][i] = y[m0 + 16 rb + r][n0 + 32 wave + 16 nb + 4 o + i]
    const _Float16* bias = (const _Float16*)a.bias;
    const _Float16* res = (const _Float16*)a.residual;
    const _Float16* gate = (const _Float16*)a.gate;
    _Float16* y = (_Float16*)a.y;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        if (cb0 + nb > nblk_last) continue;
        const int n = (cb0 + nb) * 16 + 4 * o;
        h4 bv = {0, 0, 0, 0};
        if (bias) bv = *(const h4*)(bias + n);
        h4 rv[NRB];
        const _Float16* const side = res ? res : gate;     // (a launch carries a residual or a gate, not both: checked by the caller)
        if (side) {                                        // all residual / gate loads of the column block first, ONE wait (not one per row block)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                int m = m0 + 16 * rb + r;
                m = m < a.M ? m : a.M - 1;
                rv[rb] = *(const h4*)(side + (size_t)m * a.y_stride + n);
            }
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int m = m0 + 16 * rb + r;
            h4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (_Float16)acc[rb][nb][i];
            if (bias) v = v + bv;
            if (res) v = rv[rb] + v;
            else if (gate) {                               // act = fp16(silu(gate)) * up   (silu_mul_kernel's expression)
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

template <int BITS, int MODE, int BM>
static hipError_t ring_launch_bm(const GemmArgs& a, hipStream_t st) {
    StreamDevice sd_(st);                                  // attributes / CU counts of the stream's device
    auto k = gemm_ring_kernel<BITS, MODE, BM>;
    static unsigned long long attr_done = 0;
    const hipError_t attr = ensure_dyn_lds(attr_done, (const void*)k, rg_lds(BM));
    if (attr != hipSuccess) return attr;
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + RG_BN - 1) / RG_BN;
    hipLaunchKernelGGL(k, dim3(ntm * ntn), dim3(RG_THREADS), rg_lds(BM), st, a, ntm, ntn);
    return hipGetLastError();
}

template <int BITS, int MODE>
static hipError_t ring_launch(const GemmArgs& a, hipStream_t st, int bm) {
    if (bm == 128) return ring_launch_bm<BITS, MODE, 128>(a, st);
    return ring_launch_bm<BITS, MODE, 256>(a, st);
}

bool gemm_ring_ok(const GemmArgs& a) {
    // 8-byte row-segment stores / residual loads need 4-element alignment of every row; the DMA sources are addressed as
    // scalar base + 32-bit lane offset, so x, the packed weights and the meta must each span < 4 GiB (else: the tiled kernel)
    const unsigned long long lim = 1ull << 32;
    const unsigned long long xspan = ((unsigned long long)(a.M - 1) * (unsigned long long)a.x_stride + (unsigned long long)a.K) * 2ull;
    const unsigned long long wspan = (unsigned long long)a.N * (unsigned long long)a.K * (unsigned long long)a.bits / 8ull;
    return (a.y_stride & 3) == 0 && a.splits <= 1 && a.K >= 128 && a.M >= 1 && xspan < lim && wspan < lim;
}

// Which many-row kernel serves a launch: 256 / 128 = this kernel with that many rows per tile, -1 = the wave-specialised kernel
// (amq_gemm_ws.hip, 256 x 128 tiles), 0 = the launch is too small for any of them.  All tiles of a launch cost the same, so a launch
// runs in rounds of 256 workgroups: a tile shape is scored by its last-round fill, t / (256 ceil(t / 256)), times its measured efficiency
// at full fill relative to the 256 x 256 ring tile -- 0.93 for the wave-specialised 256 x 128 tile, 0.85 for 128-row ring tiles (twice
// the unpack work per MFMA) -- and needs >= 150 tiles to be considered (profiles/r02_gemm_routes.txt: 128 tiles lose to the round-1
// kernel, 160+ win).  The three kernels accumulate every output in the same order, so the choice never changes a bit.  Round 3
// (profiles/r03_gemm_mid_sweep.txt, 3-bit TFLOP/s ring | wave-specialised): wherever the ring kernel would fall back to 128-row tiles,
// or its 256-row tiles leave the last round mostly empty, the 256 x 128 tile wins by 4-9 % (4096x11008 M = 2048: 1099 | 1169;
// 5120x13824 M = 4096: 1070 | 1135; 13824x5120 M = 512: 927 | 966); at equal fill the ring kernel wins by 5-15 %.
// workgroup slots of one round = CUs of the current device (256 on an MI355X; smaller on a partitioned part)
static long plan_slots() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cus[dev & 63];
    if (c == 0) {
        int v = 0;
        c = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return c;
}

int gemm_many_rows_plan(int M, int N) {
    const long nt = (N + RG_BN - 1) / RG_BN, ntw = (N + 127) / 128;
    const long t256 = (long)((M + 255) / 256) * nt, t128 = (long)((M + 127) / 128) * nt, tws = (long)((M + 255) / 256) * ntw;
    const long S = plan_slots();
    auto fill = [S](long t) { return (double)t / ((double)S * (double)((t + S - 1) / S)); };
    const double s256 = t256 >= 150 ? fill(t256) : 0.0, s128 = t128 >= 150 ? 0.85 * fill(t128) : 0.0;
#ifdef AMQ_PLAN_NO_WS               /* A/B build: the round-2 policy (ring tiles only) */
    const double sws = 0.0;
#else
    const double sws = tws >= 150 ? 0.93 * fill(tws) : 0.0;
#endif
    if (s256 == 0.0 && s128 == 0.0 && sws == 0.0) return 0;
    if (s256 >= s128 && s256 >= sws) return 256;
    return sws >= s128 ? -1 : 128;
}

// the ring kernel's own choice of tile rows (forced routes, tools)
int gemm_ring_rows(int M, int N) {
    const long nt = (N + RG_BN - 1) / RG_BN;
    const long t256 = (long)((M + 255) / 256) * nt, t128 = (long)((M + 127) / 128) * nt;
    const long S = plan_slots();
    auto fill = [S](long t) { return (double)t / ((double)S * (double)((t + S - 1) / S)); };
    const double s256 = t256 >= 150 ? fill(t256) : 0.0, s128 = t128 >= 150 ? 0.85 * fill(t128) : 0.0;
    if (s256 == 0.0 && s128 == 0.0) return 0;
    return s256 >= s128 ? 256 : 128;
}

hipError_t launch_gemm_ring(const GemmArgs& a, hipStream_t st, int bm) {
    if (bm != 128 && bm != 256) bm = gemm_ring_rows(a.M, a.N);
    if (bm == 0) bm = 128;                                  // (forced onto a small launch: GEMM_ROUTE_RING in tests / tools)
    if (a.mode == MODE_HQQ) {
        if (a.bits == 4) return ring_launch<4, MODE_HQQ>(a, st, bm);
        if (a.bits == 3) return ring_launch<3, MODE_HQQ>(a, st, bm);
        return ring_launch<2, MODE_HQQ>(a, st, bm);
    }
    if (a.bits == 4) return ring_launch<4, MODE_FMA>(a, st, bm);
    if (a.bits == 3) return ring_launch<3, MODE_FMA>(a, st, bm);
    return ring_launch<2, MODE_FMA>(a, st, bm);
}

}  // namespace amq
