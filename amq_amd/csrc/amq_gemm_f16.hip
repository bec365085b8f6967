// amq_gemm_f16.hip -- y[M,N] = x[M,K] . W^T with W given as fp16 [N,K]: the MFMA-bound end of the batched path (gfx950).
//
// In the MFMA-bound regime (BASELINE.json configs[3]: 16 x 2048 prompt rows) the in-loop unpack of amq_gemm_ring.hip is 1.2 VALU
// instructions per MFMA of pure overhead (profiles/r02_gemm_ring_pmc.txt: "no unpack arithmetic: +11 %"), and the reference itself
// does not fuse there: GPTQLinear.forward unpacks once and calls matmul from 128 rows on (hqq/backends/autogptq.py:245-283),
// gemm_w4a16_T2 is its fused counterpart (ft/quantization_new/gemm/gemm_cuda.cu:746-927).  This file is the hand-written
// "matmul": amq_dequantize_f16 writes the exact fp16 weights once per launch (141 MB for 13824 x 5120: ~40 us against a 3.4 ms
// GEMM), this kernel multiplies them with NO vector-ALU work in its K loop.
//
// Structure (8 waves, 256 x 256 output tile, 64-deep K tiles, 128 KiB of LDS = 2 K-tiles x 4 staging units):
//   * waves as 2 (rows) x 4 (columns): wave (wr, wc) owns rows [128 wr, +128) x columns [64 wc, +64) = 8 x 4 accumulator
//     fragments (128 registers); per K-tile it runs TWO phases of 32 MFMAs:
//         X: quadrants (a0, b0), (a0, b1)   reads a0 (8 ds_read_b128) + b0 (4) + b1 (4)    a0 / a1 = first / second 64 rows of the wave
//         Y: quadrants (a1, b1), (a1, b0)   reads a1 (8)                                   b0 / b1 = first / second 32 columns
//     24 operand reads per 64 MFMAs (the ring kernel: 36), no other LDS traffic, no VALU.
//   * the two waves of a SIMD (wr = 0 and wr = 1) run one barrier apart: while one issues its 32 MFMAs (s_setprio 1) the other
//     issues its operand reads and its LDS-DMA pieces, then they swap -- the matrix pipe always has a wave that does nothing else.
//   * staging units are cut by WHEN they are read, not by where they lie: AF = the a0 rows of both wave rows, AS = the a1 rows,
//     BF = the b0 columns of all four wave columns, BS = the b1 columns (128 rows x 64 k = 16 KiB each).  AF, BF, BS are read in
//     X only, AS in Y only, so a slot can be refilled from the next phase on and the stream of LDS-DMA runs seven units ahead of
//     its consumption:  Y(t) issues AF, BF and the first half of BS(t + 2), X(t + 1) the second half of BS(t + 2) and AS(t + 2) (5 / 3
//     pieces per wave: 6 / 2 made the group behind Y the late arriver of its hand-over), each followed by ONE counted wait that
//     leaves the youngest eight (X) / seven (Y) pieces in flight: behind X everything up to AS(t) has landed (read in Y), behind Y
//     everything up to BS(t + 1) (read in the next X); every piece has a phase and a half or more (>= 1500 matrix-pipe cycles) to land.  A unit is read one phase or
//     more after the wait + barrier that covers it and refilled one phase or more after a barrier that every reading wave reached
//     with its reads retired (lgkmcnt(0) BEFORE the barrier): both orders hold by construction.
//   * PERSISTENT: one workgroup per CU walks the tiles b, b + grid, ...; the DMA stream does not stop at a tile's end -- the last
//     two K-tiles of a tile already stage the first two of the next, so no tile but a workgroup's first pays a prologue, and the
//     epilogue stores of a tile run with the next tile's operands in flight.
//   * LDS image of a unit: 128-byte rows, 16-byte chunk c of row R at chunk position c ^ ((R >> 1) & 7) (the ring kernel's image:
//     lane-linear per DMA instruction, swizzle on the SOURCE address, ds_read_b128 of a 16 x 32 operand conflict-free).
//   * LDS-DMA as `buffer_load_dwordx4 ... offen lds` (inline asm, M0 = destination set and left): buffer resource in SGPRs, 32-bit
//     lane offset, K-tile advance in the SGPR offset -- 1-2 % faster here than the global_load_lds form of the ring kernel.
//   * MFMA roles as in the ring kernel: W fragment = A operand, x fragment = B operand, so a lane holds four CONSECUTIVE output
//     columns of one row (one 8-byte store); bias / residual / SiLU-gate epilogues as there.
//   * tile order: bijective XCD remap + bands of 4 row-tiles (x / W panels shared in an XCD's L2).
//
// Measured and NOT adopted (profiles/r04_gemm_f16pp.txt): four phases of 16 MFMAs (the same speed), the operand wait behind the barrier
// instead of in front of it (same), accumulators pinned to AGPRs by inline-asm MFMAs (same), s_setprio 0 / 3 for the MFMA burst (same),
// one phase of 64 MFMAs per K-tile (needs a third LDS set: a refilled set would be overwritten under the other wave group's reads),
// half of the waves issuing their pieces BEFORE their operand reads (hand-over gap 59 -> 80-98 cycles: slower), the pieces placed between the
// MFMAs of the issuing wave's OWN burst instead of its read segment (one per four MFMAs: the burst grows from 593 to 627 cycles, -2.5 %),
// and hipBLASLt's wave geometry -- four waves, one per SIMD, 128 x 128 each (256 accumulators pinned to AGPRs, 0.25 operand reads per MFMA), the
// whole K-tile written out as ONE hand-ordered inline-asm stream of 128 MFMAs, 32 ds_read_b128, 16 pieces, one counted wait and one barrier:
// 1219-1223 vs 1270-1276 TFLOP/s (-4 %), not repeatable across launches; rebuilt with compiler-tracked operand reads and pinned order (correct, repeatable):
// 1140-1153 vs 1310-1326.  tools/ubench/mfma_shadow.hip says why (profiles/r04_mfma_shadow.txt): a SIMD's lone wave has three issue slots per MFMA, and a 1 KiB
// LDS-DMA piece holds its issue for ~27 cycles = 11-17 cycles of matrix-pipe time per piece (9-13 % of a K-tile), whatever the spacing; with two waves per SIMD the
// partner's pieces issue beside the MFMAs for free -- this kernel's structure.
#include "amq_common.cuh"
#include "amq_kernels.h"

#include <utility>

namespace amq {

namespace {

constexpr int PP_THREADS = 512;
constexpr int PP_BM = 256, PP_BN = 256;             // (K tiles are 64 deep)
constexpr int PP_UNIT = 128 * 128;                  // one staging unit: 128 rows x 64 halves
constexpr int PP_LDS = 8 * PP_UNIT;                 // 131,072 B: one workgroup per CU
#ifdef AMQ_PP_TRACE
constexpr int PP_LDS_ALLOC = PP_LDS + 32768;
#else
constexpr int PP_LDS_ALLOC = PP_LDS;
#endif
enum { U_AF = 0, U_BF = 1, U_BS = 2, U_AS = 3 };    // slot order inside a K-tile's set

typedef int pp_i4 __attribute__((ext_vector_type(4)));
// one 1 KiB piece: lane l's 16 bytes from (resource base + voff + soff) to LDS byte lds_dst + 16 l
__device__ __forceinline__ void pp_blds16(pp_i4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ pp_i4 pp_make_rsrc(const void* base) {      // raw buffer over [base, base + 4 GiB), dword format
    const unsigned long long b = (unsigned long long)base;
    pp_i4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff);
    r.z = -1;
    r.w = 0x00020000;
    return r;
}

#ifdef AMQ_PP_ABL_NOBAR            /* timing-only ablation: results are wrong */
#define PP_BARRIER() do { } while (0)
#else
#define PP_BARRIER() __builtin_amdgcn_s_barrier()
#endif

}  // namespace

#ifdef AMQ_PP_STAMP
__device__ unsigned amq_pp_stamp_buf[256 * 8 * 16];
#endif
#ifdef AMQ_PP_TRACE
__device__ unsigned long long amq_pp_trace_buf[16 * 8 * 128 * 4];
#endif

typedef __bf16 pp_b8 __attribute__((ext_vector_type(8)));
typedef __bf16 pp_b4 __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ f4 pp_mfma(const h8& w, const h8& x, const f4& c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pp_b8, w), __builtin_bit_cast(pp_b8, x), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0);
}

struct GemmF16Args {
    const void* x; const void* w; const void* bias; const void* residual; const void* gate; void* y;
    int M, N, K, x_stride, y_stride;
};

// BF = false: fp16 operands and output (the product path).  BF = true: the same stream of bytes multiplied as bfloat16 (v_mfma_f32_16x16x32_bf16 has the
// fp16 instruction's operand map), output / bias / residual bfloat16 -- the batched end of the optional bf16 entry points (amq_bf16.hip).
template <bool BF>
__device__ __forceinline__ void gemm_pp_body(const GemmF16Args& a, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, o = lane >> 4;
    const int T = a.K >> 6;                                // K tiles (K % 128 == 0: T is even)
    const int NT = ntm * ntn;

    // tile b of the launch order (the ring kernel's: bijective XCD remap, bands of 4 row-tiles) -> first row / column
    auto tile_origin = [&](int b, int& m0, int& n0) {
        const int q = NT >> 3, rem = NT & 7, xcd = b & 7;
        const int v = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
        constexpr int GM = 4;
        const int width = GM * ntn, first = (v / width) * GM;
        const int gs = (ntm - first) < GM ? (ntm - first) : GM;
        m0 = (first + (v % width) % gs) * PP_BM;
        n0 = ((v % width) / gs) * PP_BN;
    };
    // DMA sources of a tile.  A unit = 2 instructions per thread; instruction j: thread t -> unit row 64 j + (t >> 3), LDS chunk position
    // t & 7  <-  global chunk (t & 7) ^ ((row >> 1) & 7).  Unit row i of AF / AS = x row m0 + 128 (i >> 6) + 64 [AS] + (i & 63);
    // of BF / BS = W row n0 + 64 (i >> 5) + 32 [BS] + (i & 31).  Rows past M / N are clamped: computed, never stored.
    auto tile_offsets = [&](int m0, int n0, unsigned (&ao)[2][2], unsigned (&bo)[2][2]) {
        const int t = (int)threadIdx.x;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 64 * j + (t >> 3);
            const int chunk = (t & 7) ^ ((i >> 1) & 7);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                int m = m0 + 128 * (i >> 6) + 64 * s + (i & 63);
                m = m < a.M ? m : a.M - 1;
                ao[s][j] = ((unsigned)m * (unsigned)a.x_stride + chunk * 8) * 2u;
                int n = n0 + 64 * (i >> 5) + 32 * s + (i & 31);
                n = n < a.N ? n : a.N - 1;
                bo[s][j] = ((unsigned)n * (unsigned)a.K + chunk * 8) * 2u;
            }
        }
    };

    int tb = (int)blockIdx.x;                              // the tile being multiplied
    int m0, n0;
    tile_origin(tb, m0, n0);
    unsigned aoff[2][2], boff[2][2];                       // [second half?][j]: sources of the tile the DMA stream is staging
    tile_offsets(m0, n0, aoff, boff);
    const unsigned lds_w = lds0 + wave * 1024;             // this wave's 1 KiB of every DMA instruction's 8 KiB
    const pp_i4 xrsrc = pp_make_rsrc(a.x), wrsrc = pp_make_rsrc(a.w);

    // unit U of K-tile kt of the staged tile (kt >= T: the stream has moved on to the next tile, see the loop) into set `set`
    auto issue = [&](int kt, int set, int U, int half = 3) {      // half: bit j = instruction j of the unit (BS is issued in two places)
#ifdef AMQ_PP_ABL_NODMA            /* timing-only ablation: nothing is staged at all */
        return;
#endif
#ifdef AMQ_PP_ABL_L2HOT            /* timing-only ablation: every piece re-reads one of four K-tiles (always in L2): issue cost without miss latency */
        const int kc = kt & 3;
#else
        const int kc = kt < T ? kt : kt - T;
#endif
        const unsigned dst = lds_w + (unsigned)(set * 4 + U) * PP_UNIT;
        if (U == U_AF || U == U_AS) {
            const int s = U == U_AS;
            pp_blds16(xrsrc, aoff[s][0], (unsigned)kc * 128u, dst);
            pp_blds16(xrsrc, aoff[s][1], (unsigned)kc * 128u, dst + 8192);
        } else {
            const int s = U == U_BS;
            if (half & 1) { pp_blds16(wrsrc, boff[s][0], (unsigned)kc * 128u, dst); }
            if (half & 2) { pp_blds16(wrsrc, boff[s][1], (unsigned)kc * 128u, dst + 8192); }
        }
    };

    f4 acc[8][4];

    // operand read offsets inside a unit: row 16 b + r, chunk (4 ks + o) ^ ((r >> 1) & 7)
    const int cx = (r >> 1) & 7;
    const int ko0 = r * 128 + (((0 + o) ^ cx) << 4);
    const int ko1 = r * 128 + (((4 + o) ^ cx) << 4);
    const int a_row0 = (64 * wr) * 128;                    // first of the wave's 4 row blocks in AF / AS
    const int b_row0 = (32 * wc) * 128;                    // first of the wave's 2 column blocks in BF / BS

    h8 af[4][2];                                           // x fragments of the current row half   [row block][k-step]
    h8 b0f[2][2], b1f[2][2];                               // W fragments of the two column halves  [column block][k-step]

    auto read_a = [&](int set, int U) {
#ifdef AMQ_PP_ABL_NOLDS            /* timing-only ablation: operands are read once */
        if (set == 1) return;
#endif
        const unsigned char* ub = smem + (set * 4 + U) * PP_UNIT + a_row0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            af[b][0] = *(const h8*)(ub + b * 2048 + ko0);
            af[b][1] = *(const h8*)(ub + b * 2048 + ko1);
        }
    };
    auto read_b = [&](int set, int U, h8 (&bf)[2][2]) {
#ifdef AMQ_PP_ABL_NOLDS
        if (set == 1) return;
#endif
        const unsigned char* ub = smem + (set * 4 + U) * PP_UNIT + b_row0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            bf[b][0] = *(const h8*)(ub + b * 2048 + ko0);
            bf[b][1] = *(const h8*)(ub + b * 2048 + ko1);
        }
    };
    // 16 MFMAs of one quadrant: accumulator rows 4 ah .. 4 ah + 3, columns 2 bh, 2 bh + 1
    auto mfma_quadrant = [&](auto ah_c, auto bh_c, const h8 (&bf)[2][2]) {
        constexpr int ah = decltype(ah_c)::value, bh = decltype(bh_c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[4 * ah + b][2 * bh + c] = pp_mfma<BF>(bf[c][ks], af[b][ks], acc[4 * ah + b][2 * bh + c]);
    };

#ifdef AMQ_PP_TRACE                /* diagnostic build: MFMA-burst begin / end stamps of the first 128 phases, per wave (tools/f16pp_trace.py) */
    int tr_n = 0;
    unsigned long long tr_t3 = 0, tr_t3p = 0, tr_t4 = 0, tr_t5 = 0;      // end of the read segment (just before its barrier), burst begin, burst end
#define PP_TRACE_READY() do { __builtin_amdgcn_sched_barrier(0); tr_t3 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_TRACE_BEGIN() do { __builtin_amdgcn_sched_barrier(0); tr_t3p = tr_t3; tr_t4 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_TRACE_END() do { __builtin_amdgcn_sched_barrier(0); tr_t5 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
// (the stamps are sampled asynchronously; they are written out in the NEXT phase's read segment, where their latency hides)
#define PP_TRACE_FLUSH() do { if (tr_t5) { if (tr_n < 128 && lane == 0) { unsigned long long* d_ = (unsigned long long*)(smem + PP_LDS + (wave * 128 + tr_n) * 32); \
        d_[0] = tr_t4; d_[1] = tr_t5; d_[2] = tr_t3p; } ++tr_n; } } while (0)
#else
#define PP_TRACE_READY() do { } while (0)
#define PP_TRACE_BEGIN() do { } while (0)
#define PP_TRACE_END() do { } while (0)
#define PP_TRACE_FLUSH() do { } while (0)
#endif
#ifdef AMQ_PP_STAMP                /* diagnostic build: per-wave cycle accounting of the loop (tools/f16pp_stamps.py) */
    unsigned long long st_t[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = 0;
    unsigned st_acc[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
#define PP_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); st_t[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_STAMP_ACC(P) do { \
        st_acc[P][0] += (unsigned)(st_t[1] - st_t[0]); st_acc[P][1] += (unsigned)(st_t[2] - st_t[1]); st_acc[P][2] += (unsigned)(st_t[3] - st_t[2]); \
        st_acc[P][3] += (unsigned)(st_t[4] - st_t[3]); st_acc[P][4] += (unsigned)(st_t[5] - st_t[4]); \
        if (st_prev) st_acc[P][5] += (unsigned)(st_t[0] - st_prev); st_prev = st_t[5]; } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#define PP_STAMP_ACC(P) do { } while (0)
#endif

    // One phase: [operand reads + LDS-DMA pieces + the counted wait] lgkmcnt(0) | barrier | 32 MFMAs | barrier.
    // The sched_barriers pin that order: register-only MFMAs are not held by an asm memory clobber (amq_gemm_ring.hip, finding 2).
    auto phase = [&](auto p_c, int set, int kt) {
        constexpr int P = decltype(p_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(0);
        PP_TRACE_FLUSH();
        // (3 pieces behind X's 16 operand reads, 5 behind Y's 8 -- the second half of BS is issued one phase later: with 2 / 6 the four waves of a group queue
        //  24 pieces at the texture addresser behind Y and arrive late at the hand-over; 4 / 4 is no better than 2 / 6: profiles/r05_gemm_f16pp.txt)
        if constexpr (P == 0) { read_b(set, U_BF, b0f); read_b(set, U_BS, b1f); read_a(set, U_AF); issue(kt + 1, set ^ 1, U_BS, 2); issue(kt + 1, set ^ 1, U_AS); }
        if constexpr (P == 1) { read_a(set, U_AS); issue(kt + 2, set, U_AF); issue(kt + 2, set, U_BF); issue(kt + 2, set, U_BS, 1); }
        PP_STAMP(1);
        // (check_waits: a P = 0 phase issues exactly 3 pieces since the previous phase's wait, a P = 1 phase exactly 5 -- what 8 = 3 + 5 and 7 = 5 + 2 count)
        if constexpr (P == 0) AMQ_WAIT_VM("f16pp.x", 8, "from=f16pp.y:3 from=f16pp.pro:3 from=f16pp.drain:3");   // X's three and Y's five youngest stay in flight: AS(kt) has landed
        else AMQ_WAIT_VM("f16pp.y", 7, "from=f16pp.x:5");                           // Y's five and AS(kt + 1): AF, BF, BS(kt + 1) have landed
        PP_STAMP(2);
        AMQ_WAIT_LGKM0("f16pp.operands");                               // this phase's operands are in registers; the slots they came from may be refilled
        PP_STAMP(3);
        PP_TRACE_READY();
        __builtin_amdgcn_sched_barrier(0);
        PP_BARRIER();
        PP_STAMP(4);
        PP_TRACE_BEGIN();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (P == 0) {
            mfma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, b0f);
            mfma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, b1f);
        } else {
            mfma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, b1f);
            mfma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, b0f);
        }
        __builtin_amdgcn_s_setprio(0);
        PP_TRACE_END();
        PP_STAMP(5);
        PP_STAMP_ACC(P);
        __builtin_amdgcn_sched_barrier(0);
        PP_BARRIER();
    };

    // ---- prologue (once per workgroup): K-tile 0 complete, AF / BF / BS of K-tile 1 -- the seven units of lead the loop keeps
    issue(0, 0, U_AF); issue(0, 0, U_BF); issue(0, 0, U_BS); issue(0, 0, U_AS);
    issue(1, 1, U_AF); issue(1, 1, U_BF); issue(1, 1, U_BS, 1);
    AMQ_WAIT_VM("f16pp.pro", 5, "from=entry:13");                       // K-tile 0 has landed (this wave's pieces: the 8 oldest of 13)
    PP_BARRIER();                                                       // ... everybody's
    if (wr == 1) PP_BARRIER();                                          // the second wave of every SIMD runs one barrier behind

    const _Float16* bias = (const _Float16*)a.bias;
    const _Float16* res = (const _Float16*)a.residual;
    const _Float16* gate = (const _Float16*)a.gate;
    _Float16* y = (_Float16*)a.y;
    const _Float16* const side = res ? res : gate;

    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        // the tile the stream stages next: this workgroup's next one, or (behind its last) this one again -- harmless re-reads that keep
        // the counts uniform
        const int nb = tb + (int)gridDim.x < NT ? tb + (int)gridDim.x : tb;
        int nm0, nn0;
        tile_origin(nb, nm0, nn0);

        for (int kt = 0; kt < T; kt += 2) {
            phase(std::integral_constant<int, 0>{}, 0, kt);             // X(kt) issues AS(kt + 1): the last piece of THIS tile when kt = T - 2
            if (kt == T - 2) tile_offsets(nm0, nn0, aoff, boff);        // from here on the stream stages the next tile (K-tiles T, T + 1 = its 0, 1)
            phase(std::integral_constant<int, 1>{}, 0, kt);
            phase(std::integral_constant<int, 0>{}, 1, kt + 1);
            phase(std::integral_constant<int, 1>{}, 1, kt + 1);
        }

        // ---- epilogue of this tile (the next tile's first K-tiles are landing meanwhile):
        // acc[b][c][i] = y[m0 + 128 wr + 16 b + r][n0 + 64 wc + 16 c + 4 o + i]
        if constexpr (BF) {
            // bfloat16: y = bf16(acc), + bias and + residual as separate bf16 adds (each the fp32 sum of two bf16 values, rounded to nearest even)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int n = n0 + 64 * wc + 16 * c + 4 * o;
                if (n >= a.N) continue;
                pp_b4 bv = {0, 0, 0, 0};
                if (bias) bv = *(const pp_b4*)((const __bf16*)a.bias + n);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int m = m0 + 128 * wr + 16 * b + r;
                    if (m >= a.M) continue;
                    pp_b4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (__bf16)acc[b][c][i];
                    if (bias) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = (__bf16)((float)v[i] + (float)bv[i]);
                    }
                    if (res) {
                        const pp_b4 rv = *(const pp_b4*)((const __bf16*)a.residual + (size_t)m * a.y_stride + n);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = (__bf16)((float)rv[i] + (float)v[i]);
                    }
                    *(pp_b4*)((__bf16*)a.y + (size_t)m * a.y_stride + n) = v;
                }
            }
        } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = n0 + 64 * wc + 16 * c + 4 * o;
            if (n >= a.N) continue;                                     // (N % 16 == 0: a column block is inside or outside as a whole)
            h4 bv = {0, 0, 0, 0};
            if (bias) bv = *(const h4*)(bias + n);
            h4 rv[8];
            if (side) {
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    int m = m0 + 128 * wr + 16 * b + r;
                    m = m < a.M ? m : a.M - 1;
                    rv[b] = *(const h4*)(side + (size_t)m * a.y_stride + n);
                }
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int m = m0 + 128 * wr + 16 * b + r;
                h4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (_Float16)acc[b][c][i];
                if (bias) v = v + bv;
                if (res) v = rv[b] + v;
                else if (gate) {                                        // act = fp16(silu(gate)) * up   (silu_mul_kernel's expression)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float gf = (float)rv[b][i];
                        v[i] = (_Float16)(gf / (1.0f + __expf(-gf))) * v[i];
                    }
                }
                if (m < a.M) *(h4*)(y + (size_t)m * a.y_stride + n) = v;
            }
        }
        }
        // the epilogue's loads and stores sit on the same counter as the DMA pieces (and stores may return out of order with loads):
        // drain it, so that the loop's vmcnt(8) counts pieces only (the next tile's units in flight were issued >= 2 phases ago).
        // (In front of the exit test, not behind it: every path back to the loop then passes this wait in the control-flow graph
        //  tools/check_waits.py walks -- behind a `break` the structurizer's flag blocks leave a static path around it.)
        AMQ_WAIT_VM("f16pp.drain", 0, "");
        if (nb == tb) break;
        tb = nb; m0 = nm0; n0 = nn0;
    }
#ifdef AMQ_PP_TRACE
    __syncthreads();
    if (blockIdx.x < 16)
        for (int i = threadIdx.x; i < 8 * 128 * 4; i += PP_THREADS)
            amq_pp_trace_buf[(size_t)blockIdx.x * 8 * 128 * 4 + i] = ((const unsigned long long*)(smem + PP_LDS))[i];
#endif
#ifdef AMQ_PP_STAMP
    if (lane == 0 && blockIdx.x < 256) {
        unsigned* o_ = amq_pp_stamp_buf + ((int)blockIdx.x * 8 + wave) * 16;
        for (int p = 0; p < 2; ++p) for (int i = 0; i < 6; ++i) o_[p * 6 + i] = st_acc[p][i];
        o_[12] = (unsigned)T;
    }
#endif
    if (wr == 0) PP_BARRIER();                                          // barrier counts of the two wave groups match again
    AMQ_WAIT_VM("f16pp.exit", 0, "");                                   // the trailing re-reads must not land after the workgroup has gone
}

__global__ __launch_bounds__(PP_THREADS) void gemm_f16_pp_kernel(GemmF16Args a, int ntm, int ntn) { gemm_pp_body<false>(a, ntm, ntn); }
__global__ __launch_bounds__(PP_THREADS) void gemm_bf16_pp_kernel(GemmF16Args a, int ntm, int ntn) { gemm_pp_body<true>(a, ntm, ntn); }


bool gemm_f16w_ok(int M, int N, int K, int x_stride, int y_stride) {
    // DMA sources are buffer base + 32-bit offset: x and W must each span < 4 GiB; 8-byte row-segment stores need 4-element
    // alignment of every output row; 64-deep K tiles in pairs
    const unsigned long long lim = 1ull << 32;
    const unsigned long long xspan = ((unsigned long long)(M - 1) * (unsigned long long)x_stride + (unsigned long long)K) * 2ull;
    const unsigned long long wspan = (unsigned long long)N * (unsigned long long)K * 2ull;
    return M >= 1 && N >= 16 && (N % 16) == 0 && K >= 128 && (K % 128) == 0 && (y_stride & 3) == 0 && (x_stride & 7) == 0 &&
           xspan < lim && wspan < lim;
}

// workgroups of a launch: one per CU of the current device (cached per device ordinal)
static int pp_grid_limit() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cus[dev & 63];
    if (c == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        c = v;
    }
    return c;
}

// the bfloat16 instantiation: x, w, bias, residual, y are bfloat16 (no gate epilogue)
hipError_t launch_gemm_bf16w(const void* x, const void* w, const void* bias, const void* residual, void* y,
                             int M, int N, int K, int x_stride, int y_stride, hipStream_t st) {
    StreamDevice sd_(st);
    static unsigned long long attr_done = 0;
    const hipError_t attr = ensure_dyn_lds(attr_done, (const void*)gemm_bf16_pp_kernel, PP_LDS_ALLOC);
    if (attr != hipSuccess) return attr;
    GemmF16Args a{x, w, bias, residual, nullptr, y, M, N, K, x_stride, y_stride};
    const int ntm = (M + PP_BM - 1) / PP_BM, ntn = (N + PP_BN - 1) / PP_BN;
    const int nt = ntm * ntn, lim = pp_grid_limit();
    hipLaunchKernelGGL(gemm_bf16_pp_kernel, dim3(nt < lim ? nt : lim), dim3(PP_THREADS), PP_LDS_ALLOC, st, a, ntm, ntn);
    return hipGetLastError();
}

hipError_t launch_gemm_f16w(const void* x, const void* w, const void* bias, const void* residual, const void* gate, void* y,
                            int M, int N, int K, int x_stride, int y_stride, hipStream_t st) {
    StreamDevice sd_(st);                                  // attributes / CU counts of the stream's device
    static unsigned long long attr_done = 0;
    const hipError_t attr = ensure_dyn_lds(attr_done, (const void*)gemm_f16_pp_kernel, PP_LDS_ALLOC);
    if (attr != hipSuccess) return attr;
    GemmF16Args a{x, w, bias, residual, gate, y, M, N, K, x_stride, y_stride};
    const int ntm = (M + PP_BM - 1) / PP_BM, ntn = (N + PP_BN - 1) / PP_BN;
    const int nt = ntm * ntn, lim = pp_grid_limit();
#ifdef AMQ_PP_NOT_PERSISTENT       /* A/B build: one workgroup per tile, as the ring kernel launches */
    hipLaunchKernelGGL(gemm_f16_pp_kernel, dim3(nt), dim3(PP_THREADS), PP_LDS_ALLOC, st, a, ntm, ntn);
#else
    hipLaunchKernelGGL(gemm_f16_pp_kernel, dim3(nt < lim ? nt : lim), dim3(PP_THREADS), PP_LDS_ALLOC, st, a, ntm, ntn);
#endif
    return hipGetLastError();
}

}  // namespace amq

#ifdef AMQ_PP_TRACE
extern "C" int amq_debug_pp_trace(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amq::amq_pp_trace_buf), bytes < sizeof(amq::amq_pp_trace_buf) ? bytes : sizeof(amq::amq_pp_trace_buf));
}
#endif
#ifdef AMQ_PP_STAMP
extern "C" int amq_debug_pp_stamps(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amq::amq_pp_stamp_buf), bytes < sizeof(amq::amq_pp_stamp_buf) ? bytes : sizeof(amq::amq_pp_stamp_buf));
}
#endif
