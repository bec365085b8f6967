// amq_attn_prefill.hip -- causal self-attention over a whole prompt (many query rows), gfx950.
//
// Replaces the eager attention of the reference's patched prefill branch (q_len > 8:
// amq/kernel/monkeypatch/ftllama_modeling.py:88-126 -- repeat_kv, matmul(q, k^T) / sqrt(d), causal mask, fp32 softmax,
// fp16 probabilities, matmul with v) with one flash-style MFMA kernel; the previous round ran torch SDPA (AOTriton) here.
//
// Workgroup = 4 waves = 64 query rows of one (sequence, head); wave w owns rows 16w .. 16w+15 and walks the keys in
// tiles of 64, K and V tiles staged once per workgroup in LDS (double-buffered, the next tile's global loads are in
// flight under the current tile's MFMAs).  Everything is computed TRANSPOSED so that no operand ever changes lanes:
//   S^T = K . Q^T      A operand = K rows from LDS (ds_read_b128), B operand = the wave's Q fragments (registers, loaded once)
//                      -> a lane holds S^T[key = 4o + i][q = lane & 15]: its own query row's scores
//   O^T = V^T . P^T    B operand = P^T: exactly the registers the softmax leaves behind (keys 4o..4o+3 of two 16-key
//                      blocks = k-slots 8o + j; both operands use the same key permutation), A operand = V^T through the
//                      hardware transpose read ds_read_b64_tr_b16 of the row-major V tile
//                      -> a lane holds O^T[d = 4o + i][q = lane & 15]: four consecutive output columns of its row (8-byte stores)
// The online softmax is per lane (one query row per lane): the running max needs two cross-lane steps per tile (the four
// 16-lane groups hold different keys of the same rows), the running sum is kept per lane and reduced once at the end.
// HF-Llama numerics as in the decode kernel: fp16 q/k/v, fp32 scores and softmax, probabilities rounded to fp16 before P.V.
#include <type_traits>
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

constexpr int AP_BKV = 64, AP_D = 128;
constexpr int AP_TILE = AP_BKV * AP_D * 2;             // 16 KiB per K or V tile
typedef __fp16 ap_v4h __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f2 __attribute__((ext_vector_type(2)));

// LDS-DMA (global -> LDS, no registers) as inline assembly: with the builtin anywhere in a kernel hipcc (ROCm 7.2) makes every LDS
// read wait lgkmcnt(0) (see amq_gemm_ring.hip).  sbase = wave-uniform 64-bit base, voff = this lane's byte offset, lds_dst =
// wave-uniform LDS byte address; lane l's 16 bytes land at lds_dst + 16 l.  M0 is compiler-reserved: saved and restored.
__device__ __forceinline__ void ap_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// QB = 16-row query blocks per wave (1: 64 query rows per workgroup; 2: 128 -- every K / V fragment read from LDS then feeds two
// MFMAs and a staged tile serves twice the rows)
template <int QB>
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(AttnPrefillArgs a) {
    constexpr int BQ = 64 * QB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2 buffers][K tile | V tile]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r = lane & 15, o = lane >> 4;
    // Workgroup -> (sequence, head, query block).  The dispatcher deals workgroups round-robin over the 8 XCDs, each with its own
    // L2: the query blocks of one (sequence, head) all walk the same K / V, so they are given to ONE XCD (that XCD's share of the
    // launch is a contiguous run of (head, block) ids) -- with blocks dealt across XCDs every L2 sees every head in flight and
    // the tiles come from the memory side instead.  Within a head the heaviest blocks go first (a causal block walks keys
    // 0 .. its last row): the launch does not end on a few long workgroups.
    const int nqb = (a.S + BQ - 1) / BQ;
    const int total = (int)gridDim.x, L = (int)blockIdx.x;
    const int xcd = L & 7, per = total >> 3, rem = total & 7;
    const int vid = xcd * per + (xcd < rem ? xcd : rem) + (L >> 3);      // bijective: XCD x owns per (+1 if x < rem) consecutive ids
    const int hb = vid / nqb;
    const int qb = nqb - 1 - (vid - hb * nqb), b = hb / a.n_heads, h = hb - b * a.n_heads;
    const int kvh = h / (a.n_heads / a.n_kv_heads);
    const int q0 = qb * BQ;
    const int n_keys_seq = a.pos0 + a.S;                // keys of this sequence visible to its last query
    const _Float16* qp = (const _Float16*)a.q + (size_t)b * a.q_bstride + (size_t)h * AP_D;
    const _Float16* kp = (const _Float16*)a.k + (size_t)b * a.k_bstride + (size_t)kvh * a.k_hstride;
    const _Float16* vp = (const _Float16*)a.v + (size_t)b * a.v_bstride + (size_t)kvh * a.v_hstride;

    // the wave's Q fragments (B operand of S^T = K . Q^T): lane (r, o) holds Q[row r][32t + 8o .. +8] of each of its QB row blocks
    int qrow[QB], qpos[QB];
    h8 qf[QB][4];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        qrow[qi] = q0 + 16 * (QB * wave + qi) + r;
        const int qrow_c = qrow[qi] < a.S ? qrow[qi] : a.S - 1;      // rows past S: computed, never stored
#pragma unroll
        for (int t = 0; t < 4; ++t) qf[qi][t] = *(const h8*)(qp + (size_t)qrow_c * a.q_rstride + 32 * t + 8 * o);
        qpos[qi] = a.pos0 + qrow[qi];                   // this lane's query attends keys <= qpos
    }

    // causal: this workgroup needs keys 0 .. pos0 + q0 + BQ - 1
    int last_key = a.pos0 + q0 + BQ - 1;
    if (last_key > n_keys_seq - 1) last_key = n_keys_seq - 1;
    const int n_tiles = last_key / AP_BKV + 1;

    // Staging: LDS-DMA, no registers.  A tile is 16 pieces of 1 KiB = 4 key rows; wave w moves pieces w, w + 4, w + 8, w + 12 of K
    // and of V (8 DMA instructions per tile): lane l -> row 4 piece + (l >> 4), 16-byte position p = l & 15.  The swizzles are
    // applied on the SOURCE side:
    //   K: position p of a row holds chunk p ^ (row & 15)            (conflict-free ds_read_b128 of 16 rows x 4 chunks)
    //   V: position p holds chunk 2 ((p >> 1) ^ (row & 7)) + (p & 1)  (32-byte segments: the transpose read of a 32-lane half
    //      touches 8 rows x 32 bytes: distinct segments -> distinct banks)
    // Lane offsets are loop invariants (the tile start is a multiple of 16 rows, so row & 15 does not depend on the tile) and
    // the tile advances a scalar base: no vector work per tile.  Only a tile that reaches past the sequence's last key (rows
    // there may hold anything, and 0 * NaN is NaN in P.V) takes the clamped per-lane path.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned krs2 = (unsigned)a.k_rstride * 2, vrs2 = (unsigned)a.v_rstride * 2;       // row strides in bytes
    const unsigned rit = 4 * wave + (lane >> 4), pp = lane & 15;                             // row in tile (piece j: + 16 j), position
    const unsigned kswz = (pp ^ (rit & 15)) << 4, vswz = ((((pp >> 1) ^ (rit & 7)) << 1) | (pp & 1)) << 4;
    unsigned koff[4], voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        koff[j] = __umul24(rit + 16 * j, krs2) + kswz;
        voff[j] = __umul24(rit + 16 * j, vrs2) + vswz;
    }
    auto load_tile = [&](int kt, int buf) {
        const unsigned dst = lds0 + buf * (2 * AP_TILE) + wave * 1024;
        if ((kt + 1) * AP_BKV <= n_keys_seq) {
            const unsigned char* kb = (const unsigned char*)kp + (size_t)kt * AP_BKV * krs2;
            const unsigned char* vb = (const unsigned char*)vp + (size_t)kt * AP_BKV * vrs2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ap_glds16(kb, koff[j], dst + j * 4096);
                ap_glds16(vb, voff[j], dst + AP_TILE + j * 4096);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned key = (unsigned)(kt * AP_BKV) + rit + 16 * j;
                key = key < (unsigned)n_keys_seq ? key : (unsigned)n_keys_seq - 1;      // clamp: masked below
                ap_glds16(kp, __umul24(key, krs2) + kswz, dst + j * 4096);
                ap_glds16(vp, __umul24(key, vrs2) + vswz, dst + AP_TILE + j * 4096);
            }
        }
    };
    // this wave's DMA has landed, then every wave's
    auto tile_ready = [&]() {
        AMQ_WAIT_VM("prefill.kv", 0, "");
        __syncthreads();
    };

    f4 oacc[QB][8];
    float m_run[QB], l_run[QB];                         // running max (row-global), running sum (this lane's keys only)
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        m_run[qi] = -INFINITY; l_run[qi] = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) oacc[qi][d] = (f4){0.f, 0.f, 0.f, 0.f};
    }

    load_tile(0, 0);
    tile_ready();
    // The Q fragments must have LANDED before the loop: hipcc's wait-count pass merges the loop header's state with the
    // prologue's, so with the Q loads still counted as outstanding there it waits vmcnt(0) at the first MFMA of EVERY
    // tile -- i.e. for the next tile's global loads issued just above it, the full memory latency once per tile.
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(qf[qi][t]));
    // One key tile.  BUF (the LDS buffer) and DIAG (the tile reaches past the workgroup's first query row: causal mask) are
    // compile-time: every LDS address is then a loop-invariant register + an immediate, and the compare / select pair of the
    // mask exists only in the one or two last tiles of a workgroup.
    const float sl2 = 0.08838834764831845f * 1.4426950408889634f;           // 1 / sqrt(128) * log2(e): scores live in the exp2 domain
    auto tile = [&](auto BUFC, auto DIAGC, int kt) {
        const int buf = BUFC;                           // an integral_constant (addresses fold into immediates) or a plain int
        constexpr bool DIAG = decltype(DIAGC)::value;
        if (kt + 1 < n_tiles) load_tile(kt + 1, buf ^ 1);   // in flight under this tile's MFMAs (the other buffer was last read before the previous barrier)
        const unsigned char* kb_ = smem + buf * (2 * AP_TILE);
        const unsigned char* vb_ = kb_ + AP_TILE;
        const int k0 = kt * AP_BKV;

        // ---- S^T = K . Q^T for the tile's four 16-key blocks (each K fragment feeds the MFMAs of all QB row blocks)
        f4 st[QB][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) st[qi][kb] = (f4){0.f, 0.f, 0.f, 0.f};
            const int row = 16 * kb + r;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const h8 kf = *(const h8*)(kb_ + row * 256 + (((4 * t + o) ^ r) << 4));
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) st[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qi][t], st[qi][kb], 0, 0, 0);
            }
        }
        // ---- causal mask, running max (on the raw scores: the scale is positive), then p = exp2(s * sl2 - m * sl2) as one
        // packed fma + exp per score; the row sum is packed too
        h8 pb[QB][2];                                    // P^T as the B operand of O^T = V^T . P^T, per 32-key step
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            float mt = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (DIAG) st[qi][kb][i] = (k0 + 16 * kb + 4 * o + i) <= qpos[qi] ? st[qi][kb][i] : -INFINITY;
                    mt = fmaxf(mt, st[qi][kb][i]);
                }
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m_run[qi], mt);    // key 0 is visible to every query: finite from the first tile on
            const bool grew = __any(m_new != m_run[qi]); // wave-uniform: no row's maximum moved -> no rescale pass
            const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * sl2);   // exp2(-inf) = 0 on the first tile
            m_run[qi] = m_new;
            const f2 nm = {-m_new * sl2, -m_new * sl2}, sc = {sl2, sl2};
            f2 ls = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const f2 e = __builtin_elementwise_fma((f2){st[qi][kb][i], st[qi][kb][i + 1]}, sc, nm);
                    const f2 p = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                    ls += p;                                                  // fp32 softmax denominator (HF: softmax in fp32, then .to(fp16))
                    pb[qi][kb >> 1][4 * (kb & 1) + i] = (_Float16)p[0];       // normalised at the end
                    pb[qi][kb >> 1][4 * (kb & 1) + i + 1] = (_Float16)p[1];
                }
            l_run[qi] = l_run[qi] * alpha + (ls[0] + ls[1]);
            if (grew) {
#pragma unroll
                for (int d = 0; d < 8; ++d)
#pragma unroll
                    for (int i = 0; i < 4; ++i) oacc[qi][d][i] *= alpha;
            }
        }
        // ---- O^T += V^T . P^T: A operand through the transpose read.  Lane 4q + p of a 16-lane group addresses row q of the
        // group's 4-key block at columns 4p .. 4p+3 and receives column (lane & 15) of the four rows.
        const int tq = r >> 2, tp = r & 3;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row0 = 32 * ks + 4 * o + tq;                       // keys of k-slots j < 4; j >= 4: + 16
                const int seg = 2 * d + (tp >> 1);                           // 16-byte chunk of columns 16 d + 4 tp
                const int off0 = row0 * 256 + ((((seg >> 1) ^ (row0 & 7)) << 5) | ((seg & 1) << 4)) + ((tp & 1) << 3);
                const int row1 = row0 + 16;
                const int off1 = row1 * 256 + ((((seg >> 1) ^ (row1 & 7)) << 5) | ((seg & 1) << 4)) + ((tp & 1) << 3);
                h8 vf;
                // (the _v4f16 form: with the _v4i16 form + per-element bit casts hipcc (ROCm 7.2) built the operand from the first
                //  dword of each result only -- v_perm + v_mov of the low half into the high half -- i.e. keys 4o, 4o+1 twice)
                const ap_v4h t0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) ap_v4h*)(vb_ + off0));
                const ap_v4h t1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) ap_v4h*)(vb_ + off1));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vf[i] = (_Float16)t0[i];
                    vf[4 + i] = (_Float16)t1[i];
                }
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) oacc[qi][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[qi][ks], oacc[qi][d], 0, 0, 0);
            }
        }
        tile_ready();
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // tiles 0 .. n_plain-1 end at or before the workgroup's first query position: no mask for any row
    int n_plain = (a.pos0 + q0 + 1) / AP_BKV;
    n_plain = n_plain < n_tiles ? n_plain : n_tiles;
    int kt = 0;
    if (QB == 1) {
        for (; kt + 2 <= n_plain; kt += 2) {
            tile(B0{}, std::false_type{}, kt);
            tile(B1{}, std::false_type{}, kt + 1);
        }
        for (; kt < n_tiles; ++kt) {                     // at most one unmasked tile, then the masked ones (one or two)
            if (kt < n_plain) tile(B0{}, std::false_type{}, kt);             // kt is even here
            else if (kt & 1) tile(B1{}, std::true_type{}, kt);
            else tile(B0{}, std::true_type{}, kt);
        }
    } else {                                             // two tile bodies in one loop do not fit the register file at QB = 2
        for (; kt < n_plain; ++kt) tile(kt & 1, std::false_type{}, kt);
        for (; kt < n_tiles; ++kt) tile(kt & 1, std::true_type{}, kt);
    }

    // ---- normalise and store: lane holds O[q = qrow][d = 16 db + 4o + i]
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        float l = l_run[qi];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        // (the fp16 values are formed once, ahead of the two store layouts: formed inside each branch the compiler picked a
        //  fused multiply-convert for one and multiply + convert for the other -- results one fp16 ulp apart in ~1e-4 of the elements)
        h4 ov[8];
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) ov[d][i] = (_Float16)(oacc[qi][d][i] * inv);
        if (a.out_xfrag) {
            // out in fragment order (amq_xfrag_f16's layout of the [S, heads * 128] matrix, K tile = head): element (s, h, dd) at
            // ((((s >> 6) * heads + h) * 16 + ((s & 63) >> 4) * 4 + dd / 32) * 64 + 16 * ((dd & 31) >> 3) + (s & 15)) * 8 + (dd & 7);
            // rows S .. 64 ceil(S / 64) - 1 of the last group are written as zeros (the few-row GEMM reads whole 64-row groups)
            const int s = qrow[qi];
            if (s < ((a.S + 63) & ~63)) {
                _Float16* const xf = (_Float16*)a.out;
                const size_t grp = ((size_t)(s >> 6) * a.n_heads + h) * 16 + ((s & 63) >> 4) * 4;
                const h4 zero = {0, 0, 0, 0};
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const int ox = 2 * (d & 1) + (o >> 1);
                    *(h4*)(xf + ((grp + (d >> 1)) * 64 + 16 * ox + (s & 15)) * 8 + 4 * (o & 1)) = s < a.S ? ov[d] : zero;
                }
            }
        } else if (qrow[qi] < a.S) {
            _Float16* op = (_Float16*)a.out + (size_t)b * a.o_bstride + (size_t)qrow[qi] * a.o_rstride + (size_t)h * AP_D + 4 * o;
#pragma unroll
            for (int d = 0; d < 8; ++d) *(h4*)(op + 16 * d) = ov[d];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Decode step of a GROUPED-QUERY model over a long cache: one workgroup per (kv head, chunk of 64 NT keys) serves all G = n_heads / n_kv_heads
// query heads of the group -- the group's rotated queries are the (up to 16) rows of ONE MFMA row block, so the chunk's K / V rows are taken in
// once and scored against every head by the matrix cores.  attn_decode_split_kernel (amq_decode.hip) gives every QUERY head its own workgroups of
// 16-lane dot products: the heads of a group each take in the same chunk (Llama-3.1-8B, 32 / 8 heads, 8192 cached keys: 1024 workgroups, 23.6 us
// for 33.6 MB of K + V; a first grouped form of THAT kernel -- rows loaded once, scored against the G queries by the same VALU code, bit-identical
// -- moved the work, not the time: 22.9 us, and 18.2 against 10.4 at 2048 keys, profiles/r06_attn_gqa.txt: the launch is bound by its vector
// arithmetic, not by its bytes).
//   * the chunk's NT tiles of 64 keys x (K | V) are requested at once by LDS-DMA (source-side swizzles of the prompt kernel above), rows at or past
//     the new token's position clamped to the last cached row; the raw q / k / v of the step and the position's cos / sin pairs are requested
//     behind them, ONE vmcnt(0) covers everything;
//   * every lane rotates its query fragments in registers (HF's fp16 expression, the decode kernels'), the first 64 threads the new key -- appended
//     to the cache by the workgroup whose chunk ends at the new token, which also writes it (and the new value) over the LDS rows at or past the
//     position, so that no row the MFMAs read was never written;
//   * wave w takes tile w: S^T = K . Q^T, key mask, per-row maximum and exp2, O^T = V^T . P^T (the prompt kernel's operand layouts); the waves'
//     (m, l, O) are merged in wave order through LDS;
//   * one active chunk: normalised output.  Several: (O, m, l) per query head into the split kernel's workspace; attn_gqa_combine_kernel, the next
//     launch, adds them in chunk order.
// Numerics: the prompt kernel's -- fp32 scores, exp2 domain, un-normalised probabilities rounded to fp16 before P.V -- within the tests' bound of
// the eager fp32 formula and of the per-head kernels (tests/test_gpu_decode.py::test_attn_decode_gqa_*); deterministic (fixed merge orders).
struct AttnGqaArgs {
    const void* q; const void* k; const void* v; void* kc; void* vc; void* out;
    const void* state;          // cur mode: the step-state block (cos/sin row, position at byte 256, error word at 260); else a device int32 position or null
    const void* rope_table;     // [max_seq][64] (cos, sin) pairs, or null (cur mode / computed from rope_theta)
    float* ws;
    int pos, n_heads, n_kv_heads, max_seq, n_splits, cur_mode;
    float rope_theta;
    int iters;                  // stages of AG_STAGE_KEYS keys per workgroup (chunk = AG_STAGE_KEYS * iters)
};
constexpr int AG_WS_STRIDE = 132;       // floats per (query head, chunk) in the workspace: O[128], m, l, pad (amq_decode.hip: ATT_WS_STRIDE)
constexpr int AG_OM_STRIDE = 132;       // floats per (wave, query row) of the cross-wave merge area
constexpr int AG_STAGE = 512;           // bytes in front of the stage buffers: rotated new key [128], new value [128]
constexpr int AG_STAGE_KEYS = 128;      // keys per stage: two tiles of 64

__global__ __launch_bounds__(256) void attn_decode_gqa_kernel(AttnGqaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r = lane & 15, o = lane >> 4;
    const int kvh = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
    const int G = a.n_heads / a.n_kv_heads, h0 = kvh * G;
    int pos;
    if (a.cur_mode) pos = *(const int*)((const char*)a.state + 256);
    else if (a.state) pos = *(const int*)a.state;
    else pos = a.pos;
    if (pos < 0 || pos >= a.max_seq) {              // a position outside the cache: nothing is appended or written (attn_decode_kernel's guard)
        if (a.cur_mode && tid == 0 && z == 0 && kvh == 0) *(int*)((char*)const_cast<void*>(a.state) + 260) = 1;
        return;
    }
    const int chunk = AG_STAGE_KEYS * a.iters;
    const int T = pos + 1;
    const int n_act = (T + chunk - 1) / chunk;      // chunks that hold keys: workgroups z >= n_act have nothing to do
    if (z >= n_act) return;
    const int t0 = z * chunk;
    const int t1 = t0 + chunk < T ? t0 + chunk : T; // this workgroup's keys: t0 .. t1 - 1
    const int n_it = (t1 - t0 + AG_STAGE_KEYS - 1) / AG_STAGE_KEYS;     // stages of 128 keys that hold any
    const bool has_new = t1 == T;                   // the chunk that ends at the new token
    const unsigned last_old = pos > 0 ? pos - 1 : 0;    // rows >= pos are never read from the cache
    _Float16* const kc = (_Float16*)a.kc + ((size_t)b * a.n_kv_heads + kvh) * (size_t)a.max_seq * AP_D;
    _Float16* const vc = (_Float16*)a.vc + ((size_t)b * a.n_kv_heads + kvh) * (size_t)a.max_seq * AP_D;

    // ---- a stage = two tiles of 64 keys x (K | V) by LDS-DMA into one of two stage buffers (the prompt kernel's staging: 16 pieces of 1 KiB per tile
    // and operand, swizzles on the source side); rows at or past the new token's position clamped to the last cached row
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + AG_STAGE;
    const unsigned rit = 4 * wave + (lane >> 4), pp = lane & 15;
    const unsigned kswz = (pp ^ (rit & 15)) << 4, vswz = ((((pp >> 1) ^ (rit & 7)) << 1) | (pp & 1)) << 4;
    auto load_stage = [&](int it) {
        const unsigned dst0 = lds0 + (it & 1) * (4 * AP_TILE) + wave * 1024;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned key = (unsigned)(t0 + AG_STAGE_KEYS * it + 64 * ti) + rit + 16 * j;
                key = key < last_old ? key : last_old;
                ap_glds16(kc, __umul24(key, 256u) + kswz, dst0 + ti * (2 * AP_TILE) + j * 4096);
                ap_glds16(vc, __umul24(key, 256u) + vswz, dst0 + ti * (2 * AP_TILE) + AP_TILE + j * 4096);
            }
        }
    };
    load_stage(0);
    // ---- behind it: this lane's raw query fragments (row r = query head h0 + r; rows past the group repeat its last head: computed, never stored),
    // the cos / sin pairs they rotate with, and -- the first 64 threads -- the new key / value
    const int rq = r < G ? r : G - 1;
    const _Float16* const qb = (const _Float16*)a.q + ((size_t)b * a.n_heads + h0 + rq) * AP_D + 8 * o;
    h8 qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = *(const h8*)(qb + 32 * t);
    const h2* const cs_src = a.cur_mode ? (const h2*)a.state : (a.rope_table ? (const h2*)a.rope_table + (size_t)pos * 64 : nullptr);
    h8 cs[2][2];                                    // [t][half]: (cos, sin) of pairs 32 t + 8 o .. + 3 and .. + 4 .. + 7
    _Float16 k0 = 0, k1 = 0, v0 = 0, v1 = 0;
    h2 csk = {(_Float16)1.f, (_Float16)0.f};
    if (cs_src) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            cs[t][0] = *(const h8*)(cs_src + 32 * t + 8 * o);
            cs[t][1] = *(const h8*)(cs_src + 32 * t + 8 * o + 4);
        }
        if (tid < 64) csk = cs_src[tid];
    } else {                                        // no table, no step state: the pairs from rope_theta (attn_decode_kernel's rope_cs)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = 32 * t + 8 * o + e;
                float sn, c_;
                sincosf((float)pos * (1.0f / powf(a.rope_theta, (float)(2 * i) / (float)AP_D)), &sn, &c_);
                cs[t][e >> 2][2 * (e & 3)] = (_Float16)c_;
                cs[t][e >> 2][2 * (e & 3) + 1] = (_Float16)sn;
            }
        if (tid < 64) {
            float sn, c_;
            sincosf((float)pos * (1.0f / powf(a.rope_theta, (float)(2 * tid) / (float)AP_D)), &sn, &c_);
            csk = (h2){(_Float16)c_, (_Float16)sn};
        }
    }
    if (tid < 64) {
        const _Float16* const kn = (const _Float16*)a.k + ((size_t)b * a.n_kv_heads + kvh) * AP_D;
        const _Float16* const vn = (const _Float16*)a.v + ((size_t)b * a.n_kv_heads + kvh) * AP_D;
        k0 = kn[tid]; k1 = kn[tid + 64];
        v0 = vn[tid]; v1 = vn[tid + 64];
    }
    AMQ_WAIT_VM("attn.gqa.landed", 0, "");          // this wave's share of stage 0 and its loads
    // rotation: q' = q * cos + rotate_half(q) * sin in fp16 (HF apply_rotary_pos_emb; the decode kernels' expression): pair (d, d + 64) = fragments t, t + 2
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const h8 lo = qf[t], hi = qf[t + 2];
        h8 nlo, nhi;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 c16 = cs[t][e >> 2][2 * (e & 3)], s16 = cs[t][e >> 2][2 * (e & 3) + 1];
            nlo[e] = lo[e] * c16 + (-hi[e]) * s16;
            nhi[e] = hi[e] * c16 + lo[e] * s16;
        }
        qf[t] = nlo; qf[t + 2] = nhi;
    }
    _Float16* const ks = (_Float16*)smem;           // [128] rotated new key | [128] new value
    _Float16* const vs = ks + AP_D;
    if (tid < 64) {
        const _Float16 r0 = k0 * csk.x + (-k1) * csk.y, r1 = k1 * csk.x + k0 * csk.y;
        ks[tid] = r0; ks[tid + 64] = r1;
        vs[tid] = v0; vs[tid + 64] = v1;
        if (has_new) {                              // the one workgroup of this kv head that appends
            kc[(size_t)pos * AP_D + tid] = r0;
            kc[(size_t)pos * AP_D + tid + 64] = r1;
            vc[(size_t)pos * AP_D + tid] = v0;
            vc[(size_t)pos * AP_D + tid + 64] = v1;
        }
    }
    __syncthreads();                                // every wave's share of stage 0 has landed; ks / vs are visible
    unsigned char* const tiles = smem + AG_STAGE;

    // ---- the stages: wave w takes keys 32 (w & 1) .. + 31 of tile w >> 1 (the prompt kernel's tile body cut to one 32-key step, every row at the same
    // position), its running (m, l, O) carried across the stages; stage it + 1 is in flight under stage it's MFMAs
    const float sl2 = 0.08838834764831845f * 1.4426950408889634f;           // 1 / sqrt(128) * log2(e)
    const int tile = wave >> 1, half = wave & 1;
    f4 oacc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) oacc[d] = (f4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    for (int it = 0; it < n_it; ++it) {
        if (it + 1 < n_it) load_stage(it + 1);      // (its buffer was last read in stage it - 1, behind that stage's barrier)
        unsigned char* const sb = tiles + (it & 1) * (4 * AP_TILE);
        if (has_new && it == n_it - 1) {
            // rows at or past the position (the new token's own row and the clamped repeats behind it in its tile) take the new key / value: 16 chunks
            // of 16 bytes per row and operand, placed where the staging's swizzles would have put them -- no row the MFMAs read was never written
            const int lr0 = pos - (t0 + AG_STAGE_KEYS * it);    // first such row, stage-local
            const int lr1 = ((lr0 >> 6) + 1) << 6;              // end of its tile (a later tile of the stage holds no keys: not computed)
            for (int idx = tid; idx < (lr1 - lr0) * 16; idx += 256) {
                const int lr = lr0 + (idx >> 4), c = idx & 15;
                unsigned char* const tb = sb + (lr >> 6) * (2 * AP_TILE);
                const int row = lr & 63;
                *(h8*)(tb + row * 256 + ((c ^ (row & 15)) << 4)) = *(const h8*)(ks + 8 * c);
                *(h8*)(tb + AP_TILE + row * 256 + (((((c >> 1) ^ (row & 7)) << 1) | (c & 1)) << 4)) = *(const h8*)(vs + 8 * c);
            }
            __syncthreads();
        }
        const int k0_ = t0 + AG_STAGE_KEYS * it + 64 * tile + 32 * half;      // this wave's first key of the stage
        if (k0_ < t1) {
            const unsigned char* const kb_ = sb + tile * (2 * AP_TILE);
            const unsigned char* const vb_ = kb_ + AP_TILE;
            f4 st[2];
#pragma unroll
            for (int kl = 0; kl < 2; ++kl) {
                st[kl] = (f4){0.f, 0.f, 0.f, 0.f};
                const int row = 32 * half + 16 * kl + r;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const h8 kf = *(const h8*)(kb_ + row * 256 + (((4 * t + o) ^ r) << 4));
                    st[kl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[t], st[kl], 0, 0, 0);
                }
            }
            float mt = -INFINITY;
#pragma unroll
            for (int kl = 0; kl < 2; ++kl)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    st[kl][i] = (k0_ + 16 * kl + 4 * o + i) < t1 ? st[kl][i] : -INFINITY;
                    mt = fmaxf(mt, st[kl][i]);
                }
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));     // the wave's first key is inside the chunk: finite
            const float m_new = fmaxf(m_run, mt);
            const bool grew = __any(m_new != m_run);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);      // exp2(-inf) = 0 on the wave's first stage
            m_run = m_new;
            h8 pb;
            float ls = 0.f;
#pragma unroll
            for (int kl = 0; kl < 2; ++kl)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __builtin_amdgcn_exp2f(st[kl][i] * sl2 - m_new * sl2);
                    ls += p;
                    pb[4 * kl + i] = (_Float16)p;
                }
            l_run = l_run * alpha + ls;
            if (grew) {
#pragma unroll
                for (int d = 0; d < 8; ++d)
#pragma unroll
                    for (int i = 0; i < 4; ++i) oacc[d][i] *= alpha;
            }
            const int tq = r >> 2, tp = r & 3;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int row0 = 32 * half + 4 * o + tq;
                const int seg = 2 * d + (tp >> 1);
                const int off0 = row0 * 256 + ((((seg >> 1) ^ (row0 & 7)) << 5) | ((seg & 1) << 4)) + ((tp & 1) << 3);
                const int row1 = row0 + 16;
                const int off1 = row1 * 256 + ((((seg >> 1) ^ (row1 & 7)) << 5) | ((seg & 1) << 4)) + ((tp & 1) << 3);
                h8 vf;
                const ap_v4h t0_ = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) ap_v4h*)(vb_ + off0));
                const ap_v4h t1_ = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) ap_v4h*)(vb_ + off1));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vf[i] = (_Float16)t0_[i];
                    vf[4 + i] = (_Float16)t1_[i];
                }
                oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb, oacc[d], 0, 0, 0);
            }
        }
        AMQ_WAIT_VM("attn.gqa.stage", 0, "");       // this wave's share of the next stage
        __syncthreads();                            // ... every wave's; and this stage's buffer has been read
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    // ---- merge the waves' (m, l, O) in wave order (the stage buffers are free: the loop ends on a barrier): lane holds O^T[d = 16 db + 4 o + i][query r]
    float* const Om = (float*)tiles;                // [4][16][AG_OM_STRIDE]
    float* const mlw = Om + 4 * 16 * AG_OM_STRIDE;  // [4][16][2]
#pragma unroll
    for (int d = 0; d < 8; ++d) *(f4*)(Om + ((size_t)wave * 16 + r) * AG_OM_STRIDE + 16 * d + 4 * o) = oacc[d];
    if (o == 0) { mlw[(wave * 16 + r) * 2] = m_run; mlw[(wave * 16 + r) * 2 + 1] = l_run; }
    __syncthreads();
    const bool single = n_act == 1;                 // wave-uniform
    for (int idx = tid; idx < G * AP_D; idx += 256) {
        const int qh = idx >> 7, d = idx & 127;
        float M = mlw[qh * 2];
#pragma unroll
        for (int w = 1; w < 4; ++w) M = fmaxf(M, mlw[(w * 16 + qh) * 2]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __builtin_amdgcn_exp2f(mlw[(w * 16 + qh) * 2] * sl2 - M * sl2);       // a wave that never had keys: exp2(-inf) = 0
            L += mlw[(w * 16 + qh) * 2 + 1] * f;
            O += Om[((size_t)w * 16 + qh) * AG_OM_STRIDE + d] * f;
        }
        if (single) {
            ((_Float16*)a.out)[((size_t)b * a.n_heads + h0 + qh) * AP_D + d] = (_Float16)(O / L);
        } else {
            float* const wz = a.ws + (((size_t)b * a.n_heads + h0 + qh) * (size_t)a.n_splits + z) * AG_WS_STRIDE;
            wz[d] = O;                              // (read by attn_gqa_combine_kernel, the next launch)
            if (d == 0) {
                wz[AP_D] = M * 0.08838834764831845f;    // natural-exponent domain, as the split kernel's
                wz[AP_D + 1] = L;
            }
        }
    }
}

// The combine of a grouped-query decode step as a launch of its own: one 128-thread workgroup per (sequence, QUERY head) adds the chunks' (O, m, l)
// in chunk order -- out = sum_c O_c e^(m_c - M) / sum_c l_c e^(m_c - M), attn_decode_split_kernel's expressions.  (As the last arriver's job inside
// attn_decode_gqa_kernel it was ONE workgroup per kv head reading G x n_act x 528 bytes through dependent round trips: 25 of the launch's 40 us at
// 8192 keys of a 32 / 8-head model, growing with G x n_act -- profiles/r06_attn_gqa.txt.  The kernel boundary is the hand-over: no tickets, no
// agent-scope publish.)  One active chunk: attn_decode_gqa_kernel has written the output itself, nothing to do.
__global__ __launch_bounds__(128) void attn_gqa_combine_kernel(const float* ws, void* out, const void* state, int pos_host, int cur_mode,
                                                               int n_heads, int max_seq, int n_splits, int chunk) {
    __shared__ float ml[2 * 1024];
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* const wsh = ws + ((size_t)b * n_heads + h) * (size_t)n_splits * AG_WS_STRIDE;
    // Everything the first 32 chunks hold is requested before the position (hence the number of active chunks) is known: the workspace has n_splits
    // slots per head whatever the position, a slot past the active chunks holds an older step's values and is not used.  One round trip for the
    // position, the (m, l) pairs and 32 O rows instead of four dependent ones (5.9 -> ~3.5 us per launch at 32 chunks).
    constexpr int SPEC = 32;
    float ov[SPEC];
#pragma unroll
    for (int j = 0; j < SPEC; ++j) ov[j] = wsh[(size_t)(j < n_splits ? j : n_splits - 1) * AG_WS_STRIDE + d];
    float m_d = 0.f, l_d = 0.f;
    if (d < n_splits) { m_d = wsh[(size_t)d * AG_WS_STRIDE + AP_D]; l_d = wsh[(size_t)d * AG_WS_STRIDE + AP_D + 1]; }
    int pos;
    if (cur_mode) pos = *(const int*)((const char*)state + 256);
    else if (state) pos = *(const int*)state;
    else pos = pos_host;
    if (pos < 0 || pos >= max_seq) return;
    const int n_act = (pos + chunk) / chunk;        // ceil((pos + 1) / chunk)
    if (n_act <= 1) return;
    if (d < n_splits) { ml[2 * d] = m_d; ml[2 * d + 1] = l_d; }
    for (int c = d + 128; c < n_act; c += 128) {
        ml[2 * c] = wsh[(size_t)c * AG_WS_STRIDE + AP_D];
        ml[2 * c + 1] = wsh[(size_t)c * AG_WS_STRIDE + AP_D + 1];
    }
    __syncthreads();
    float M = -INFINITY;
    for (int c = 0; c < n_act; ++c) M = fmaxf(M, ml[2 * c]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int j = 0; j < SPEC; ++j) {
        if (j < n_act) {                            // chunk order
            const float f = __expf(ml[2 * j] - M);
            L += ml[2 * j + 1] * f;
            O += ov[j] * f;
        }
    }
    for (int c0 = SPEC; c0 < n_act; c0 += 16) {
        float o16[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {              // sixteen chunks' O rows per round trip
            const int cc = c0 + j < n_act ? c0 + j : n_act - 1;
            o16[j] = wsh[(size_t)cc * AG_WS_STRIDE + d];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (c0 + j < n_act) {
                const float f = __expf(ml[2 * (c0 + j)] - M);
                L += ml[2 * (c0 + j) + 1] * f;
                O += o16[j] * f;
            }
        }
    }
    ((_Float16*)out)[((size_t)b * n_heads + h) * AP_D + d] = (_Float16)(O / L);
}

// stages of 128 keys per workgroup for a cache of max_seq rows cut n_splits ways
int attn_decode_gqa_iters(int max_seq, int n_splits) {
    const int per = (max_seq + n_splits - 1) / n_splits;
    return (per + AG_STAGE_KEYS - 1) / AG_STAGE_KEYS;
}
bool attn_decode_takes_gqa(int n_heads, int n_kv_heads, int max_seq, int n_splits) {
    const int G = n_kv_heads > 0 ? n_heads / n_kv_heads : 0;
    (void)max_seq;
    return AMQ_ATT_GQA && G >= 2 && G <= 16 && n_splits > 1;
}

hipError_t launch_attn_decode_gqa(const AttnArgs& a, int batch, int n_splits, void* ws, void* tickets, hipStream_t st) {
    (void)tickets;                                         // (the grouped form hands over at a kernel boundary)
    StreamDevice sd_(st);
    const bool cur = a.rope_cur != nullptr;
    AttnGqaArgs g{a.q, a.k, a.v, a.kcache, a.vcache, a.out, cur ? a.rope_cur : (const void*)a.pos_dev, cur ? nullptr : a.rope_table,
                  (float*)ws, a.pos, a.n_heads, a.n_kv_heads, a.max_seq, n_splits, (int)cur, a.rope_theta, attn_decode_gqa_iters(a.max_seq, n_splits)};
    const size_t lds = AG_STAGE + (size_t)2 * 4 * AP_TILE;         // two stage buffers of two tiles x (K | V); the cross-wave merge reuses them
    static_assert((size_t)(4 * 16 * AG_OM_STRIDE + 4 * 16 * 2) * sizeof(float) <= (size_t)2 * 4 * AP_TILE, "merge area inside the stage buffers");
    static unsigned long long attr_done = 0;
    if (hipError_t e = ensure_dyn_lds(attr_done, (const void*)attn_decode_gqa_kernel, (int)lds)) return e;
    hipLaunchKernelGGL(attn_decode_gqa_kernel, dim3(g.n_kv_heads, batch, g.n_splits), dim3(256), lds, st, g);
    if (hipError_t e = hipGetLastError()) return e;
#ifdef AMQ_GQA_ABL_NO_COMBINE          /* timing-only ablation: what the combine launch adds to the step (results wrong beyond one chunk) */
    return hipSuccess;
#endif
    hipLaunchKernelGGL(attn_gqa_combine_kernel, dim3(g.n_heads, batch), dim3(128), 0, st, (const float*)g.ws, g.out, g.state, g.pos, g.cur_mode,
                       g.n_heads, g.max_seq, g.n_splits, AG_STAGE_KEYS * g.iters);
    return hipGetLastError();
}

hipError_t launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t st) {
    StreamDevice sd_(st);                                  // attributes / CU counts of the stream's device
    const int lds = 2 * 2 * AP_TILE;                    // 64 KiB
    static unsigned long long attr1_done = 0, attr2_done = 0;
    const hipError_t attr1 = ensure_dyn_lds(attr1_done, (const void*)attn_prefill_kernel<1>, lds);
    if (attr1 != hipSuccess) return attr1;
    const hipError_t attr2 = ensure_dyn_lds(attr2_done, (const void*)attn_prefill_kernel<2>, lds);
    if (attr2 != hipSuccess) return attr2;
    // 128-row workgroups (QB = 2: every K / V fragment read from LDS feeds two MFMAs, a staged tile serves twice the rows) once
    // they still fill the chip several times over (2 workgroups per CU = 512 in flight); 64-row workgroups otherwise.
    // Measured (profiles/r02_attn_prefill_vs_sdpa.txt): 16 x 2048 x 40 heads 681 vs 577 TFLOP/s, 1 x 2048 x 32 heads 419 vs 427,
    // 1 x 512 x 32 heads 106 vs 138.
#ifdef AP_QB
    const int qb = AP_QB;                               // A/B builds
#else
    const long wg2 = (long)((a.S + 127) / 128) * a.n_heads * a.batch;
    const int qb = wg2 >= 2048 ? 2 : 1;
#endif
    const int nwg = ((a.S + 64 * qb - 1) / (64 * qb)) * a.n_heads * a.batch;
    if (qb == 2) hipLaunchKernelGGL(attn_prefill_kernel<2>, dim3(nwg), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(attn_prefill_kernel<1>, dim3(nwg), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace amq
