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
