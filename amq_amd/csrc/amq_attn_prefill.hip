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
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

constexpr int AP_BKV = 64, AP_D = 128;
constexpr int AP_TILE = AP_BKV * AP_D * 2;             // 16 KiB per K or V tile
typedef __fp16 ap_v4h __attribute__((__vector_size__(4 * sizeof(__fp16))));

// QB = 16-row query blocks per wave (1: 64 query rows per workgroup; 2: 128 -- every K / V fragment read from LDS then feeds two
// MFMAs and a staged tile serves twice the rows: at 1 the kernel is LDS-bound, 1 KB read per MFMA)
template <int QB>
__global__ __launch_bounds__(256) void attn_prefill_kernel(AttnPrefillArgs a) {
    constexpr int BQ = 64 * QB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2 buffers][K tile | V tile]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r = lane & 15, o = lane >> 4;
    // heaviest query blocks first (a causal block walks keys 0 .. its last row): the launch does not end on a few long workgroups
    const int qb = (int)gridDim.x - 1 - (int)blockIdx.x, h = (int)blockIdx.y, b = (int)blockIdx.z;
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

    // staging: 1024 16-byte chunks per tile, 4 per thread; chunk id c = tid + 256 j -> key row c >> 4, 16-byte chunk c & 15
    h8 kreg[4], vreg[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (int)threadIdx.x + 256 * j;
            int key = kt * AP_BKV + (c >> 4);
            key = key < n_keys_seq ? key : n_keys_seq - 1;      // clamp: masked below
            kreg[j] = *(const h8*)(kp + (size_t)key * a.k_rstride + (c & 15) * 8);
            vreg[j] = *(const h8*)(vp + (size_t)key * a.v_rstride + (c & 15) * 8);
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* kb_ = smem + buf * (2 * AP_TILE);
        unsigned char* vb_ = kb_ + AP_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (int)threadIdx.x + 256 * j;
            const int row = c >> 4, ch = c & 15;
            // K: 16-byte chunk ch of row at position ch ^ (row & 15)        (conflict-free ds_read_b128 of 16 rows x 4 chunks)
            *(h8*)(kb_ + row * 256 + ((ch ^ (row & 15)) << 4)) = kreg[j];
            // V: 32-byte segment (ch >> 1) of row at position (ch >> 1) ^ (row & 7)  (the transpose read of a 32-lane half
            //    touches 8 rows x 32 bytes: distinct segments -> distinct banks)
            *(h8*)(vb_ + row * 256 + ((((ch >> 1) ^ (row & 7)) << 5) | ((ch & 1) << 4))) = vreg[j];
        }
    };

    f4 oacc[QB][8];
    float m_run[QB], l_run[QB];                         // running max (row-global), running sum (this lane's keys only)
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        m_run[qi] = -INFINITY; l_run[qi] = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) oacc[qi][d] = (f4){0.f, 0.f, 0.f, 0.f};
    }

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < n_tiles) load_tile(kt + 1);        // in flight under this tile's MFMAs
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
        // ---- scale, causal mask, running max.  Scores are kept in the exp2 domain (scale * log2(e) folded into one multiply).
        // Only tiles that reach past a row block's first query row need the mask (wave-uniform test): for a 2048-row prompt
        // that is one tile in 16 on average.
        const float sl2 = 0.08838834764831845f * 1.4426950408889634f;       // 1 / sqrt(128) * log2(e)
        h8 pb[QB][2];                                    // P^T as the B operand of O^T = V^T . P^T, per 32-key step
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            const bool diag = k0 + AP_BKV - 1 > a.pos0 + q0 + 16 * (QB * wave + qi);   // some key may be masked for some row of the block
            float mt = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float s = st[qi][kb][i] * sl2;
                    if (diag) s = (k0 + 16 * kb + 4 * o + i) <= qpos[qi] ? s : -INFINITY;
                    st[qi][kb][i] = s;
                    mt = fmaxf(mt, s);
                }
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m_run[qi], mt);    // key 0 is visible to every query: finite from the first tile on
            const bool grew = __any(m_new != m_run[qi]); // wave-uniform: no row's maximum moved -> no rescale pass
            const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new);   // exp2(-inf) = 0 on the first tile
            m_run[qi] = m_new;
            float ls = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16 p16 = (_Float16)__builtin_amdgcn_exp2f(st[qi][kb][i] - m_new);   // softmax(...).to(fp16), normalised at the end
                    ls += (float)p16;
                    pb[qi][kb >> 1][4 * (kb & 1) + i] = p16;
                }
            l_run[qi] = l_run[qi] * alpha + ls;
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
        if (kt + 1 < n_tiles) store_tile(buf ^ 1);       // the other buffer was last read before the previous barrier
        __syncthreads();
    }

    // ---- normalise and store: lane holds O[q = qrow][d = 16 db + 4o + i]
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        float l = l_run[qi];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        if (qrow[qi] < a.S) {
            _Float16* op = (_Float16*)a.out + (size_t)b * a.o_bstride + (size_t)qrow[qi] * a.o_rstride + (size_t)h * AP_D + 4 * o;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                h4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (_Float16)(oacc[qi][d][i] * inv);
                *(h4*)(op + 16 * d) = v;
            }
        }
    }
}

hipError_t launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t st) {
    const int lds = 2 * 2 * AP_TILE;                    // 64 KiB
    static hipError_t attr1 = hipFuncSetAttribute((const void*)attn_prefill_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr1 != hipSuccess) return attr1;
    // QB = 2 (128-row workgroups, every K / V fragment feeding two MFMAs) was built and measured: 244 registers, and SLOWER --
    // 16 x 2048 x 40 heads 194 vs 261 TFLOP/s, 1 x 2048 x 32 heads 194 vs 226 (profiles/r02_attn_prefill_vs_sdpa.txt) -- the
    // kernel is bound by the per-tile dependency chain (barrier, S^T, two cross-lane max steps, exp, P.V), not by LDS bytes.
    hipLaunchKernelGGL(attn_prefill_kernel<1>, dim3((a.S + 63) / 64, a.n_heads, a.batch), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace amq
