// amq_engine.hip -- one decode token (all decoder blocks) as ONE persistent launch, gfx950.
//
// Reference counterpart: the per-token loop of the patched Llama forward,
//   amq/kernel/monkeypatch/ftllama_modeling.py:167-230 (decoder layer: norm, q/k/v, cached attention, o_proj, MLP)
// whose seven linears per block are the 2/3/4-bit modules of amq_speed_benchmark.py:231-256.
//
// Why: the five-launch-per-block step (amq_gemv.hip + amq_decode.hip under a hipGraph) spends ~3.7-4.4 us of FIXED cost per
// dependent launch (boundary, prologue, first weight data, tail) -- two thirds of a 7B token.  Here the stages
// of a block are phases of one kernel separated by a device-wide barrier, and every wave keeps its share of the weight
// stream running ACROSS those barriers: weights do not depend on activations, so the first tiles of stage s+1 are already
// in registers when the barrier of stage s opens.
//
// Structure
//   * grid = one 16-wave workgroup per CU (128 VGPRs: the whole register file of the CU, so the dispatcher cannot put two
//     on one CU and every workgroup is resident -- the barrier needs that; every spin is bounded all the same).
//   * a GEMV stage (q/k/v, o_proj, gate/up, down_proj) = the 16-wave form of amq::gemv_kernel: the stage's row-tiles (16
//     output rows each, all segments concatenated) are cut into gridDim.x contiguous ranges; wave w of a workgroup takes
//     the k-tiles w, w + 16, ... of each of its row-tiles and accumulates them with the SAME instruction sequence
//     (dequant_lane_sd + 4 x v_mfma_f32_16x16x32_f16 per tile, x rows from LDS); the 16 partial sums of a row-tile are
//     added in wave order.  Results are therefore bit-identical to amq_gemv_grouped_f16 with amq_gemv_opts.waves = 16.
//     Partials of ALL the workgroup's row-tiles go to LDS and are summed behind ONE workgroup barrier per stage.
//   * the weight ring: U tiles per wave in flight in a wave-PRIVATE ring of LDS slots, filled by LDS-DMA
//     (global_load_lds_dwordx4 for the packed tile as it lies in HBM: 64 / 48 / 32 lanes x 16 bytes for 4 / 3 / 2 bit;
//     global_load_lds_dword for its 16 (scale, zero) pairs) issued from inline asm with hand-counted s_waitcnt vmcnt -- two
//     operations per tile, so the count is a constant; no barrier guards a slot, only the issuing wave reads it.  The issue
//     cursor runs ahead of the compute cursor through segment, stage and block boundaries.  (A ring of VGPRs was built
//     first: with a run-time slot index hipcc copies the in-flight destination registers at every join -- copies of data
//     that has not landed -- and with compile-time indices the stage boundaries would have to be inlined U times.)
//   * hand-off between stages: outputs are agent-scope (sc1, write-through) stores, drained (s_waitcnt vmcnt(0)) by the
//     storing wave before ONE lane arrives at the barrier; inputs are sc1 loads issued only after the barrier has opened
//     (cdna_hip_programming.md Guideline 16, R1; no cache-wide fence anywhere).  Barrier: 32 arrival counters + one top
//     counter + 32 generation words (tools/ubench/grid_barrier2.hip: 1.5-1.8 us).  The words are NEVER reset: they count
//     barriers across launches (a `base` word holds the number completed by earlier launches; workgroup 0 advances it at
//     the end of a launch), compared modulo 2^32.  No memset node: on ROCm 7.2 a hipMemsetAsync captured into a hipGraph
//     filled this block with its own ADDRESS instead of zeros on replay (tools/exp_engine_dbg.py), and it would cost a
//     graph node per token besides.  The caller zero-fills the block once (and again after an error or a grid change).
//   * attention stage: amq::attn_decode_kernel's arithmetic, expression for expression (512 active threads per head), on
//     workgroups 0 .. n_heads-1; bit-identical to amq_attn_decode_cur_f16.
#include "amq_common.cuh"
#include "amq_kernels.h"

namespace amq {

constexpr int ENG_SLOT = 1024 + 64;      // bytes of one ring slot: packed tile (<= 1 KiB) + 16 (scale, zero) pairs
constexpr int EW = 16;                   // waves per workgroup
constexpr int ET = EW * 64;
constexpr int ENG_NG = 32;               // arrival groups
constexpr int ENG_LINE = 64;             // uint32 per 256-byte line
constexpr unsigned ENG_SPIN_LIMIT = 1u << 18;
constexpr int ENG_D = 128;               // head_dim
constexpr int EA_THREADS = 512;          // active threads of the attention stage (= attn_decode_kernel's block)
constexpr int EA_GROUPS = EA_THREADS / 16;
constexpr int EA_PF = 6;                 // K / V rows per 16-lane group held in registers (192 keys; longer contexts continue from memory)

// sync workspace (uint32 words): line 0 top counter, lines 1..32 group counters, lines 33..64 generation words, line 65 error
// words, line 66: base = barriers completed by earlier launches
constexpr int ENG_SYNC_WORDS = (3 + 2 * ENG_NG) * ENG_LINE;

struct EngLinearD { const void* qw; const void* mt; int n_rt; int key; };   // key = bits * 2 + mode
struct EngBlockD { EngLinearD lin[7]; const void* ln1; const void* ln2; void* kc; void* vc; };

struct EngArgs {
    const EngBlockD* blocks; int n_block;
    int H, I, n_heads, n_kv_heads, max_seq;
    float eps;
    _Float16* x;                                   // [H] residual stream: read at entry, final value at exit
    char* scratch;                                 // hand-off vectors q, k, v, att, gate, up (eng_vec)
    const void* state;                             // step-state block: fp16 cos/sin [64][2], int32 pos @256, int32 err @260
    unsigned* sync;
    int lds_x_bytes;                               // bytes of the x / attention region of the dynamic LDS
    int red_rows;                                  // row-tiles per workgroup the partial-sum buffer holds
#ifdef AMQ_ENG_STAMP
    unsigned long long* stamps;
    int stamp_block;
#endif
};

// hand-off vectors inside the scratch buffer, each on a 256-byte boundary (computed where needed: pointers held in SGPRs for the
// whole kernel cost more than a few scalar adds)
enum { EV_Q = 0, EV_K, EV_V, EV_ATT, EV_GATE, EV_UP };
__host__ __device__ __forceinline__ size_t eng_r256(size_t halves) { return (halves * 2 + 255) & ~(size_t)255; }
__host__ __device__ __forceinline__ size_t eng_vec_off(int which, int H, int I, int n_kv_heads) {
    const size_t kvd = (size_t)n_kv_heads * 128;
    size_t off = 0;
    if (which > EV_Q) off += eng_r256(H);
    if (which > EV_K) off += eng_r256(kvd);
    if (which > EV_V) off += eng_r256(kvd);
    if (which > EV_ATT) off += eng_r256(H);
    if (which > EV_GATE) off += eng_r256(I);
    return off;
}
__device__ __forceinline__ _Float16* eng_vec(const EngArgs& a, int which) {
    return (_Float16*)(a.scratch + eng_vec_off(which, a.H, a.I, a.n_kv_heads));
}

enum { EK_QKV = 0, EK_O = 1, EK_GU = 2, EK_DOWN = 3 };

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ float eng_wave_sum(float v) {      // == amq_gemv.hip wave_sum (same DPP order)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
}
__device__ __forceinline__ float eng_silu(float g) { return g / (1.0f + __expf(-g)); }

// agent-scope (sc1) accesses of hand-off data
__device__ __forceinline__ __amdgpu_buffer_rsrc_t eng_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ h8 eng_load16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));      // aux 16 = sc1
}
__device__ __forceinline__ _Float16 eng_ldh_sc1(const _Float16* p) {
    return __builtin_bit_cast(_Float16, __hip_atomic_load((const unsigned short*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void eng_sth_sc1(_Float16* p, _Float16 v) {
    __hip_atomic_store((unsigned short*)p, __builtin_bit_cast(unsigned short, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef AMQ_ENG_STAMP
// diagnostic build (-DAMQ_ENG_STAMP): workgroup w, block `stamp_block`: slot s <- 100 MHz realtime counter (comparable across CUs)
#define ENG_STAMP(slot_) do { if (a.stamps && threadIdx.x == 0 && b == a.stamp_block) a.stamps[((size_t)blockIdx.x * 64 + (slot_))] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ENG_STAMP(slot_) do { } while (0)
#endif

// ---------------------------------------------------------------- stage description (wave-uniform)
struct StageD {
    const EngLinearD* lin;      // first of nseg consecutive linears
    int nseg, K, pro;
    const _Float16* x;          // input vector (hand-off buffer)
    const _Float16* x2;         // up (SiLU*mul) or gamma (RMSNorm)
    _Float16* y0; _Float16* y1; _Float16* y2;
    bool residual;              // y0 += (in place on the residual stream)
};

// `img`: the block table in LDS (copied there at kernel entry: read from global memory -- dependent scalar loads, cold at every
// stage -- a change of the issue cursor's run cost ~4000 cycles, tools/stamp_engine.py: 655 cycles per TILE)
__device__ __forceinline__ StageD eng_stage(const EngArgs& a, const EngBlockD* img, int gstage) {
    const EngBlockD* blk = img + (gstage >> 2);
    StageD s;
    s.y1 = s.y2 = nullptr;
    switch (gstage & 3) {
        case EK_QKV:
            s.lin = blk->lin; s.nseg = 3; s.K = a.H; s.pro = PRO_RMSNORM; s.x = a.x; s.x2 = (const _Float16*)blk->ln1;
            s.y0 = eng_vec(a, EV_Q); s.y1 = eng_vec(a, EV_K); s.y2 = eng_vec(a, EV_V); s.residual = false; break;
        case EK_O:
            s.lin = blk->lin + 3; s.nseg = 1; s.K = a.H; s.pro = PRO_NONE; s.x = eng_vec(a, EV_ATT); s.x2 = nullptr;
            s.y0 = a.x; s.residual = true; break;
        case EK_GU:
            s.lin = blk->lin + 4; s.nseg = 2; s.K = a.H; s.pro = PRO_RMSNORM; s.x = a.x; s.x2 = (const _Float16*)blk->ln2;
            s.y0 = eng_vec(a, EV_GATE); s.y1 = eng_vec(a, EV_UP); s.residual = false; break;
        default:
            s.lin = blk->lin + 6; s.nseg = 1; s.K = a.I; s.pro = PRO_SILU_MUL; s.x = eng_vec(a, EV_GATE); s.x2 = eng_vec(a, EV_UP);
            s.y0 = a.x; s.residual = true; break;
    }
    return s;
}
__host__ __device__ __forceinline__ int eng_stage_rowtiles(int H, int I, int n_kv_heads, int kind) {   // row-tiles of a stage (same for every block)
    const int kvd = n_kv_heads * 128;
    return kind == EK_QKV ? (H + 2 * kvd) >> 4 : kind == EK_GU ? (2 * I) >> 4 : H >> 4;
}
// this workgroup's contiguous range of a stage's row-tiles: [first, end) = [wg * T / P, (wg + 1) * T / P) -- computed once per
// kernel into LDS (grange[2 * kind], [2 * kind + 1]): a division per stage per wave is scalar code the hot path does not need
__device__ __forceinline__ void eng_fill_ranges(const EngArgs& a, int* grange, int* rinfo, int wg, int P) {
    if (threadIdx.x < 4) {
        const unsigned T = (unsigned)eng_stage_rowtiles(a.H, a.I, a.n_kv_heads, (int)threadIdx.x);
        grange[2 * threadIdx.x] = (int)(((unsigned)wg * T) / (unsigned)P);              // wg * T < 2^31 (capi bounds)
        grange[2 * threadIdx.x + 1] = (int)((((unsigned)wg + 1u) * T) / (unsigned)P);
    }
    if (threadIdx.x >= 64 && threadIdx.x < 71) {              // per linear of a block: this workgroup's piece of that segment
        const int e = (int)threadIdx.x - 64;
        const int kind = e < 3 ? EK_QKV : e == 3 ? EK_O : e < 6 ? EK_GU : EK_DOWN;
        const int kvd = a.n_kv_heads * 128;
        const int nrt[7] = {a.H >> 4, kvd >> 4, kvd >> 4, a.H >> 4, a.I >> 4, a.I >> 4, a.H >> 4};
        const int first = e < 3 ? 0 : e == 3 ? 3 : e < 6 ? 4 : 6;
        int cum = 0;
        for (int q = first; q < e; ++q) cum += nrt[q];
        const unsigned T = (unsigned)eng_stage_rowtiles(a.H, a.I, a.n_kv_heads, kind);
        const int g0 = (int)(((unsigned)wg * T) / (unsigned)P), g1 = (int)((((unsigned)wg + 1u) * T) / (unsigned)P);
        const int lo = g0 > cum ? g0 : cum, hi = g1 < cum + nrt[e] ? g1 : cum + nrt[e];
        rinfo[2 * e] = lo - cum;
        rinfo[2 * e + 1] = hi > lo ? hi - lo : 0;
    }
}

// ---------------------------------------------------------------- the weight ring (LDS-DMA, wave-private)
// issue cursor of one wave.  Everything is wave-uniform and deliberately small: it lives in SGPRs for the whole kernel.
#define ENG_UNI(x_) __builtin_amdgcn_readfirstlane(x_)
struct Issue {
    uint32_t qw_lo, qw_hi;      // base of the current segment's packed payload
    uint32_t mt_lo, mt_hi;      // base of the current segment's (scale, zero) pairs
    int tile;                   // next tile of this wave: row-tile * G + k-tile
    int bits;
    int left_i, left_j;         // k-tiles left in the current row-tile / row-tiles left in the piece (both incl. the current)
    int nt, row_adv;            // k-tiles per row-tile of this wave; tile increment from a row-tile's last k-tile to the next's first
    int gstage, seg;            // where the cursor is: GEMV stage (block * 4 + kind), segment
    int slot_addr;              // LDS byte address of the slot the next tile goes to
#ifdef AMQ_ENG_CYCLES
    unsigned long long c_wait, c_lds, c_issue, c_math, n_tiles, c_next;     // diagnostic: shader-clock cycles of this wave per phase
#endif
};
#ifdef AMQ_ENG_CYCLES
#define ENG_T() __builtin_amdgcn_s_memtime()
#define ENG_ACC(field_, t0_) do { is.field_ += ENG_T() - (t0_); } while (0)
#else
#define ENG_T() 0ull
#define ENG_ACC(field_, t0_) do { (void)(t0_); } while (0)
#endif

// moves the cursor to the next non-empty (segment of a stage) piece of this workgroup's work; gstage >= 4 * n_block: exhausted.
// rinfo (LDS, filled once per kernel): for linear e = 0 .. 6 of a block (q k v | o | gate up | down) the workgroup's piece of
// that segment: rinfo[2e] = first row-tile (segment-local), rinfo[2e + 1] = row-tiles (0: none).
__device__ __forceinline__ void eng_next_run(const EngArgs& a, Issue& is, const EngBlockD* img, const int* rinfo, int wave) {
    for (;;) {
        const int nstage = 4 * a.n_block;
        if (is.gstage >= nstage) return;
        const int kind = is.gstage & 3;
        const int nseg = kind == EK_QKV ? 3 : kind == EK_GU ? 2 : 1;
        is.seg = ENG_UNI(is.seg + 1);
        if (is.seg >= nseg) { is.gstage = ENG_UNI(is.gstage + 1); is.seg = -1; continue; }
        const int e = (kind == EK_QKV ? 0 : kind == EK_O ? 3 : kind == EK_GU ? 4 : 6) + is.seg;
        const int j0 = ENG_UNI(rinfo[2 * e]), cnt = ENG_UNI(rinfo[2 * e + 1]);
        const int G = (kind == EK_DOWN ? a.I : a.H) >> 7;
        const int nt = (G - wave + EW - 1) / EW;
        if (cnt > 0 && nt > 0) {
            const EngLinearD& l = img[is.gstage >> 2].lin[e];
            is.bits = ENG_UNI(l.key >> 1);
            const unsigned long long qb = (unsigned long long)l.qw, m = (unsigned long long)l.mt;
            is.qw_lo = ENG_UNI((uint32_t)qb); is.qw_hi = ENG_UNI((uint32_t)(qb >> 32));
            is.mt_lo = ENG_UNI((uint32_t)m); is.mt_hi = ENG_UNI((uint32_t)(m >> 32));
            is.tile = ENG_UNI(j0 * G + wave);
            is.left_i = ENG_UNI(nt); is.left_j = cnt;
            is.nt = ENG_UNI(nt); is.row_adv = ENG_UNI(G - EW * (nt - 1));
            return;
        }
    }
}

// LDS-DMA: every lane passes a 32-bit offset to a 64-bit scalar base; the data lands at M0 + 16 (4) * lane.  M0 is written in the
// same statement that uses it (cdna_hip_programming.md 5.7); inactive lanes neither load nor write.
__device__ __forceinline__ void eng_dma16(uint32_t voff, unsigned long long sbase, int lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void eng_dma4(uint32_t voff, unsigned long long sbase, int lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1 nt" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// Issue the next tile of the wave's stream into the slot at is.slot_addr: ALWAYS two vector-memory operations (packed tile,
// its meta), so the wait count of eng_ring_wait is a constant.  An exhausted stream re-reads the sync workspace (valid, unused).
template <int U>
__device__ __forceinline__ void eng_issue(const EngArgs& a, Issue& is, int ring_lo, const EngBlockD* img, const int* rinfo, int wave, int lane) {
    const bool live = is.gstage < 4 * a.n_block;
    if (live) {
        const unsigned long long qb = ((unsigned long long)is.qw_hi << 32) | is.qw_lo;
        const unsigned long long mb = ((unsigned long long)is.mt_hi << 32) | is.mt_lo;
        const uint32_t voq = (uint32_t)is.tile * (uint32_t)(256 * is.bits) + (uint32_t)lane * 16u;
        const uint32_t vom = ((uint32_t)is.tile * 16u + (uint32_t)lane) * 4u;
#ifndef AMQ_ENG_ABL_NODMA          /* timing ablations (results wrong): no weight traffic at all / no (scale, zero) DMA */
        if (lane < 16 * is.bits) eng_dma16(voq, qb, is.slot_addr);           // 64 / 48 / 32 lanes: 1024 / 768 / 512 bytes
#ifndef AMQ_ENG_ABL_NOMETA
        if (lane < 16) eng_dma4(vom, mb, is.slot_addr + 1024);
#endif
#endif
        if (--is.left_i == 0) {
            is.left_i = is.nt;
            is.tile += is.row_adv;
            if (--is.left_j == 0) { const unsigned long long tn_ = ENG_T(); eng_next_run(a, is, img, rinfo, wave); ENG_ACC(c_next, tn_); }
        } else {
            is.tile += EW;
        }
    } else {
        const unsigned long long sb = (unsigned long long)a.sync;
        eng_dma16((uint32_t)lane * 16u, sb, is.slot_addr);
        if (lane < 16) eng_dma4((uint32_t)lane * 4u, sb, is.slot_addr + 1024);
    }
    is.slot_addr = is.slot_addr + ENG_SLOT == ring_lo + U * ENG_SLOT ? ring_lo : is.slot_addr + ENG_SLOT;
}

// wait until the OLDEST slot of the ring has landed: every slot is 2 operations and U - 1 slots are younger.  (Other vector-memory
// operations of the wave -- all younger or long complete -- can only make this wait longer than needed, never too short.)
template <int U>
__device__ __forceinline__ void eng_ring_wait() {
#if defined(AMQ_ENG_ABL_NODMA)
    asm volatile("" ::: "memory");
#elif defined(AMQ_ENG_ABL_NOMETA)
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(U - 1) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * (U - 1)) : "memory");
#endif
}

// ---------------------------------------------------------------- device-wide barrier
// Every workgroup calls it the same number of times with epoch = base + 1, base + 2, ... (base: barriers of earlier launches;
// all comparisons modulo 2^32).  All stores a workgroup wants seen behind the barrier must have been drained (s_waitcnt
// vmcnt(0)) by the waves that issued them BEFORE the call.  Returns false when a poll ran into its bound or another
// workgroup raised the error word: the caller leaves the kernel (the words are then inconsistent: the host re-zeroes them).
__device__ __forceinline__ bool eng_grid_sync(unsigned* sync, unsigned epoch, int wg, int P, int* lds_flag) {
    __syncthreads();                                     // the storing waves have drained; nobody still reads last stage's LDS
    if (threadIdx.x == 0) {
        const unsigned g = (unsigned)wg % ENG_NG;
        const unsigned gsize = ((unsigned)P - g + ENG_NG - 1) / ENG_NG;
        const unsigned ngroups = (unsigned)P < ENG_NG ? (unsigned)P : ENG_NG;
        unsigned* const top = sync;
        unsigned* const grp = sync + (1 + g) * ENG_LINE;
        unsigned* const gen = sync + (1 + ENG_NG) * ENG_LINE;
        unsigned* const err = sync + (1 + 2 * ENG_NG) * ENG_LINE;
        bool released = false;
        int bad = 0;
        const unsigned prev = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gsize * epoch - 1) {
            const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == ngroups * epoch - 1) {
                for (unsigned i = 0; i < ngroups; ++i)
                    __hip_atomic_store(gen + i * ENG_LINE, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                released = true;
            }
        }
        if (!released) {
            unsigned* const mine = gen + g * ENG_LINE;
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
                ++spins;
                if ((spins & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { bad = 1; break; }
                if (spins > ENG_SPIN_LIMIT) {             // error word: 0x10000 | epoch; the word behind it: the workgroup that gave up first
                    if (__hip_atomic_exchange(err, 0x10000u | epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                        __hip_atomic_store(err + 1, (unsigned)wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bad = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        *lds_flag = bad;
    }
    __syncthreads();
    return *lds_flag == 0;
}

// ---------------------------------------------------------------- x staging (all 1024 threads)
// The expressions and the summation order of amq_gemv.hip's x_issue / x_finish<PRO, 16, XCH> (decode fast path).
template <int PRO, int XCH>
__device__ __forceinline__ void eng_stage_x(const StageD& s, float eps, _Float16* xl, float* reds) {
    const int tid = threadIdx.x;
    const int chunks = s.K >> 3, last = chunks - 1;
    const __amdgpu_buffer_rsrc_t rx = eng_rsrc(s.x, (unsigned)s.K * 2u);
    h8 v[XCH], w[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        int c = tid + i * ET;
        c = c < last ? c : last;
        v[i] = eng_load16_sc1(rx, (unsigned)c * 16u);
        if (PRO == PRO_SILU_MUL) w[i] = eng_load16_sc1(eng_rsrc(s.x2, (unsigned)s.K * 2u), (unsigned)c * 16u);
        if (PRO == PRO_RMSNORM) w[i] = *(const h8*)(s.x2 + 8 * c);          // gamma: constant data
    }
    float rstd = 1.0f;
    if (PRO == PRO_RMSNORM) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            if (tid + i * ET < chunks) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { float f = (float)v[i][e]; ss += f * f; }
            }
        }
        ss = eng_wave_sum(ss);
        if ((tid & 63) == 0) reds[tid >> 6] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < EW; ++q) tot += reds[q];
        rstd = rsqrtf(tot / (float)s.K + eps);
    }
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + i * ET;
        if (c < chunks) {
            h8 r;
            if (PRO == PRO_NONE) {
                r = v[i];
            } else if (PRO == PRO_SILU_MUL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { _Float16 sg = (_Float16)eng_silu((float)v[i][e]); r[e] = sg * w[i][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { _Float16 nrm = (_Float16)((float)v[i][e] * rstd); r[e] = w[i][e] * nrm; }
            }
            *(h8*)(xl + 8 * c) = r;
        }
    }
    __syncthreads();
}

template <int PRO>
__device__ __forceinline__ void eng_stage_x_k(const StageD& s, float eps, _Float16* xl, float* reds) {
    const int chunks = s.K >> 3;
    if (chunks <= ET) eng_stage_x<PRO, 1>(s, eps, xl, reds);
    else if (chunks <= 2 * ET) eng_stage_x<PRO, 2>(s, eps, xl, reds);
    else eng_stage_x<PRO, 4>(s, eps, xl, reds);
}

// ---------------------------------------------------------------- one (segment of a stage) piece: consume its tiles
template <int BITS, int MODE, int U>
__device__ __forceinline__ void eng_run(const EngArgs& a, Issue& is, int& cslot, int lds0, int ring_off, int j0, int j1, int nt, int jj, int stage_rows,
                                        const _Float16* xl, float* red, const unsigned char* smem, const EngBlockD* img, const int* rinfo,
                                        int wave, int lane) {
    const int o = lane >> 4;
    const _Float16* xrow = xl + 8 * o;                       // M = 1: every A row of the MFMA is x row 0
    for (int j = j0; j < j1; ++j, ++jj) {
#ifndef AMQ_ENG_NO_PRIO
        // Issue priority falls with the wave's progress through the stage (quartiles of its row-tiles): the SIMD arbiter is
        // oldest-first, so without it the four oldest waves of the workgroup race ahead and the youngest finish the stage
        // alone at the two-wave issue efficiency (first timeline: wave 0 done 2.5 us before the workgroup's last wave;
        // the same effect and remedy as amq_gemv.hip's, profiles/r01b_gemv_prio.txt)
        {
            const int q4 = (4 * jj) / (stage_rows > 0 ? stage_rows : 1);
            if (q4 <= 0) __builtin_amdgcn_s_setprio(3);
            else if (q4 == 1) __builtin_amdgcn_s_setprio(2);
            else if (q4 == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nt; ++i) {
            const unsigned long long t0_ = ENG_T();
            eng_ring_wait<U>();
            ENG_ACC(c_wait, t0_);
            const unsigned long long t1_ = ENG_T();
            // the lane's 4 * BITS payload bytes lie at 4 * BITS * lane of the slot (the tile as it lies in HBM), its row's pair behind
            const unsigned char* sp = smem + cslot;
            uint32_t wr[4];
            if (BITS == 4) { const u4 v = *(const u4*)(sp + 16 * lane); wr[0] = v.x; wr[1] = v.y; wr[2] = v.z; wr[3] = v.w; }
            else if (BITS == 2) { const u2 v = *(const u2*)(sp + 8 * lane); wr[0] = v.x; wr[1] = v.y; }
            else { const uint32_t* q = (const uint32_t*)(sp + 12 * lane); wr[0] = q[0]; wr[1] = q[1]; wr[2] = q[2]; }
            const h2 meta = *(const h2*)(sp + 1024 + 4 * (lane & 15));
            // the tile's four x operands are requested with it: ONE exposed LDS round trip per tile (requested next to each MFMA
            // they cost four: 4 waves per SIMD do not hide them -- tools/stamp_engine.py, first timeline: ~1 us per tile and wave)
            const int kbase = (wave + EW * i) << 7;
            h8 av[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) av[t] = *(const h8*)(xrow + kbase + 32 * t);
            // the slot is refilled only when its contents are in registers (the DMA writes LDS behind the LDS reads' backs)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ENG_ACC(c_lds, t1_);
            const unsigned long long t2_ = ENG_T();
            eng_issue<U>(a, is, lds0 + ring_off, img, rinfo, wave, lane);
            ENG_ACC(c_issue, t2_);
            [[maybe_unused]] const unsigned long long t3_ = ENG_T();      // (read by the -DAMQ_ENG_CYCLES build only)
            cslot = cslot + ENG_SLOT == ring_off + U * ENG_SLOT ? ring_off : cslot + ENG_SLOT;
            h2 wv[16];
            dequant_lane_sd<BITS, MODE>(wr, meta, wv);
            f4 c_ = acc;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                h8 b;
#pragma unroll
                for (int p = 0; p < 4; ++p) { b[2 * p] = wv[4 * t + p].x; b[2 * p + 1] = wv[4 * t + p].y; }
                c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[t], b, c_, 0, 0, 0);
            }
            acc = c_;
#ifdef AMQ_ENG_CYCLES
            asm volatile("" :: "v"(acc));
            ENG_ACC(c_math, t3_);
            is.n_tiles += 1;
#endif
        }
        if (lane < 16) red[jj * (EW * 16) + wave * 16 + lane] = acc[0];
    }
}

// one GEMV stage for this workgroup: x has been staged in xl
template <int U>
__device__ __forceinline__ void eng_gemv_stage(const EngArgs& a, int gstage, Issue& is, int& cslot, int lds0, int ring_off, const _Float16* xl,
                                               float* red, const unsigned char* smem, const EngBlockD* img, const int* grange,
                                               const int* rinfo, int wave, int lane) {
    const StageD s0 = eng_stage(a, img, gstage);
    const StageD& s = s0;
    const int g0 = ENG_UNI(grange[2 * (gstage & 3)]), g1 = ENG_UNI(grange[2 * (gstage & 3) + 1]);
    const int G = s.K >> 7;
    const int nt = (G - wave + EW - 1) / EW;
    // residual value of this thread's output (thread c < nout finishes output c; an in-place stage owns one or two row-tiles per
    // workgroup on a full grid): requested now and forced to land before the tile loop, so that the compiler's wait for it
    // cannot drain the ring later
    const int nout = (g1 - g0) * 16;
    const int tid = (int)threadIdx.x;
    uint32_t res0 = 0u;
    if (s.residual && tid < nout) {
        res0 = __hip_atomic_load((const unsigned short*)(s.y0 + (size_t)g0 * 16 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" : "+v"(res0));
    }
    int cum = 0, jj = 0;
    for (int q = 0; q < s.nseg; ++q) {
        const EngLinearD& l = s.lin[q];
        const int lo = g0 > cum ? g0 : cum, hi = g1 < cum + l.n_rt ? g1 : cum + l.n_rt;
        if (lo < hi) {
            const int j0 = lo - cum, j1 = hi - cum;
            if (nt > 0) {
                switch (l.key) {
                    case 4 * 2 + MODE_HQQ: eng_run<4, MODE_HQQ, U>(a, is, cslot, lds0, ring_off, j0, j1, nt, jj, g1 - g0, xl, red, smem, img, rinfo, wave, lane); break;
                    case 3 * 2 + MODE_HQQ: eng_run<3, MODE_HQQ, U>(a, is, cslot, lds0, ring_off, j0, j1, nt, jj, g1 - g0, xl, red, smem, img, rinfo, wave, lane); break;
                    case 2 * 2 + MODE_HQQ: eng_run<2, MODE_HQQ, U>(a, is, cslot, lds0, ring_off, j0, j1, nt, jj, g1 - g0, xl, red, smem, img, rinfo, wave, lane); break;
                    case 4 * 2 + MODE_FMA: eng_run<4, MODE_FMA, U>(a, is, cslot, lds0, ring_off, j0, j1, nt, jj, g1 - g0, xl, red, smem, img, rinfo, wave, lane); break;
                    case 3 * 2 + MODE_FMA: eng_run<3, MODE_FMA, U>(a, is, cslot, lds0, ring_off, j0, j1, nt, jj, g1 - g0, xl, red, smem, img, rinfo, wave, lane); break;
                    default:               eng_run<2, MODE_FMA, U>(a, is, cslot, lds0, ring_off, j0, j1, nt, jj, g1 - g0, xl, red, smem, img, rinfo, wave, lane); break;
                }
            } else if (lane < 16) {                        // K < 128 * 16: this wave owns no tile, its partials are zero
                for (int j = j0; j < j1; ++j) red[(jj + j - j0) * (EW * 16) + wave * 16 + lane] = 0.f;
            }
            jj += j1 - j0;
        }
        cum += l.n_rt;
    }
#ifndef AMQ_ENG_NO_PRIO
    __builtin_amdgcn_s_setprio(3);                            // epilogue, barrier and the next stage's prologue at top priority
#endif
    { const int b = gstage >> 2; (void)b; ENG_STAMP(1 + 8 * (gstage & 3) + 2); }
    __syncthreads();                                          // every wave's partials of every row-tile are in LDS
    { const int b = gstage >> 2; (void)b; ENG_STAMP(1 + 8 * (gstage & 3) + 3); }
    if (tid < nout || nout > ET) {
        const StageD s = eng_stage(a, img, gstage);          // (output pointers are derived again here rather than held in SGPRs)
        // fixed-order sum over the 16 waves, one fp16 rounding, residual as a separate fp16 add (amq_gemv.hip AMQ_FINISH); output c
        // by thread c: every storing wave drains its own stores before the barrier's workgroup barrier (Guideline 16, R1)
        const int n0 = s.lin[0].n_rt, n1 = s.nseg > 1 ? n0 + s.lin[1].n_rt : 0x7fffffff;
        auto finish = [&](int c, bool have_res, uint32_t resv) {
            const float* rp = red + (c >> 4) * (EW * 16) + (c & 15);
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < EW; ++q) tot += rp[q * 16];
            _Float16 y = (_Float16)tot;
            const int g = g0 + (c >> 4);
            _Float16* dst = g < n0 ? s.y0 + (size_t)g * 16 : g < n1 ? s.y1 + (size_t)(g - n0) * 16 : s.y2 + (size_t)(g - n1) * 16;
            if (s.residual) {
                if (!have_res) resv = __hip_atomic_load((const unsigned short*)(dst + (c & 15)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                y = __builtin_bit_cast(_Float16, (unsigned short)resv) + y;
            }
            eng_sth_sc1(dst + (c & 15), y);
        };
        if (tid < nout) finish(tid, true, res0);
        for (int c = tid + ET; c < nout; c += ET) finish(c, false, 0u);        // small grids only: more than 64 row-tiles per workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // publish: every storing wave drains before the barrier's arrival
    }
    { const int b = gstage >> 2; (void)b; ENG_STAMP(1 + 8 * (gstage & 3) + 4); }
}

// ---------------------------------------------------------------- attention stage (workgroup = one head)
__device__ __forceinline__ float eng_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ float eng_row16_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));
    return v;
}
__device__ __forceinline__ float eng_wave_max_dpp(float v) {
    v = eng_row16_max(v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48))));
}
__device__ __forceinline__ float eng_wave_sum_dpp(float v) {
    v = eng_row16_sum(v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
}

// amq::attn_decode_kernel (amq_decode.hip) for head h, with the new token's q / k / v read as agent-scope loads and the head's
// output written as agent-scope stores.  Threads 512 .. 1023 only take part in the workgroup barriers.  LDS: `sm` (the x region).
__device__ __forceinline__ void eng_attention(const EngArgs& a, const EngBlockD* blk, int h, int pos, unsigned char* sm) {
    _Float16* qs = (_Float16*)sm;
    _Float16* ks = qs + ENG_D;
    _Float16* vs = qs + 2 * ENG_D;
    float* red = (float*)(sm + 6 * ENG_D);                      // [16]
    float* part = (float*)(sm + 6 * ENG_D + 64);                // [32][128]
    float* sc = (float*)(sm + 6 * ENG_D + 64 + EA_GROUPS * ENG_D * 4);   // [T]
    const int tid = threadIdx.x;
    const bool act = tid < EA_THREADS;
    const int n_heads = a.n_heads, n_kv_heads = a.n_kv_heads, max_seq = a.max_seq;
    const int group = n_heads / n_kv_heads;
    const int kvh = h / group;
    const int grp = tid >> 4, l16 = tid & 15;
    const _Float16* q = eng_vec(a, EV_Q) + (size_t)h * ENG_D;
    const _Float16* kn = eng_vec(a, EV_K) + (size_t)kvh * ENG_D;
    const _Float16* vn = eng_vec(a, EV_V) + (size_t)kvh * ENG_D;
    _Float16* kc = (_Float16*)blk->kc + (size_t)kvh * (size_t)max_seq * ENG_D;
    _Float16* vc = (_Float16*)blk->vc + (size_t)kvh * (size_t)max_seq * ENG_D;
    const int T = pos + 1;
    const int last_old = pos > 0 ? pos - 1 : 0;

    _Float16 q0 = 0, q1 = 0, k0 = 0, k1 = 0, v0 = 0, v1 = 0;
    h2 cs = {(_Float16)1.f, (_Float16)0.f};
    if (tid < 64) {
        q0 = eng_ldh_sc1(q + tid); q1 = eng_ldh_sc1(q + tid + 64);
        k0 = eng_ldh_sc1(kn + tid); k1 = eng_ldh_sc1(kn + tid + 64);
        cs = ((const h2*)a.state)[tid];
        v0 = eng_ldh_sc1(vn + tid); v1 = eng_ldh_sc1(vn + tid + 64);
    }
    h8 krow[EA_PF], vrow[EA_PF];
    if (act) {
#pragma unroll
        for (int i = 0; i < EA_PF; ++i) {
            if (EA_GROUPS * i < pos) {
                int t = grp + EA_GROUPS * i;
                t = t < last_old ? t : last_old;
                krow[i] = *(const h8*)(kc + (size_t)t * ENG_D + 8 * l16);
            }
        }
#pragma unroll
        for (int i = 0; i < EA_PF; ++i) {
            if (EA_GROUPS * i < pos) {
                int t = grp + EA_GROUPS * i;
                t = t < last_old ? t : last_old;
                vrow[i] = *(const h8*)(vc + (size_t)t * ENG_D + 8 * l16);
            }
        }
    }
    if (tid < 64) {
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
            kc[(size_t)pos * ENG_D + i] = r0;
            kc[(size_t)pos * ENG_D + i + 64] = r1;
            vc[(size_t)pos * ENG_D + i] = v0;
            vc[(size_t)pos * ENG_D + i + 64] = v1;
        }
    }
    __syncthreads();
    const float scale = rsqrtf((float)ENG_D);
    if (act) {
        const h8 qv = *(const h8*)(qs + 8 * l16);
        const h8 knew = *(const h8*)(ks + 8 * l16);
        auto score = [&](const h8& kv) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s = __builtin_amdgcn_fdot2((h2){qv[2 * e], qv[2 * e + 1]}, (h2){kv[2 * e], kv[2 * e + 1]}, s, false);
            s = eng_row16_sum(s);
            return (float)(_Float16)((float)(_Float16)s * scale);
        };
#pragma unroll
        for (int i = 0; i < EA_PF; ++i) {
            if (EA_GROUPS * i < T) {
                const int t = grp + EA_GROUPS * i;
                const float sv = score(t == pos ? knew : krow[i]);
                if (t < T && l16 == 0) sc[t] = sv;
            }
        }
        for (int t = grp + EA_GROUPS * EA_PF; t < T; t += EA_GROUPS) {
            const h8 kv = (t == pos) ? knew : *(const h8*)(kc + (size_t)t * ENG_D + 8 * l16);
            const float sv = score(kv);
            if (l16 == 0) sc[t] = sv;
        }
    }
    __syncthreads();
    float lmax = -INFINITY;
    if (act) {
        for (int t = tid; t < T; t += EA_THREADS) lmax = fmaxf(lmax, sc[t]);
        lmax = eng_wave_max_dpp(lmax);
        if ((tid & 63) == 0) red[tid >> 6] = lmax;
    }
    __syncthreads();
    float inv = 0.f;
    if (act) {
        float gmax = red[0];
#pragma unroll
        for (int w = 1; w < EA_THREADS / 64; ++w) gmax = fmaxf(gmax, red[w]);
        float lsum = 0.f;
        for (int t = tid; t < T; t += EA_THREADS) {
            const float e = __expf(sc[t] - gmax);
            sc[t] = e;
            lsum += e;
        }
        lsum = eng_wave_sum_dpp(lsum);
        if ((tid & 63) == 0) red[EA_THREADS / 64 + (tid >> 6)] = lsum;
    }
    __syncthreads();
    if (act) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < EA_THREADS / 64; ++w) tot += red[EA_THREADS / 64 + w];
        inv = 1.0f / tot;
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const h8 vnew = *(const h8*)(vs + 8 * l16);
#pragma unroll
        for (int i = 0; i < EA_PF; ++i) {
            const int t = grp + EA_GROUPS * i;
            if (EA_GROUPS * i < T && t < T) {
                const _Float16 p16 = (_Float16)(sc[t] * inv);
                const h8 vv = (t == pos) ? vnew : vrow[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += (float)p16 * (float)vv[e];
            }
        }
        for (int t = grp + EA_GROUPS * EA_PF; t < T; t += EA_GROUPS) {
            const _Float16 p16 = (_Float16)(sc[t] * inv);
            const h8 vv = (t == pos) ? vnew : *(const h8*)(vc + (size_t)t * ENG_D + 8 * l16);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += (float)p16 * (float)vv[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[grp * ENG_D + 8 * l16 + e] = o[e];
    }
    __syncthreads();
    if (tid < ENG_D) {
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < EA_GROUPS; ++g) acc += part[g * ENG_D + tid];
        eng_sth_sc1(eng_vec(a, EV_ATT) + (size_t)h * ENG_D + tid, (_Float16)acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // both storing waves drain before the barrier's arrival
    }
}

// ---------------------------------------------------------------- the kernel
template <int U>
__global__ __launch_bounds__(ET) void decode_engine_kernel(EngArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* xl = (_Float16*)smem;
    float* red = (float*)(smem + a.lds_x_bytes);                         // [red_rows][16 waves][16]
    float* reds = red + (size_t)a.red_rows * (EW * 16);                  // [16]
    int* flag = (int*)(reds + EW);
    int* grange = flag + 4;                                              // [4 kinds][first, end)
    int* rinfo = grange + 8;                                             // [7 linears][first row-tile, count]
    EngBlockD* img = (EngBlockD*)(rinfo + 16);                           // the block table (n_block entries)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x, P = gridDim.x;
    // the wave's ring: U slots behind the x / partial-sum regions.  Two views of one address: `ring_off` = byte offset into smem
    // (what the compiler-visible LDS reads use), lds0 + ring_off = the LDS byte address the DMA's M0 takes.
    const int lds0 = (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int img_bytes = (a.n_block * (int)sizeof(EngBlockD) + 15) & ~15;
    const int ring_off = ENG_UNI(a.lds_x_bytes + a.red_rows * (EW * 16) * 4 + EW * 4 + 48 + 64 + img_bytes + wave * (U * ENG_SLOT));
    eng_fill_ranges(a, grange, rinfo, wg, P);
    for (int w = threadIdx.x; w < a.n_block * (int)(sizeof(EngBlockD) / 4); w += ET) ((uint32_t*)img)[w] = ((const uint32_t*)a.blocks)[w];
    __syncthreads();

    Issue is;
    is.gstage = 0; is.seg = -1;
    is.bits = 4; is.tile = 0; is.left_i = is.left_j = 1; is.nt = 1; is.row_adv = 0;
    is.qw_lo = is.qw_hi = is.mt_lo = is.mt_hi = 0;
    is.slot_addr = lds0 + ring_off;
#ifdef AMQ_ENG_CYCLES
    is.c_wait = is.c_lds = is.c_issue = is.c_math = is.n_tiles = is.c_next = 0;
    const unsigned long long tk0_ = ENG_T();
#endif
    eng_next_run(a, is, img, rinfo, wave);
    for (int u = 0; u < U; ++u) eng_issue<U>(a, is, lds0 + ring_off, img, rinfo, wave, lane);
    int cslot = ring_off;                                                // consume cursor (offset into smem)
    const unsigned base = (unsigned)ENG_UNI(*(const int*)(a.sync + (2 + 2 * ENG_NG) * ENG_LINE));   // written by the previous launch
    unsigned epoch = base;
    const int pos = ENG_UNI(*(const int*)((const char*)a.state + 256));
    const bool pos_ok = pos >= 0 && pos < a.max_seq;
    if (!pos_ok && wg == 0 && threadIdx.x == 0) *(int*)((char*)const_cast<void*>(a.state) + 260) = 1;   // sticky error word (amq_decode.hip)

    for (int b = 0; b < a.n_block; ++b) {
        const EngBlockD* blk = img + b;
#pragma unroll 1
        for (int kind = 0; kind < 4; ++kind) {
            if (b > 0 || kind > 0) {
                if (!eng_grid_sync(a.sync, ++epoch, wg, P, flag)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
            }
            if (kind == EK_O) {
                // the attention stage sits between q/k/v and o_proj
                ENG_STAMP(1 + 8 * kind + 5);
                if (pos_ok && wg < a.n_heads) eng_attention(a, blk, wg, pos, smem);
                ENG_STAMP(1 + 8 * kind + 6);
                if (!eng_grid_sync(a.sync, ++epoch, wg, P, flag)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
            }
            ENG_STAMP(1 + 8 * kind + 0);
            const StageD s = eng_stage(a, img, 4 * b + kind);
            if (s.pro == PRO_RMSNORM) eng_stage_x_k<PRO_RMSNORM>(s, a.eps, xl, reds);
            else if (s.pro == PRO_SILU_MUL) eng_stage_x_k<PRO_SILU_MUL>(s, a.eps, xl, reds);
            else eng_stage_x_k<PRO_NONE>(s, a.eps, xl, reds);
            ENG_STAMP(1 + 8 * kind + 1);
            eng_gemv_stage<U>(a, 4 * b + kind, is, cslot, lds0, ring_off, xl, red, smem, img, grange, rinfo, wave, lane);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the trailing (unused) DMAs must not land after the workgroup has gone
    // every workgroup read `base` before its first barrier and this point lies behind the last one: nobody reads it again in this launch
    if (wg == 0 && threadIdx.x == 0) a.sync[(2 + 2 * ENG_NG) * ENG_LINE] = epoch;
#ifdef AMQ_ENG_CYCLES
    if (a.stamps && threadIdx.x == 0) {
        unsigned long long* o = a.stamps + (size_t)blockIdx.x * 64 + 40;
        o[0] = is.c_wait; o[1] = is.c_lds; o[2] = is.c_issue; o[3] = is.c_math; o[4] = is.n_tiles; o[5] = is.c_next; o[6] = ENG_T() - tk0_;
    }
#endif
}

// ---------------------------------------------------------------- host side
#ifdef AMQ_ENG_STAMP
static unsigned long long* g_eng_stamps = nullptr;
static int g_eng_stamp_block = 1;
extern "C" int amq_debug_engine_stamps(void* p, int block) { g_eng_stamps = (unsigned long long*)p; g_eng_stamp_block = block; return 0; }
#endif
size_t engine_sync_bytes() { return (size_t)ENG_SYNC_WORDS * 4; }
size_t engine_image_bytes(int n_block) { return (size_t)n_block * sizeof(EngBlockD); }

void engine_fill_image(void* image, int n_block, const EngineLinearH* lin, const void* const* ln1, const void* const* ln2,
                       void* const* kc, void* const* vc, int H, int I) {
    EngBlockD* out = (EngBlockD*)image;
    for (int b = 0; b < n_block; ++b) {
        for (int i = 0; i < 7; ++i) {
            const EngineLinearH& l = lin[b * 7 + i];
            const int K = i == 6 ? I : H;
            EngLinearD& d = out[b].lin[i];
            d.qw = l.qweight; d.mt = l.meta; d.n_rt = l.N / 16; d.key = l.bits * 2 + l.mode;
            (void)K;
        }
        out[b].ln1 = ln1[b]; out[b].ln2 = ln2[b]; out[b].kc = kc[b]; out[b].vc = vc[b];
    }
}

static int engine_red_rows(const EngineDesc& d, int P) {
    int best = 1;
    for (int k = 0; k < 4; ++k) {
        const int r = (eng_stage_rowtiles(d.H, d.I, d.n_kv_heads, k) + P - 1) / P + 1;
        if (r > best) best = r;
    }
    return best;
}

static size_t engine_x_bytes(const EngineDesc& d) {
    const size_t kmax = (size_t)(d.H > d.I ? d.H : d.I);
    size_t xb = (kmax + 8) * 2;
    const size_t att = 6 * ENG_D + 64 + (size_t)EA_GROUPS * ENG_D * 4 + (size_t)d.max_seq * 4;
    if (att > xb) xb = att;
    return (xb + 15) & ~(size_t)15;
}

static size_t engine_fixed_lds(const EngineDesc& d, int P) {
    return engine_x_bytes(d) + (size_t)engine_red_rows(d, P) * (EW * 16) * 4 + EW * 4 + 48 + 64 + (((size_t)d.n_block * sizeof(EngBlockD) + 15) & ~(size_t)15);
}
// ring depth the LDS budget allows: 6 slots per wave (102 KB of weights in flight per CU), else 4; 0: does not fit
static int engine_depth(const EngineDesc& d, int P) {
    const size_t fixed = engine_fixed_lds(d, P), cap = 160 * 1024;
    if (d.depth == 4 || d.depth == 6) return fixed + (size_t)EW * d.depth * ENG_SLOT <= cap ? d.depth : 0;
    if (fixed + (size_t)EW * 6 * ENG_SLOT <= cap) return 6;
    if (fixed + (size_t)EW * 4 * ENG_SLOT <= cap) return 4;
    return 0;
}
size_t engine_lds_bytes(const EngineDesc& d, int P) {
    const int u = engine_depth(d, P);
    return u ? engine_fixed_lds(d, P) + (size_t)EW * u * ENG_SLOT : (size_t)1 << 30;
}

hipError_t launch_decode_engine(const EngineDesc& d, hipStream_t st) {
    StreamDevice sd_(st);                                  // attributes / CU counts of the stream's device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    const int P = d.grid > 0 ? d.grid : cus;
    EngArgs a{};
    a.blocks = (const EngBlockD*)d.blocks_dev; a.n_block = d.n_block;
    a.H = d.H; a.I = d.I; a.n_heads = d.n_heads; a.n_kv_heads = d.n_kv_heads; a.max_seq = d.max_seq; a.eps = d.eps;
    a.x = (_Float16*)d.x;
    a.scratch = (char*)d.scratch;
    a.state = d.state; a.sync = (unsigned*)d.sync;
    a.lds_x_bytes = (int)engine_x_bytes(d);
    a.red_rows = engine_red_rows(d, P);
#ifdef AMQ_ENG_STAMP
    a.stamps = g_eng_stamps; a.stamp_block = g_eng_stamp_block;
#endif
    const int u = engine_depth(d, P);
    if (u == 0) return hipErrorInvalidValue;
    const size_t lds = engine_lds_bytes(d, P);
    static unsigned long long attr6_done = 0, attr4_done = 0;
    e = u == 6 ? ensure_dyn_lds(attr6_done, (const void*)decode_engine_kernel<6>, 160 * 1024)
               : ensure_dyn_lds(attr4_done, (const void*)decode_engine_kernel<4>, 160 * 1024);
    if (e != hipSuccess) return e;
    if (u == 6) hipLaunchKernelGGL(decode_engine_kernel<6>, dim3(P), dim3(ET), lds, st, a);
    else hipLaunchKernelGGL(decode_engine_kernel<4>, dim3(P), dim3(ET), lds, st, a);
    return hipGetLastError();
}

size_t engine_scratch_bytes(int H, int I, int n_kv_heads) { return eng_vec_off(EV_UP, H, I, n_kv_heads) + eng_r256(I); }

}  // namespace amq
