// amq_kernels.h -- internal launcher interfaces between the .hip translation
// units and the C-ABI layer (amq_capi.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace amq {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: a process that launches on a second GPU
// must set it there too, and a failed first call must not stay cached for the life of the process.  One bit per
// (call site, device ordinal): set on success only.  A racing second thread at worst repeats the (idempotent) call.
inline hipError_t ensure_dyn_lds(unsigned long long& done_mask, const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&done_mask, __ATOMIC_RELAXED) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_fetch_or(&done_mask, bit, __ATOMIC_RELAXED);
    return e;
}

// Launchers that set a per-device kernel attribute or size a grid by the CU count must do so for the device the STREAM belongs to, not for
// whatever device happens to be current: a process that holds weights on cuda:1 while cuda:0 is current (HF device_map, no
// set_device) launches on cuda:1's stream.  Makes the stream's device current for the scope; the null stream means
// "the current device" and changes nothing.
struct StreamDevice {
    int prev = -1;
    bool switched = false;
    explicit StreamDevice(hipStream_t st) {
        hipDevice_t dev = 0;
        if (st == nullptr || hipGetDevice(&prev) != hipSuccess || hipStreamGetDevice(st, &dev) != hipSuccess) return;
        if ((int)dev != prev && hipSetDevice((int)dev) == hipSuccess) switched = true;
    }
    ~StreamDevice() { if (switched) (void)hipSetDevice(prev); }
    StreamDevice(const StreamDevice&) = delete;
    StreamDevice& operator=(const StreamDevice&) = delete;
};

enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_SILU_MUL = 2,
       PRO_RMSNORM_SUMS = 3 /* internal (amq_gemv_grouped_sums_f16): RMSNorm whose sums of squares arrive as per-row-tile partials from the launch that produced x */ };
enum { FMT_HQQ = 0, FMT_GPTQ = 1, FMT_AWQ = 2 };
constexpr int GEMV_MAX_SEG = 4;

struct GemvSeg {
    const void* qweight;   // native AMQ-T16 payload
    const void* meta;      // native (scale, zero|c) half2
    const void* bias;      // fp16 [N] or null
    const void* residual;  // fp16 [M, y_stride] or null: y = residual + (xW^T + bias)
    void* y;               // fp16 [M, y_stride]
    int N;
    int bits;              // 2 | 3 | 4
    int mode;              // MODE_HQQ | MODE_FMA
    int wg_begin;          // first workgroup of this segment          (filled by launch_gemv)
    int wg_count;          // workgroups serving this segment          (filled by launch_gemv)
    int n_rt;              // row-tiles (N / 16)                        (filled by launch_gemv)
    int y_stride;          // elements between output rows
};

struct GemvArgs {
    GemvSeg seg[GEMV_MAX_SEG];
    int nseg;
    int M, K;
    int x_stride;          // elements between x rows
    const void* x;         // fp16 [M, x_stride]   (PRO_SILU_MUL: gate)
    const void* x2;        // PRO_SILU_MUL: up
    const void* gamma;     // PRO_RMSNORM: fp16 [K]
    float eps;
    int prologue;
    int flags;             // GEMV_FLAG_*
    int force_waves;       // 0 = auto, else 4 / 8 / 16 waves per workgroup
    int force_depth;       // 0 = auto, else 2 / 4 tile loads in flight per wave
    int force_rpt;         // 0 = auto, else row-tiles per workgroup
    int gp;                // (scale, zero) pairs per (row, tile) of every segment's meta: 0 / 1 (groups of 128), 2 (64), 4 (32)
    const void* sums_in;   // PRO_RMSNORM_SUMS: fp32 [M][K / 16] partial sums of squares of the x rows (one per 16 columns: what the producing launch's row-tiles left)
    void* sums_out;        // 5 .. 8-row launches of ONE segment: fp32 [M][N / 16], partial sums of squares of the y rows this launch writes (or null)
};
// how a segment's row-tiles are dealt to its workgroups: the first n_rt % wg_count workgroups walk one row-tile more than the others
// (packed base | rem << 12: base <= 4095 row-tiles per workgroup, rem < 2^19 workgroups -- launch_gemv refuses what does not fit)
constexpr int GEMV_SPLIT_BASE_BITS = 12, GEMV_SPLIT_BASE_MASK = (1 << GEMV_SPLIT_BASE_BITS) - 1;
inline int gemv_split(int n_rt, int wg_count) { return (n_rt / wg_count) | ((n_rt % wg_count) << GEMV_SPLIT_BASE_BITS); }
enum { GEMV_FLAG_DOT = 1, GEMV_FLAG_LINEAR = 2, GEMV_FLAG_RS128 = 4 /* internal: half-size cross-wave sum buffer (<= 8 rows) */,
       GEMV_FLAG_GS = 8 /* group-scale arithmetic for the two-rounding (HQQ) segments */,
       GEMV_FLAG_RS64 = 16 /* internal: the 2 .. 4-row kernels */,
       GEMV_FLAG_PH2 = 32 /* internal: x staged in two K phases (5 .. 8 rows of a K whose rows do not fit LDS whole) */ };
constexpr int GEMV_MAX_M = 16;

size_t gemv_lds_bytes(int M, int K, int copies);
size_t gemv_lds_bytes_rows(int M, int K, int nw);                   // the 2 .. 8-row kernels
size_t gemv_min_lds_bytes(int M, int K, bool plain, bool norm = true);   // what the C ABI checks against the LDS limit (norm: an RMSNorm prologue or strided x rows, which cannot be staged in K phases)
bool gemv_rows_phased(int M, int K, bool plain, bool norm);      // the launch stages x in two K phases (norm: as above -- callers pass `RMSNorm prologue || x_stride != K`)
hipError_t launch_gemv(GemvArgs& a, hipStream_t st);

// y[M,N] = x[M,K] . W^T for any M (MFMA, LDS-staged x tiles)
struct GemmArgs {
    const void* x; const void* qweight; const void* meta; const void* bias; void* y;
    int M, N, K, bits, mode, x_stride, y_stride;
    float* ws;      // split-K partials [splits][M][N] fp32 (or null)
    int splits;     // >= 1
    const void* residual;   // fp16 [M, y_stride] added to the rounded result (y = residual + fp16(acc (+ bias))), or null
    const void* gate;       // few-row kernel only: fp16 [M, y_stride]; y = fp16(silu(gate)) * fp16(acc (+ bias)) (LlamaMLP), or null
    void* w16;              // GEMM_ROUTE_DEQ: caller-owned scratch for the dequantized fp16 weights [N, K] (else null)
    int gp;                 // meta pairs per (row, tile): 0 / 1, or 2 / 4 for groups of 64 / 32 -- those run GEMM_ROUTE_DEQ (or the GEMV kernel up to 16 rows) only
};
// route: which kernel family serves the launch (AUTO: by shape; the others force one for tests / A-B tools)
enum { GEMM_ROUTE_AUTO = 0, GEMM_ROUTE_TILED = 1, GEMM_ROUTE_SKINNY = 2, GEMM_ROUTE_RING = 3, GEMM_ROUTE_RING128 = 4, GEMM_ROUTE_WS = 5,
       GEMM_ROUTE_DEQ = 6 /* launch_dequantize into a.w16, then launch_gemm_f16w */ };
// an RMSNorm of the result rows into fragment order behind the GEMM (gamma [N], xf: amq_xfrag_bytes(M, N)); launch_gemm makes it part of the split-K
// reduce where the launch has one (splitk_reduce_norm_kernel) and a launch of its own otherwise; `done` is launch_gemm's bookkeeping
struct GemmNorm { const void* gamma; float eps; void* xf; bool done; };
hipError_t launch_gemm(const GemmArgs& a, hipStream_t st, int route = GEMM_ROUTE_AUTO, GemmNorm* norm = nullptr);
bool gemm_gate_fused(const GemmArgs& a, int route);                 // a.gate applied in the kernel's epilogue (else: element-wise launch behind it)
hipError_t launch_gemm_ring(const GemmArgs& a, hipStream_t st, int bm = 0);      // amq_gemm_ring.hip: 256 (or 128) x 256 tiles, LDS rings, counted waits; bm 0 = by shape
bool gemm_ring_ok(const GemmArgs& a);
hipError_t launch_gemm_ws(const GemmArgs& a, hipStream_t st);                  // amq_gemm_ws.hip: 256 x 128 tiles, 4 MFMA waves + 4 DMA / unpack waves (same shape conditions: gemm_ring_ok)
int gemm_ring_rows(int M, int N);                                   // 256 / 128 rows per ring tile, 0: launch too small
int gemm_many_rows_plan(int M, int N);                              // GEMM_ROUTE_AUTO: 256 / 128 ring tile rows, -1 the wave-specialised kernel, 0 none
bool gemm_takes_ring(int M, int N, int K);                          // GEMM_ROUTE_AUTO's choice for the shape
hipError_t launch_gemm_xfrag(const GemmArgs& a, hipStream_t st);     // a.x in fragment order (launch_xfrag)
hipError_t launch_xfrag(const void* src, void* xf, int M, int K, long stride_m, long stride_kt, hipStream_t st);
// several linears over the same fragment-ordered x as segments of one few-row launch (uses qweight / meta / bias / residual / y / N / bits / mode / y_stride)
hipError_t launch_gemm_xfrag_grouped(const void* xf, int M, int K, const GemvSeg* segs, int nseg, hipStream_t st, int form = 0, int nsub_forced = 0);   // form: AMQ_FEWROW_*
hipError_t launch_gemm_fewrow_stream_grouped(const void* xf, int M, int K, const GemvSeg* segs, int nseg, hipStream_t st, int nsub_forced = 0);   // amq_gemm_fewrow.hip
int fewrow_stream_nsub(const int* seg_blocks, int nseg, int row_groups, int cus);      // column blocks per workgroup of that kernel
int gemm_pick_splits(int M, int N, int K, int route = GEMM_ROUTE_AUTO);
// amq_gemm_f16.hip: y = x . W^T with W as fp16 [N, K] (the dequantized weights, or any dense fp16 matrix): 256 x 256 tiles, two wave
// groups in ping-pong, no VALU in the K loop; bias / residual / gate epilogues as GemmArgs
bool gemm_f16w_ok(int M, int N, int K, int x_stride, int y_stride);
bool gemm_fine_takes_deq(int M, int N, int K);                      // groups of 64 / 32: GEMM_ROUTE_AUTO runs dequantize-once (given the scratch); else few-row / tiled kernel
bool gemm_fine_takes_skinny(int M);                                 // groups of 64 / 32: rows up to which the (GP-aware) few-row kernel serves them; beyond: dequantize-once
bool gemm_takes_deq(int M, int N, int K);                           // GEMM_ROUTE_AUTO: dequantize once + fp16 GEMM (given a scratch) beats the fused kernels
hipError_t launch_gemm_f16w(const void* x, const void* w, const void* bias, const void* residual, const void* gate, void* y,
                            int M, int N, int K, int x_stride, int y_stride, hipStream_t st);

// decode-step surroundings (amq_decode.hip)
struct AttnArgs {
    const void* q;        // fp16 [B, n_heads, 128]      (un-rotated)
    const void* k;        // fp16 [B, n_kv_heads, 128]   (un-rotated new key)
    const void* v;        // fp16 [B, n_kv_heads, 128]
    void* kcache;         // fp16 [B, n_kv_heads, max_seq, 128] rotated keys
    void* vcache;         // fp16 [B, n_kv_heads, max_seq, 128]
    void* out;            // fp16 [B, n_heads, 128]
    const int* pos_dev;   // device int32 position of the new token (graph replay), or null -> pos
    int pos;
    int n_heads, n_kv_heads, max_seq;
    float rope_theta;
    const void* rope_table;   // fp16 [max_seq][64][2] (cos, sin) from launch_rope_table, or null -> computed in-kernel
    const void* rope_cur;     // step-state block {fp16 [64][2] (cos, sin) of the CURRENT position; int32 position at byte 256}
                              // maintained by launch_decode_tail, or null
};
hipError_t launch_rope_table(void* tab, int max_seq, float theta, hipStream_t st);
hipError_t launch_rope_table_freqs(void* tab, int max_seq, const void* inv_freq, float scale, hipStream_t st);   // explicit inverse frequencies (rope_scaling)
hipError_t launch_attn_decode(const AttnArgs& a, int batch, hipStream_t st);
#ifndef AMQ_ATT_MIN_CHUNK
#define AMQ_ATT_MIN_CHUNK 256
#endif
constexpr int ATT_MIN_CHUNK = AMQ_ATT_MIN_CHUNK;      // fewest keys a workgroup of the split kernel takes (multiple of 32)
// the same step with the context split over n_splits workgroups per head: ws = fp32 [batch][n_heads][n_splits][132],
// tickets = int32 [batch][n_heads], zero before (and after) every launch
hipError_t launch_attn_decode_split(const AttnArgs& a, int batch, int n_splits, void* ws, void* tickets, hipStream_t st);
// grouped-query models over a long cache (2 <= n_heads / n_kv_heads <= 16): one workgroup per (kv head, chunk) scores the
// chunk against all the group's heads on the matrix cores (amq_attn_prefill.hip: attn_decode_gqa_kernel); same workspace and tickets.
// AMQ_ATT_GQA=0: every query head its own workgroups, as for multi-head models (A/B builds)
#ifndef AMQ_ATT_GQA
#define AMQ_ATT_GQA 1
#endif
bool attn_decode_takes_gqa(int n_heads, int n_kv_heads, int max_seq, int n_splits);
int attn_decode_gqa_iters(int max_seq, int n_splits);      // stages of 128 keys per workgroup
hipError_t launch_attn_decode_gqa(const AttnArgs& a, int batch, int n_splits, void* ws, void* tickets, hipStream_t st);
hipError_t launch_rmsnorm(const void* x, const void* gamma, void* y, int M, int K, float eps, hipStream_t st);
hipError_t launch_rmsnorm_xfrag(const void* x, const void* gamma, void* xf, int M, int K, float eps, hipStream_t st);
// causal attention over a whole prompt (amq_attn_prefill.hip).  Element (b, s, head, d) of q / out sits at
// base + b*bstride + s*rstride + head*128 + d; key / value row t of kv head g at base + b*bstride + t*rstride + g*hstride + d.
struct AttnPrefillArgs {
    const void* q; const void* k; const void* v; void* out;
    int S;                  // query rows per sequence (query s is at position pos0 + s and attends keys 0 .. pos0 + s)
    int pos0;               // keys already in k / v before this prompt chunk
    int n_heads, n_kv_heads, batch;
    long q_rstride, q_bstride, k_rstride, k_bstride, k_hstride, v_rstride, v_bstride, v_hstride, o_rstride, o_bstride;
    int out_xfrag;          // 1: out is the fragment-ordered image of the [S, n_heads * 128] result (batch 1), o_* strides unused
};
hipError_t launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t st);
// prefill glue (amq_decode.hip)
hipError_t launch_rope_cache(void* q, const void* k, const void* v, void* kcache, void* vcache, const void* rope_table,
                             int rope_rows, int pos0, int S, int n_heads, int n_kv_heads, int max_seq, hipStream_t st,
                             int batch = 1);           // q / k / v rows = batch * S; caches [batch][n_kv_heads][max_seq][128]
hipError_t launch_rope_rows(void* q, void* k, const void* rope_table, int rope_rows, int pos0, int rows, int seq_len, int n_heads,
                            int n_kv_heads, hipStream_t st);
hipError_t launch_silu_mul(const void* gate, const void* up, void* out, long n, hipStream_t st);
hipError_t launch_decode_tail(const void* logits, int vocab, const void* embed, int hidden, void* token, void* pos, void* x,
                              const void* rope_table, void* rope_cur, int rope_rows, hipStream_t st, int batch = 1, const void* suppress = nullptr);
hipError_t launch_set_token(const void* token_in, int n_in, const void* embed, int vocab, int hidden, void* token, const void* pos, void* x,
                            const void* rope_table, void* rope_cur, int rope_rows, int batch, hipStream_t st);
hipError_t launch_gemv_f16w(const void* x, const void* W, const void* bias, void* y, const void* gamma, float eps,
                            int N, int K, hipStream_t st, int M = 1);      // x [M, K], y [M, N], M <= 8

// q / k / v GEMV + decode attention in one launch (amq_gemv.hip): a's segments 0 .. 2 = q, k, v (M = 1, RMSNorm prologue, gamma / eps
// set); t: caches, output, step-state block (rope_cur), head counts, max_seq; tickets: int32 [n_heads], zero before and after
hipError_t launch_gemv_qkv_attn(GemvArgs& a, const AttnArgs& t, int* tickets, hipStream_t st);
size_t gemv_qkv_attn_lds_bytes(int K, int max_seq);

// one decode token as one persistent launch (amq_engine.hip)
struct EngineDesc {
    const void* blocks_dev;   // device image of the per-block table (engine_fill_image)
    int n_block;
    int H, I, n_heads, n_kv_heads, max_seq;
    float eps;
    void* x;                  // fp16 [H]: the residual stream, in / out
    void* scratch;            // engine_scratch_bytes(): q, k, v, attention output, gate, up
    const void* state;        // step-state block (cos/sin row of the current position, position, error word)
    void* sync;               // engine_sync_bytes(): barrier words -- zeroed ONCE by the caller (and again when the grid changes or after an error): epochs are monotonic across launches
    int grid;                 // workgroups; 0 = one per CU
    int depth;                // weight-ring slots per wave: 0 = by LDS budget, else 4 or 6
};
struct EngineLinearH { const void* qweight; const void* meta; int N; int bits; int mode; };    // host-side view of one linear
size_t engine_image_bytes(int n_block);
// fills `image` (host memory) with the device table of n_block blocks: lin = [n_block][7] (q, k, v, o, gate, up, down),
// ln / cache pointers per block
void engine_fill_image(void* image, int n_block, const EngineLinearH* lin, const void* const* ln1, const void* const* ln2,
                       void* const* kc, void* const* vc, int H, int I);
size_t engine_sync_bytes();
size_t engine_scratch_bytes(int H, int I, int n_kv_heads);
size_t engine_lds_bytes(const EngineDesc& d, int P);
hipError_t launch_decode_engine(const EngineDesc& d, hipStream_t st);

// reference formats -> native
hipError_t launch_repack(int fmt, int bits, const void* qsrc, const void* s_src, const void* z_src,
                         int N, int K, void* q_native, void* meta_native, hipStream_t st, int group = 128);   // group: the SOURCE format's
hipError_t launch_accumulate_f32(void* mul, const void* y, size_t n, hipStream_t st);
// native -> fp16 W[N,K]
hipError_t launch_dequantize(int bits, int mode, const void* q_native, const void* meta_native,
                             int N, int K, void* w_out, hipStream_t st, int gp = 1);    // gp: meta pairs per (row, tile), 1 / 2 / 4
// HQQ Format A -> fp16 W[N,K] directly (ATEN-style standalone dequant, f-4)
// the optional bfloat16 entry points (amq_bf16.hip; the batched end: amq_gemm_f16.hip's kernel instantiated for bf16 operands)
hipError_t launch_dequantize_bf16(int bits, const void* q_native, const void* meta_native_bf16, int N, int K, void* w_bf16, hipStream_t st);
hipError_t launch_dequantize_hqq_bf16(int bits, const void* wq, const void* scale_bf16, const void* zero_bf16, int N, int K, void* w_bf16, hipStream_t st, int gs);
hipError_t launch_gemv_bf16(int bits, const void* x, const void* q_native, const void* meta_native_bf16, const void* bias, const void* residual,
                            void* y, int M, int N, int K, int x_stride, int y_stride, hipStream_t st);
hipError_t launch_gemm_bf16w(const void* x, const void* w, const void* bias, const void* residual, void* y,
                             int M, int N, int K, int x_stride, int y_stride, hipStream_t st);
hipError_t launch_dequantize_hqq(int bits, const void* wq, const void* scale, const void* zero,
                                 int N, int K, void* w_out, hipStream_t st, int group = 128);

}  // namespace amq
