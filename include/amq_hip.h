/* amq_hip.h -- C ABI of libamq_hip.so: the MI355X (gfx950) replacement for the
 * native kernels on AMQ's mixed-precision inference hot path.
 *
 * What it replaces (reference paths relative to the dlwns147/amq checkout):
 *   pybind module `auto_gptq`          amq/kernel/AutoGPTQ/auto_gptq_kernel.cu:471-475
 *       vecquant{2,3,4}matmul_faster_old(vec, mat, mul, scales, zeros, groupsize, vec_height)
 *   pybind module `faster_transformer` amq/kernel/ft/FT.cpp:9-19
 *       gemv_4bit(x, kernel, scales, scaled_zeros, m, n, k, group_size)   gemv_cuda.cu:358-525
 *       gemm_4bit(x, kernel, scales, scaled_zeros)                        gemm_cuda.cu:929-1033
 *   host packers  GPTQLinear.pack (hqq/backends/autogptq.py:111-156),
 *                 pack_intweight  (hqq/backends/ft.py:15-55)
 *   dequantize    Quantizer.dequantize (hqq/core/quantize.py:184-199)
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t
 *     passed as void* (NULL = default stream).  All work is stream-ordered and
 *     asynchronous; nothing allocates, synchronises or touches global state,
 *     so every call may be captured into a hipGraph.
 *   - every pointer is a DEVICE pointer owned by the caller unless noted.
 *   - return value: AMQ_OK (0) or a negative AMQ_E* code; amq_last_error()
 *     returns a thread-local message for the most recent failure.
 *   - fp16 activations / outputs, fp32 accumulation; weights 2, 3 or 4 bit,
 *     group size 128 along K (what AMQ produces: amq/amq_quantization_proxy.py:22,36) or a multiple of 128 that divides K: the
 *     `group` argument of the amq_repack_from_* / amq_dequantize_hqq_f16 calls is the SOURCE format's group size (HQQ's packing
 *     geometry depends on it); each group's (scale, zero) is replicated into the native layout's per-128 pairs, so the compute
 *     entry points are the same kernels for every such group size.  Groups of 64 and 32 (HQQ's default group_size is 64,
 *     hqq/core/quantize.py:1078; the reference's GPTQ kernels take any groupsize, auto_gptq_kernel.cu:203) keep 128 / group pairs per
 *     (row, 128-column tile) in the native meta (amq_native_meta_bytes grows accordingly) and are served by: the amq_repack_from_* and
 *     amq_dequantize_* calls, the GEMV kernel (amq_gemv_f16 / amq_gemv_grouped_f16: <= 16 rows, every prologue, default options), the few-row
 *     GEMM up to 256 rows and the tiled GEMM beyond (amq_gemm_f16; AMQ_GEMM_AUTO / _SKINNY / _TILED of the route
 *     calls) and, for launches that fill 256 x 256 tiles, the dequantize-once GEMM route (amq_gemm_route_f16 / amq_gemm_res_f16 /
 *     amq_gemm_gated_f16 with the amq_gemm_route_workspace_bytes_g workspace); the ring / wave-specialised routes and the other compute
 *     entry points return an error for them.  The compute calls' `group` argument is therefore the
 *     NATIVE buffers' granularity: 64, 32, or anything >= 128.  N % 16 == 0, K % 128 == 0.
 *   - "native" buffers are in the AMQ-T16 layout (amq_common.cuh);
 *     sizes from amq_native_*_bytes(); produced by the amq_repack_from_* calls.
 */
#ifndef AMQ_HIP_H
#define AMQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMQ_VERSION 521            /* 0.5.2: amq_rope_table_freqs_f16 (rope_scaling) and amq_decode_tail_suppress_f16 added, nothing else changed.  0.5.1: the bfloat16 entry points
                                    * (amq_*_bf16) added.  0.5.0: amq_gemv_opts.math renumbered
                                    * (0 = the build's default), amq_default_gemv_math added; the decode-engine and fused q/k/v-attention entry points live in
                                    * libamq_hip_ab.so (include/amq_hip_ab.h) since 0.4 */

#define AMQ_OK            0
#define AMQ_EINVAL       -1        /* bad argument (null pointer, bits, mode ...) */
#define AMQ_ESHAPE       -2        /* unsupported shape (N % 16, K % 128, group not 32 / 64 / a multiple of 128 dividing K, M out of range) */
#define AMQ_ELAUNCH      -3        /* HIP launch failed (message carries hipGetErrorString) */
#define AMQ_EUNSUPPORTED -4        /* valid request this build does not implement */

/* dequantisation arithmetic carried by a native buffer's meta */
#define AMQ_MODE_HQQ 0             /* meta = (scale, zero):  w = fp16(fp16(q - zero) * scale)   quantize.py:198 */
#define AMQ_MODE_FMA 1             /* meta = (scale, c):     w = fp16(fma(q, scale, c))          auto_gptq_kernel.cu:206, gemv_cuda.cu:151 */
#define AMQ_MODE_FMA1 2            /* AMQ_MODE_FMA for a buffer whose scales satisfy |scale| <= amq_fma1_scale_bound(bits): the GEMV kernel then unpacks a
                                    * weight pair with one packed fma instead of two ops -- the same weights, bit for bit (the caller vouches for the bound:
                                    * larger scales overflow the pre-scaled multiplier); every other kernel treats it as AMQ_MODE_FMA */

/* fused x transforms of amq_gemv_grouped_f16 */
#define AMQ_PRO_NONE     0
#define AMQ_PRO_RMSNORM  1         /* x <- gamma * fp16(x * rsqrt(mean(x^2) + eps))   (LlamaRMSNorm; FT generalT5LayerNorm, layernorm.cu:25-51) */
#define AMQ_PRO_SILU_MUL 2         /* x <- fp16(silu(x)) * x2                           (LlamaMLP act_fn(gate) * up) */

#define AMQ_MAX_SEGMENTS 4

int amq_version(void);
const char* amq_last_error(void);

/* GEMV arithmetic (MODE_HQQ buffers, whose reference dequant has two fp16 roundings per weight: quantize.py:198).
 *   EXACT      both roundings per weight, bit-identical weights to Quantizer.dequantize; x*w accumulated in fp32.
 *   GROUPSCALE (opt-in) the first rounding exactly as EXACT, d = fp16((q - z) * 2^-9); the scale is applied once per (row, 128-group) in fp32
 *              after the tile's MFMAs: y += (s * 2^9) * sum_k x_k d_k.  One fp16 rounding per weight is not taken (<= 2^-11 relative per weight):
 *              measured 3.2e-4 of rms(y) rms from the reference result, worst element 0.93 of the parity bar |dy| <= 1e-3 |y| + 1e-3 rms(y), and
 *              2 fp16 ulps -- past the bar -- on a few elements per 10^4 when a bias add rounds a second time (profiles/r05_gemv_groupscale.txt):
 *              not the default for that reason.  6.5 instead of 12 VALU cycles per weight pair: +5 % decode tokens/s on Llama-2-7B, +13 % on 70B.
 *              The reference's own CUDA kernels round once per weight too (auto_gptq_kernel.cu:206-218).  Buffers in AMQ_MODE_FMA / _FMA1 (one
 *              rounding by definition) and groups of 64 / 32 run their exact forms under it.
 *   LINEAR     no per-weight roundings at all: y = sum_g s_g * (sum_k x_k q_k - z_g sum_k x_k) in fp32 -- the real-valued dequant (opt-in;
 *              ~3.6e-4 of rms(y) rms, beyond the parity bar at its worst elements).
 *   DEFAULT    what amq_default_gemv_math() returns for this build of the library. */
#define AMQ_MATH_DEFAULT    0
#define AMQ_MATH_LINEAR     1
#define AMQ_MATH_GROUPSCALE 2
#define AMQ_MATH_EXACT      3
int amq_default_gemv_math(void);   /* AMQ_MATH_EXACT or AMQ_MATH_GROUPSCALE */

/* Per-call launch options of amq_gemv_grouped_f16 (host struct; NULL or all-zero = defaults).  There is no
 * process-wide option state in the library: what a call computes depends on its arguments only. */
typedef struct amq_gemv_opts {
    int math;    /* AMQ_MATH_DEFAULT (0), _EXACT, _GROUPSCALE or _LINEAR */
    int waves;   /* A/B: waves per workgroup, 0 = auto, 4, 8 or 16 */
    int depth;   /* A/B: tile loads in flight per wave, 0 = auto, 2 or 4 */
    int rpt;     /* A/B: row-tiles walked by one workgroup, 0 = auto, 1..64 */
    int dot;     /* A/B: 1 = M == 1 runs the v_dot2c + wavefront-shuffle body instead of the MFMA body */
} amq_gemv_opts;

/* kernel family of amq_gemm_route_f16 (AUTO = what every other GEMM entry point uses: chosen by shape) */
#define AMQ_GEMM_AUTO   0
#define AMQ_GEMM_TILED  1          /* LDS-DMA double-buffered 64/128-row tiles (+ split-K for few rows) */
#define AMQ_GEMM_SKINNY 2          /* barrier-free K-split-by-wave kernel, M <= 64 */
#define AMQ_GEMM_RING   3          /* 256 x 256 tiles (128 x 256 when those would not fill the chip), LDS rings for x and packed W, counted waits */
#define AMQ_GEMM_RING128 4         /* the same kernel forced to 128-row tiles */
#define AMQ_GEMM_WS 5              /* amq_gemm_ws.hip: 256 x 128 tiles, 4 MFMA waves + 4 DMA / unpack waves per workgroup (AUTO takes it where that tile fills the chip better) */
#define AMQ_GEMM_DEQ 6             /* dequantize once into the caller's workspace (N * K * 2 bytes), then amq_gemm_f16.hip: a plain fp16 GEMM with no
                                      unpack in its loop -- GPTQLinear.forward's own split from 128 rows on (hqq/backends/autogptq.py:245-283), hand-written.
                                      AUTO takes it for MFMA-bound launches when the workspace is passed. */

/* capabilities: writes up to `cap` ints {max GEMV rows for K (any options / group size), LDS limit, tile rows, tile columns, max GEMV rows for K
 * with default options over groups of 128, the same without an RMSNorm prologue (x may then be staged in two K phases)}; returns the count */
int amq_query(int K, int* out, int cap);

/* ---- native buffer sizes ------------------------------------------------ */
size_t amq_native_qweight_bytes(int bits, int N, int K);
float amq_fma1_scale_bound(int bits);   /* largest |scale| of an AMQ_MODE_FMA1 buffer: 65504 / 2^18 (4 bit), 65504 / 2^20 (2, 3 bit) */
size_t amq_native_meta_bytes(int N, int K, int group);

/* ---- load-time repack: reference formats -> native ---------------------- */
/* Format A: HQQLinear.W_q (uint8 for 4/2 bit, int32 for 3 bit) + meta['scale'], meta['zero'] fp16 [N*K/group]
 * (hqq/core/bitpack.py:24-110, quantize.py:106-111).  Result carries AMQ_MODE_HQQ. */
int amq_repack_from_hqq(int bits, const void* W_q, const void* scale, const void* zero,
                        int N, int K, int group, void* qweight_native, void* meta_native, void* stream);
/* Format B: GPTQLinear buffers qweight int32 [K/32*bits, N], scales fp32 [K/group, N], zeros fp32 [K/group, N]
 * (hqq/backends/autogptq.py:55-75).  Result carries AMQ_MODE_FMA (c = -zeros). */
int amq_repack_from_gptq(int bits, const void* qweight, const void* scales, const void* zeros,
                         int N, int K, int group, void* qweight_native, void* meta_native, void* stream);
/* Format C: FT_QuantLinear buffers qweight int16 [N/4, K], scales fp16 [K/group, N], scaled_zeros fp16
 * (hqq/backends/ft.py:57-126), 4 bit only.  Result carries AMQ_MODE_FMA (c = scaled_zeros). */
int amq_repack_from_awq(const void* qweight, const void* scales, const void* scaled_zeros,
                        int N, int K, int group, void* qweight_native, void* meta_native, void* stream);

/* ---- dequantize ---------------------------------------------------------- */
/* native -> W[N,K] fp16 (row-major, nn.Linear orientation) */
int amq_dequantize_f16(int bits, int mode, const void* qweight_native, const void* meta_native,
                       int N, int K, int group, void* W_out, void* stream);
/* Format A -> W[N,K] fp16: Quantizer.dequantize (quantize.py:184-199) as one kernel */
int amq_dequantize_hqq_f16(int bits, const void* W_q, const void* scale, const void* zero,
                           int N, int K, int group, void* W_out, void* stream);

/* ---- the hot path: y[M,N] = x[M,K] . W^T (+ bias) ------------------------ */
/* few rows (decode).  The unpacked weights are the MFMA B operand straight from registers (W never touches LDS); the x rows are staged in LDS.
 * M <= 16 and the rows must fit LDS: M * (K + 8) * 2 bytes + the cross-wave sum buffer (4 - 32 KiB by row count) <= 160 KiB -- except that with
 * default options, no RMSNorm prologue and 5 .. 8 rows a K whose rows do not fit whole is staged in two K phases (8 rows of K = 11008).
 * amq_query(K) returns the row limits; AMQ_ESHAPE beyond -- use amq_gemm_f16.  x_stride / y_stride in elements (0 = dense; 2 .. 8 dense rows
 * take the LDS-DMA staging path, strided rows the generic one: same results). */
int amq_gemv_f16(int bits, int mode, const void* x, const void* qweight_native, const void* meta_native,
                 const void* bias, void* y, int M, int N, int K, int group,
                 int x_stride, int y_stride, void* stream);
/* any M (prefill / batched): LDS-staged x tiles, MFMA 16x16x32 f16 */
int amq_gemm_f16(int bits, int mode, const void* x, const void* qweight_native, const void* meta_native,
                 const void* bias, void* y, int M, int N, int K, int group,
                 int x_stride, int y_stride, void* stream);
/* dispatch the way the reference modules do (rows < threshold -> gemv, else gemm;
 * GPTQLinear.forward autogptq.py:163, FT_QuantLinear.forward_normal ft.py:128-145) */
int amq_linear_f16(int bits, int mode, const void* x, const void* qweight_native, const void* meta_native,
                   const void* bias, void* y, int M, int N, int K, int group, void* stream);

/* Few-row GEMM (16 < M <= ~256: short prompts) with split-K: a launch whose M x N tiles would occupy fewer than 192
 * workgroups splits the K loop over up to 8 workgroup layers; each layer writes fp32 partials into its own slice of a
 * caller-owned workspace, a second kernel sums the slices in a fixed order (deterministic, no atomics).
 * amq_gemm_splitk_workspace_bytes() returns the bytes required (0: the shape does not split and no workspace is
 * needed; the call then forwards to amq_gemm_f16).  Replaces the reference's cross-CTA split-K of gemm_w4a16_T1
 * (amq/kernel/ft/quantization_new/gemm/gemm_cuda.cu:514-584: semaphore + __hadd2 read-modify-write of C). */
size_t amq_gemm_splitk_workspace_bytes(int M, int N, int K);
int amq_gemm_splitk_f16(int bits, int mode, const void* x, const void* qweight_native, const void* meta_native,
                        const void* bias, void* y, int M, int N, int K, int group, int x_stride, int y_stride,
                        void* workspace, size_t workspace_bytes, void* stream);

/* several linears that consume the same x (q/k/v, gate/up), each with its own bit-width,
 * in ONE launch; optional fused prologue on x and residual add on y. */
typedef struct amq_segment {
    const void* qweight_native;
    const void* meta_native;
    const void* bias;       /* fp16 [N] or NULL */
    const void* residual;   /* fp16 [M, y_stride] or NULL: y = residual + (x W^T + bias) */
    void* y;                /* fp16 [M, y_stride] */
    int N;
    int bits;
    int mode;
    int y_stride;           /* 0 = N */
} amq_segment;

int amq_gemv_grouped_f16(const amq_segment* segments /* host */, int nseg,
                         const void* x, const void* x2, const void* gamma, float eps, int prologue,
                         int M, int K, int group, int x_stride, const amq_gemv_opts* opts /* host, may be NULL */,
                         void* stream);
/* 2 .. 8 rows (sequences decoded together), RMSNorm WITHOUT a pass over x for its statistic: a launch that writes a hidden state (o_proj, down_proj:
 * ONE segment, residual in the epilogue) also leaves, per row, one sum of squares per 16 output columns in sums_out (fp32 [M][N / 16], written
 * whole by every launch); the launch that normalises that hidden state takes them as sums_in (fp32 [M][K / 16]) with gamma and eps, adds them
 * in a fixed order and only applies gamma * fp16(x * rstd) while staging x -- LlamaRMSNorm's value up to the summation order of its fp32
 * mean (the fused AMQ_PRO_RMSNORM prologue repeats the whole statistic in every workgroup -- ~90 VALU instructions per row and thread: at 5 .. 8
 * rows that costs more than a separate amq_rmsnorm_f16 launch; this form costs less than either, from 2 rows on).  sums_in == NULL: no prologue (then gamma must be NULL too); sums_out == NULL: none
 * written.  x dense [M][K], groups of 128, default arithmetic; K <= 8192 with sums_in.  Replaces layernorm.cu:25-51 + the GEMV behind it in the
 * reference's FT step (ftllama_modeling.py:39-46) for Batch 2 .. 8 (gemv_cuda.cu:381-437 makes batches first-class). */
int amq_gemv_grouped_sums_f16(const amq_segment* segs, int nseg, const void* x, const void* gamma, float eps, const float* sums_in,
                              float* sums_out, int M, int K, int group, void* stream);


/* ---- entry points shaped like the reference's pybind functions ------------------------------------
 * They take the reference's own buffers (Format B / Format C) unchanged.  The HIP kernels read the native layout,
 * so each call needs a caller-owned `workspace` of at least amq_compat_workspace_bytes(bits, M, N, K) bytes; the
 * native copy of the weights is (re)built into it unless `workspace_valid` != 0 (pass 0 on the first call for a given
 * weight, 1 afterwards: the repack is then skipped and the call costs the same as amq_linear_f16).
 *
 * amq_vecquantmatmul_faster_old  <->  auto_gptq.vecquant{2,3,4}matmul_faster_old(vec, mat, mul, scales, zeros,
 *     groupsize, vec_height)                (auto_gptq_kernel.cu:129-158, 227-256, 345-374, 443-467)
 *     vec fp16 [batch, K], mat int32 [K/32*bits, N], mul fp32 [batch, N] ACCUMULATED IN PLACE (mul += vec . W),
 *     scales / zeros fp32 [K/groupsize, N]; height = mat rows, width = N; vec_height (= K/2) is accepted and checked.
 * amq_gemv_4bit  <->  faster_transformer.gemv_4bit(x, kernel, scales, scaled_zeros, m, n, k, group_size) -> y
 *     (gemv_cuda.cu:358-437); the reference throws for m outside 1..7, here any m >= 1 works.
 * amq_gemm_4bit  <->  faster_transformer.gemm_4bit(x, kernel, scales, scaled_zeros) -> y   (gemm_cuda.cu:929-1033)
 *     x fp16 [m, k], kernel int16 [n/4, k], scales / scaled_zeros fp16 [k/group, n], y fp16 [m, n] (caller-allocated). */
size_t amq_compat_workspace_bytes(int bits, int M, int N, int K);
int amq_vecquantmatmul_faster_old(int bits, const void* vec, const void* mat, void* mul, const void* scales,
                                  const void* zeros, int groupsize, int vec_height, int batch, int height, int width,
                                  void* workspace, size_t workspace_bytes, int workspace_valid, void* stream);
int amq_gemv_4bit(const void* x, const void* kernel, const void* scales, const void* scaled_zeros, void* y,
                  int m, int n, int k, int group_size,
                  void* workspace, size_t workspace_bytes, int workspace_valid, void* stream);
int amq_gemm_4bit(const void* x, const void* kernel, const void* scales, const void* scaled_zeros, void* y,
                  int m, int n, int k, int group_size,
                  void* workspace, size_t workspace_bytes, int workspace_valid, void* stream);

/* ---- decode-step surroundings (next tier: what sits between the linears in one token step) ---- */
/* y = gamma * fp16(x * rsqrt(mean(x^2) + eps)), rows of K; replaces FT layernorm_forward_cuda (ft/layernorm/layernorm.cu:25-77) */
int amq_rmsnorm_f16(const void* x, const void* gamma, void* y, int M, int K, float eps, void* stream);
/* y[N] = (optionally RMSNorm'ed) x[K] . W^T for fp16 W[N,K] (lm_head), K % 8 == 0; gamma NULL = no norm */
int amq_gemv_f16w(const void* x, const void* W, const void* bias, void* y, const void* gamma, float eps,
                  int N, int K, void* stream);
/* the same for M = 1 .. 8 rows (batched decode): x fp16 [M, K], y fp16 [M, N], both contiguous; W is streamed once */
int amq_gemv_f16w_rows(const void* x, const void* W, const void* bias, void* y, const void* gamma, float eps, int M,
                       int N, int K, void* stream);
/* one new token per sequence: RoPE(q, k) at position *pos_dev (or pos if pos_dev is NULL), append k/v to the
 * cache [B, n_kv_heads, max_seq, 128], out = softmax(q K^T / sqrt(128)) V.  head_dim must be 128.
 * Replaces FT single_query_attention (ft/attention/decoder_masked_multihead_attention.cu:30-61) with HF-Llama numerics. */
int amq_attn_decode_f16(const void* q, const void* k, const void* v, void* kcache, void* vcache, void* out,
                        const int* pos_dev, int pos, int batch, int n_heads, int n_kv_heads, int head_dim,
                        int max_seq, float rope_theta, const void* rope_table, void* stream);
/* optional: fp16 [max_seq][64][2] (cos, sin) table for amq_attn_decode_f16 (NULL there = computed in-kernel, same values) */
int amq_rope_table_f16(void* table, int max_seq, float rope_theta, void* stream);
/* the same table from explicit inverse frequencies: inv_freq fp32 [64] on the device -- what HF's rotary embedding holds after a static
 * rope_scaling (Llama-3.1's "llama3" factors, "linear", ...; amq/configs/llama.json:82+ lists the Llama-3.x models) -- and its attention_scaling
 * factor (1 for those): table[pos][i] = (fp16(cos(pos * inv_freq[i]) * scale), fp16(sin(...) * scale)), LlamaRotaryEmbedding.forward's expression.
 * Every kernel that rotates (amq_attn_decode_*_f16 through the step-state row, amq_rope_cache_*, amq_rope_rows_f16) reads the table it is given. */
int amq_rope_table_freqs_f16(void* table, int max_seq, const float* inv_freq, float scale, void* stream);

/* Decode attention for graph-replayed token steps.  step_state is a 264-byte device block
 *     { fp16 cos/sin [64][2] of the CURRENT position ; int32 position at byte 256 ; int32 sticky error word at byte 260 }
 * that amq_decode_tail_f16 keeps up to date (pass rope_cur = step_state, pos = step_state + 256 there): position and
 * rotation inputs are fetched by the kernel's first instructions instead of through the position -> table-row chain of
 * dependent loads of amq_attn_decode_f16.
 * Both attention entry points treat a device-side position outside 0 .. max_seq-1 as a no-op (no cache append, no
 * output); this one also sets the error word to 1 (it is never cleared by the library: zero it when the state is
 * created, read it back to learn that a replay ran past the cache).  amq_decode_tail_f16 saturates the position at
 * rope_rows (= max_seq), so a runaway replay keeps hitting the no-op path. */
int amq_attn_decode_cur_f16(const void* q, const void* k, const void* v, void* kcache, void* vcache, void* out,
                            const void* step_state, int batch, int n_heads, int n_kv_heads, int head_dim, int max_seq,
                            void* stream);

/* The same attention step for LONG contexts: n_splits workgroups share a head (grid = heads x batch x n_splits); the context
 * 0 .. pos is cut into at most n_splits chunks of >= 256 keys, each workgroup computes its chunk's scores, softmax statistics
 * and un-normalised output, leaves them in `workspace`, takes a ticket, and the last one to arrive (nobody waits) combines
 * them in chunk order.  amq_attn_decode_f16 walks a head's whole context on one CU -- right for a few hundred keys, 4.2 ms of
 * a 5.4 ms 7B token at 4000.  While the context fits ONE chunk (pos < 256, or n_splits == 1) the result is that of
 * amq_attn_decode_f16 / _cur_f16 bit for bit; with several chunks the probabilities are not rounded to fp16 before P.V
 * (exact softmax; within output rounding of the one-chunk result).  n_splits = ceil(max_seq / 384) keeps every chunk on the
 * kernel's register-prefetch path.  step_state != NULL: position / rotation from the step-state block (as _cur_f16; pos_dev,
 * pos, rope_theta, rope_table ignored); else as amq_attn_decode_f16.  workspace: amq_attn_decode_split_workspace_bytes bytes,
 * no initialisation; tickets: int32 [batch * n_heads], ZERO before the first launch, left zero by every launch; neither may
 * be shared by launches that can run concurrently.  The out-of-range position guard is the same (no-op, error word).
 * Grouped-query models (2 .. 16 query heads per kv head, n_splits > 1): ONE workgroup per (kv head, chunk) takes the chunk in once and scores it
 * against all the group's heads on the matrix cores, chunks of 128 * ceil(ceil(max_seq / n_splits) / 128) keys walked in double-buffered stages,
 * and a second small launch adds the chunks' partial results in chunk order (same workspace; the tickets are not used).  Its arithmetic is the
 * prompt kernel's (amq_attn_prefill_f16: fp32 scores, un-normalised probabilities rounded to fp16 before P.V) -- within fp16 output rounding
 * of the per-head kernels, also with one active chunk; deterministic.  Llama-3.1-8B heads at 8192 / 32768 cached keys: 23.6 -> 15.3 us,
 * 61 -> 33 us per launch (4.1 TB/s of K + V). */
size_t amq_attn_decode_split_workspace_bytes(int batch, int n_heads, int n_splits);
int amq_attn_decode_split_f16(const void* q, const void* k, const void* v, void* kcache, void* vcache, void* out,
                              const void* step_state, const int* pos_dev, int pos, int batch, int n_heads, int n_kv_heads,
                              int head_dim, int max_seq, float rope_theta, const void* rope_table, int n_splits,
                              void* workspace, size_t workspace_bytes, void* tickets, void* stream);

/* End of a greedy token step in one launch: token[0] = argmax(logits[0..vocab)) (first maximum), pos[0] += 1,
 * x[0..hidden) = embed[token][0..hidden); when rope_table / rope_cur are given (both or neither) also
 * rope_cur[0..128) = rope_table[min(pos, rope_rows - 1)][0..128), the cos/sin row amq_attn_decode_cur_f16 reads in the next step.  Replaces the `torch.argmax` / position increment / embedding gather that follow
 * the lm_head in the reference's generation loop (amq/utils/speed.py:70-76, HF `_sample`) when the step is replayed from
 * a hipGraph.  logits, embed, x: fp16; token: int64; pos: int32; all device pointers. */
int amq_decode_tail_f16(const void* logits, int vocab, const void* embed, int hidden, long long* token, int* pos, void* x,
                        const void* rope_table, void* rope_cur, int rope_rows, void* stream);
/* Batched decode (several sequences at the SAME position, one step state): logits fp16 [batch, vocab], token int64 [batch],
 * x fp16 [batch, hidden]; pos / rope_cur are advanced once. */
/* The same with up to 8 token ids that are never chosen: suppress_ids = device int32 [8], -1 = unused slot (the values may change between replays
 * of a captured step; the pointer may not).  What HF's MinNewTokensLengthLogitsProcessor does to the EOS ids while min_new_tokens has not been
 * reached -- generate(min_new_tokens = max_new_tokens = n), amq/utils/speed.py:31-36 -- i.e. on every step of the reference's TPS loop. */
/* "This is the next input token" for a captured token step whose caller feeds the tokens itself (model(ids, start_pos=...) token by token,
 * amq/utils/speed.py:76-90): token[b] = token_in[b] (n_in == batch) or token_in[0] (n_in == 1), clamped to the vocabulary; x[b] = embed[token[b]];
 * rope_cur = rope_table[min(*pos, rope_rows - 1)] (both or neither, as above).  The position is not changed.  All pointers device. */
int amq_set_token_f16(const long long* token_in, int n_in, const void* embed, int vocab, int hidden, long long* token, const int* pos, void* x,
                      const void* rope_table, void* rope_cur, int rope_rows, int batch, void* stream);
int amq_decode_tail_suppress_f16(const void* logits, int vocab, const void* embed, int hidden, long long* token, int* pos, void* x,
                                 const void* rope_table, void* rope_cur, int rope_rows, int batch, const int* suppress_ids, void* stream);
int amq_decode_tail_batch_f16(const void* logits, int vocab, const void* embed, int hidden, long long* token, int* pos, void* x,
                              const void* rope_table, void* rope_cur, int rope_rows, int batch, void* stream);

/* ---- many-row (prefill) glue --------------------------------------------------------------------------
 * The reference runs these steps as framework ops between the linears of a HF Llama block
 * (transformers LlamaAttention / LlamaMLP / LlamaDecoderLayer as driven by amq/utils/speed.py:150-200); on this path
 * they are ~20 tiny launches per block, so they are offered fused. */

/* amq_gemm_splitk_f16 plus a fused residual: y = residual + fp16(x . W^T (+ bias)); residual fp16 [M, y_stride] or
 * NULL, may alias y.  workspace NULL = single pass (no split-K). */
int amq_gemm_res_f16(int bits, int mode, const void* x, const void* qweight_native, const void* meta_native,
                     const void* bias, const void* residual, void* y, int M, int N, int K, int group, int x_stride,
                     int y_stride, void* workspace, size_t workspace_bytes, void* stream);
/* amq_gemm_res_f16 (dense y) followed by amq_rmsnorm_xfrag_f16 of its result rows -- y = residual + fp16(x . W^T (+ bias)), xf = the fragment-ordered
 * gamma * fp16(y * rsqrt(mean(y^2) + eps)) -- as ONE call: where the GEMM runs split-K (few rows, a workspace given) the sum over the splits and the
 * norm are one launch, otherwise the norm follows as its own.  Same bits as the two calls.  N % 128 == 0; xf: amq_xfrag_bytes(M, N) bytes.
 * The hand-over between a decoder block's down_proj and the next block's input_layernorm on a short prompt pass (the reference runs both as
 * separate framework ops: transformers LlamaDecoderLayer as driven by amq/utils/speed.py:150-200).  (ABI 521) */
int amq_gemm_res_norm_xfrag_f16(int bits, int mode, const void* x, const void* qweight_native, const void* meta_native, const void* bias,
                                const void* residual, void* y, int M, int N, int K, int group, int x_stride, void* workspace,
                                size_t workspace_bytes, const void* gamma, float eps, void* xf, void* stream);
/* The same with the kernel family chosen by the caller (tests, A/B tools); amq_gemm_route_workspace_bytes is the
 * matching workspace query (0: no workspace needed).  The workspace holds split-K partials (few rows) or the dequantized
 * fp16 weights (AMQ_GEMM_DEQ, and AUTO on MFMA-bound launches) -- never both; without it AUTO runs a fused kernel. */
size_t amq_gemm_route_workspace_bytes(int route, int M, int N, int K);
/* ... for a given group size: groups of 64 / 32 take the few-row kernel up to 256 rows (0 bytes), the tiled kernel beyond (split-K partials as for
 * group 128) and the dequantize-once route (N * K * 2 bytes) for launches that fill 256 x 256 tiles -- the ring / wave-specialised kernels read one
 * (scale, zero) pair per 128 columns. */
size_t amq_gemm_route_workspace_bytes_g(int route, int M, int N, int K, int group);
int amq_gemm_route_f16(int route, int bits, int mode, const void* x, const void* qweight_native, const void* meta_native,
                       const void* bias, const void* residual, void* y, int M, int N, int K, int group, int x_stride,
                       int y_stride, void* workspace, size_t workspace_bytes, void* stream);
/* y = x . W^T (+ bias) (+ residual | silu-gated) with W as DENSE fp16 [N, K] (row-major): the matmul of GPTQLinear.forward's
 * many-row branch (hqq/backends/autogptq.py:283: torch.matmul(x, weights)) as a hand-written kernel -- what AMQ_GEMM_DEQ runs
 * behind amq_dequantize_f16, exported for callers that keep fp16 weights (lm_head, an fp16 base model).  N % 16 == 0,
 * K % 128 == 0, x_stride % 8 == 0, y_stride % 4 == 0 (0 = dense); epilogue operands as amq_gemm_res_f16 / amq_gemm_gated_f16. */
int amq_gemm_f16w_f16(const void* x, const void* w_f16, const void* bias, const void* residual, const void* gate, void* y,
                      int M, int N, int K, int x_stride, int y_stride, void* stream);
/* The LlamaMLP product act_fn(gate_proj(x)) * up_proj(x) with up_proj as this GEMM:
 *     y = fp16(silu(gate)) * fp16(x . W^T (+ bias)),   gate fp16 [M, N] contiguous, y [M, N] contiguous,
 * formed in the epilogue of the kernel that serves the shape (256-row ring and few-row kernels) or by the element-wise
 * launch behind it (tiled kernel) -- the same expression as amq_silu_mul_f16 on the separate outputs, bit for bit.
 * gate may alias y only in the first case: amq_gemm_gated_fused(route, M, N, K, use_workspace) says which it is
 * (use_workspace: whether a split-K workspace will be passed).  route / workspace as amq_gemm_route_f16. */
int amq_gemm_gated_fused(int route, int M, int N, int K, int use_workspace);
int amq_gemm_gated_fused_g(int route, int M, int N, int K, int use_workspace, int group);   /* ... for a given group size (64 / 32: few-row kernel and dequantize-once yes, tiled no) */
int amq_gemm_gated_f16(int route, int bits, int mode, const void* x, const void* qweight_native, const void* meta_native,
                       const void* bias, const void* gate, void* y, int M, int N, int K, int group, int x_stride,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Fragment-ordered activations for few-row GEMMs.  For a few dozen rows the GEMM is bound by how fast a CU can pull x
 * in MFMA operand order; amq_xfrag_f16 lays x out so that every wave-load is one contiguous KiB that already IS an
 * operand:  xf[g][kt][mb*4 + t][lane = 16*o + r][8] = x[g*64 + mb*16 + r][kt*128 + 32*t + 8*o .. +8]  (rows >= M zero),
 * amq_xfrag_bytes(M, K) bytes.  Source element (m, k) is read at src[m*stride_m + (k/128)*stride_kt + k%128] (halves):
 * row-major [M, K]: stride_m = K (or more), stride_kt = 128; an attention output [heads, M, 128]: 128, M*128.
 * amq_gemm_xfrag_f16 = amq_gemm_res_f16 reading such a buffer (any M; no workspace), plus an optional SiLU gate. */
size_t amq_xfrag_bytes(int M, int K);
int amq_xfrag_f16(const void* src, void* xf, int M, int K, long long stride_m, long long stride_kt, void* stream);
/* amq_rmsnorm_f16 (rows of K, contiguous) writing its result in fragment order */
int amq_rmsnorm_xfrag_f16(const void* x, const void* gamma, void* xf, int M, int K, float eps, void* stream);
/* gate (fp16 [M, y_stride] or NULL, may alias y): y = fp16(silu(gate)) * fp16(x . W^T (+ bias)) -- the LlamaMLP product
 * act_fn(gate_proj(x)) * up_proj(x) formed in up_proj's epilogue (same expression as amq_silu_mul_f16) */
int amq_gemm_xfrag_f16(int bits, int mode, const void* xf, const void* qweight_native, const void* meta_native,
                       const void* bias, const void* gate, const void* residual, void* y, int M, int N, int K, int group,
                       int y_stride, void* stream);
/* Several linears over the SAME fragment-ordered x (q/k/v, gate/up of a prompt pass), each with its own bit-width, as
 * segments of one few-row launch (amq_segment as for amq_gemv_grouped_f16; residual optional per segment):
 * y_i = x . W_i^T (+ bias_i) (+ residual_i).  Results are those of amq_gemm_xfrag_f16 per segment, bit for bit. */
int amq_gemm_xfrag_grouped_f16(const amq_segment* segments /* host */, int nseg, const void* xf, int M, int K, int group,
                               void* stream);
/* The same launch with its kernel form chosen by the caller (tests, A/B tools; the results do not depend on it, bit for bit):
 * AMQ_FEWROW_AUTO = what amq_gemm_xfrag_grouped_f16 runs (the streaming form where it saves a round of the chip), AMQ_FEWROW_TILE = x fragments a
 * whole K tile at a time (up to four 16-column blocks per workgroup), AMQ_FEWROW_STREAM = one MFMA step at a time through a register ring
 * (amq_gemm_fewrow.hip; blocks_per_wg 1, 2, 3, 4 or 6; 0 = by the launch's size). */
#define AMQ_FEWROW_AUTO   0
#define AMQ_FEWROW_TILE   1
#define AMQ_FEWROW_STREAM 2
int amq_gemm_xfrag_grouped_form_f16(const amq_segment* segments /* host */, int nseg, const void* xf, int M, int K, int group,
                                    int form, int blocks_per_wg, void* stream);
/* Causal self-attention over a prompt: for every sequence b < batch, query row s < S (position pos0 + s) and head h,
 *     out[b, s, h, :] = softmax(q[b, s, h, :] . K[b, 0 .. pos0 + s, g, :]^T / sqrt(128)) . V[b, 0 .. pos0 + s, g, :],   g = h / (n_heads / n_kv_heads)
 * as one flash-style MFMA kernel (fp32 scores and softmax, fp16 probabilities, like the eager HF path).  q must already be
 * rotated (amq_rope_cache_f16 / amq_rope_rows_f16), k rotated keys.  Layouts are given by strides in halves (multiples of 8):
 * element (b, s, h, d) of q / out at b*bstride + s*rstride + h*128 + d; key / value row t of kv head g at
 * b*bstride + t*rstride + g*hstride + d -- so the KV cache [n_kv_heads, max_seq, 128] (rstride 128, hstride max_seq*128) and
 * a projection output [rows, n_kv_heads*128] (rstride n_kv_heads*128, hstride 128) are both read in place, and out can be the
 * [rows, n_heads*128] matrix o_proj consumes.  Replaces the eager attention of the reference's prefill branch
 * (amq/kernel/monkeypatch/ftllama_modeling.py:88-126). */
int amq_attn_prefill_f16(const void* q, const void* k, const void* v, void* out, int batch, int S, int pos0, int n_heads,
                         int n_kv_heads, int head_dim, long long q_rstride, long long q_bstride, long long k_rstride,
                         long long k_bstride, long long k_hstride, long long v_rstride, long long v_bstride, long long v_hstride,
                         long long o_rstride, long long o_bstride, void* stream);
/* The same for ONE sequence with the result written in fragment order (amq_xfrag_f16's layout of the [S, n_heads * 128]
 * matrix; rows S .. 64 ceil(S / 64) - 1 zero): what o_proj's few-row GEMM (amq_gemm_xfrag_f16) reads, without the
 * re-ordering launch in between.  out_xf: amq_xfrag_bytes(S, n_heads * 128) bytes. */
int amq_attn_prefill_xfrag_f16(const void* q, const void* k, const void* v, void* out_xf, int S, int pos0, int n_heads,
                               int n_kv_heads, int head_dim, long long q_rstride, long long k_rstride, long long k_hstride,
                               long long v_rstride, long long v_hstride, void* stream);
/* The two decode-step restructurings that were built, proven bit-identical and measured SLOWER than the five-launch step -- q / k / v +
 * attention as one launch (amq_gemv_qkv_attn_f16) and one persistent launch per token (amq_decode_engine_*) -- are A/B routes: they are
 * declared in include/amq_hip_ab.h and exported by libamq_hip_ab.so (`make -C amq_amd/csrc ab`), not by this library. */

/* RoPE + KV-cache write for S new rows at positions pos0 .. pos0+S-1 of ONE sequence: q fp16 [S, n_heads*128] is
 * rotated in place; k [S, n_kv_heads*128] is rotated into kcache[h][pos0+s][:], v copied into vcache (both
 * [n_kv_heads, max_seq, 128]); rope_table from amq_rope_table_f16 (rows past rope_rows-1 clamp).  Same numerics as the
 * rotation inside amq_attn_decode_f16. */
int amq_rope_cache_f16(void* q, const void* k, const void* v, void* kcache, void* vcache, const void* rope_table,
                       int rope_rows, int pos0, int S, int n_heads, int n_kv_heads, int head_dim, int max_seq, void* stream);
/* The same for `batch` sequences of S rows each in one launch: q [batch*S, n_heads*128], k / v [batch*S, n_kv_heads*128],
 * caches [batch, n_kv_heads, max_seq, 128]; row s belongs to sequence s / S at position pos0 + s % S.  (The prompt pass of a
 * batched decode runner; counterpart of the per-sequence loop the reference's batch-1 FT cache forces,
 * amq/kernel/monkeypatch/ftllama_modeling.py:61-68.) */
int amq_rope_cache_batch_f16(void* q, const void* k, const void* v, void* kcache, void* vcache, const void* rope_table,
                             int rope_rows, int pos0, int S, int batch, int n_heads, int n_kv_heads, int head_dim, int max_seq,
                             void* stream);
/* RoPE in place on q [rows, n_heads*128] and k [rows, n_kv_heads*128] for rows = batch * seq_len (no cache write):
 * position of row s = pos0 + s % seq_len.  For the batched prompt pass of the harness' GeMM mode (amq/utils/speed.py:61-71
 * with batch_size > 1, BASELINE.json configs[3]). */
int amq_rope_rows_f16(void* q, void* k, const void* rope_table, int rope_rows, int pos0, int rows, int seq_len, int n_heads,
                      int n_kv_heads, int head_dim, void* stream);
/* out = fp16(silu(gate)) * up elementwise, n fp16 elements (n % 8 == 0): the LlamaMLP activation between up/gate and down */
int amq_silu_mul_f16(const void* gate, const void* up, void* out, size_t n, void* stream);

/* ---- bfloat16 variants (optional) ------------------------------------------------------------------
 * For models quantized with compute_dtype = torch.bfloat16: HQQLinear keeps scale / zero in the compute dtype and dequantizes in it
 * (hqq/core/quantize.py:184-199, 396-407, 516), W = bf16(bf16(q - zero) * scale).  The native payload is the fp16 path's; the meta words hold
 * bfloat16 (scale, zero) pairs -- amq_repack_from_hqq copies 16-bit patterns, so it repacks a bf16 model as it stands.  A native buffer repacked from
 * bf16 meta must only be given to the *_bf16 entry points (and one repacked from fp16 meta only to the *_f16 ones): the meta carries no dtype tag.
 * AMQ_MODE_HQQ arithmetic and groups of 128 (or multiples) only: the reference's GPTQ / AWQ kernels are fp16-only (hqq/backends/ft.py:62).
 * x, bias, residual, y, W_out: bfloat16.  Weights bit-identical to Quantizer.dequantize; fp32 accumulation, one bf16 rounding of y, bias and residual as
 * separate bf16 adds. */
int amq_dequantize_bf16(int bits, const void* qweight_native, const void* meta_native_bf16, int N, int K, int group, void* W_out, void* stream);
/* Format A with bfloat16 scale / zero -> W[N,K] bfloat16: Quantizer.dequantize under compute_dtype = bfloat16 as one kernel (any group amq_dequantize_hqq_f16 takes) */
int amq_dequantize_hqq_bf16(int bits, const void* W_q, const void* scale, const void* zero, int N, int K, int group, void* W_out, void* stream);
/* 1 .. 16 rows: weight-streaming kernel (unpacked block = MFMA operand, v_mfma_f32_16x16x32_bf16); x_stride / y_stride in elements (0 = dense;
 * x_stride a multiple of 8: rows are read in 16-byte pieces).  residual (or null): y = residual + (x W^T + bias), bfloat16 [M, y_stride], may alias y */
int amq_gemv_bf16(int bits, const void* x, const void* qweight_native, const void* meta_native_bf16, const void* bias, const void* residual, void* y,
                  int M, int N, int K, int group, int x_stride, int y_stride, void* stream);
/* any M: up to 16 rows amq_gemv_bf16; beyond, dequantize once into `workspace` (>= amq_gemm_bf16_workspace_bytes(M, N, K) bytes = N * K * 2 from 17 rows)
 * and multiply with the bf16 instantiation of the MFMA-bound GEMM kernel (amq_gemm_f16.hip); x_stride % 8 == 0 and y_stride % 4 == 0 there */
size_t amq_gemm_bf16_workspace_bytes(int M, int N, int K);
int amq_gemm_bf16(int bits, const void* x, const void* qweight_native, const void* meta_native_bf16, const void* bias, const void* residual, void* y,
                  int M, int N, int K, int group, int x_stride, int y_stride, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMQ_HIP_H */
