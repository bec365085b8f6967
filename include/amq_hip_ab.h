/* amq_hip_ab.h -- A/B routes of the decode step: built, parity-green against the product step, SLOWER, and therefore NOT part of
 * libamq_hip.so.  `make -C amq_amd/csrc ab` builds libamq_hip_ab.so = the product library's sources + these routes
 * (-DAMQ_AB_ROUTES); amq_amd loads it only for ops.DecodeEngine / ops.gemv_qkv_attn (QuantLlama(engine=True), fuse_qkv_attn,
 * bench.py --engine / --fuse-qkv-attn) and fails loudly when it has not been built.  Same conventions as amq_hip.h (status codes,
 * caller-owned buffers, stream argument, no option state).  Measurements: HISTORY.md 3.2b / 3.2c, profiles/r03_engine_timeline.txt,
 * profiles/r03_qkv_attn_fused_negative.txt. */
#ifndef AMQ_HIP_AB_H
#define AMQ_HIP_AB_H
#include "amq_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* q / k / v projections + decode attention of one block in ONE launch (batch 1): segments[0..2] = q, k, v exactly as for
 * amq_gemv_grouped_f16 with AMQ_PRO_RMSNORM (their y = the q / k / v vectors, still written), then amq_attn_decode_cur_f16's step
 * inside the same kernel: the workgroups that produce a head's 24 row-tiles arrive on that head's ticket, the workgroup owning
 * the head's first q row-tile waits for it (bounded) and runs the attention.  out, the cache rows and the q / k / v vectors are
 * bit-identical to the two separate calls.  tickets: int32 [n_heads], zero before the first call (each call leaves them zero).
 * K <= 8192.  A ticket that does not fill (never observed) or a position outside the cache raises the step state's error word;
 * after such an error the caller re-zeroes the tickets (a timed-out ticket is left as it is: late producers may still add to it). */
int amq_gemv_qkv_attn_f16(const amq_segment* segments /* host, 3 */, const void* x, const void* gamma, float eps, int K, int group,
                          void* kcache, void* vcache, void* out, const void* step_state, int n_heads, int n_kv_heads, int head_dim,
                          int max_seq, void* tickets, void* stream);

/* ---- one decode token as ONE persistent launch -----------------------------------------------------------------------
 * The whole per-token loop body of the reference's patched decoder (amq/kernel/monkeypatch/ftllama_modeling.py:167-230: for
 * every block RMSNorm -> q/k/v -> RoPE + cache append + attention -> o_proj + residual -> RMSNorm -> gate/up -> SiLU*mul ->
 * down_proj + residual, the seven linears being the per-layer 2/3/4-bit modules of amq_speed_benchmark.py:231-256) as one
 * kernel: one 16-wave workgroup per CU, the stages separated by a device-wide barrier, every wave's weight loads running
 * ahead across the barriers (amq_engine.hip).  Batch 1, head_dim 128, no linear biases.
 * Results are bit-identical to the same step issued as amq_gemv_grouped_f16 launches with amq_gemv_opts.waves = 16 (same
 * prologues / residual epilogues) + amq_attn_decode_cur_f16.
 *
 * amq_engine_block: HOST description of one decoder block; lin[] in the order q, k, v, o, gate, up, down (native buffers;
 * N of q / o / down = hidden, of k / v = n_kv_heads * 128, of gate / up = inter; K = hidden, down: inter); ln1 / ln2 fp16
 * [hidden]; kcache / vcache fp16 [n_kv_heads, max_seq, 128].
 * amq_decode_engine_image writes the device table for n_block blocks into HOST memory `image`
 * (amq_decode_engine_image_bytes(n_block) bytes); the caller copies it to the device once (blocks_dev below) -- the library
 * itself never copies or allocates.
 * amq_decode_engine_f16: x fp16 [hidden] is the residual stream (embedding of the token in, final hidden state out);
 * scratch >= amq_decode_engine_scratch_bytes(); step_state = the 264-byte block of amq_attn_decode_cur_f16 (cos/sin row of
 * the current position, int32 position, sticky error word: a position outside the cache skips the attention and raises
 * it); sync >= amq_decode_engine_sync_bytes(): the barrier words -- zero-filled by the CALLER once before the first launch
 * and never touched by it afterwards (they count barriers across launches; re-zero them after an error or when `grid`
 * changes).  An internal poll that runs into its bound raises word [4160] (byte 16640: 0x10000 | barrier number, sticky) and
 * every workgroup leaves.  grid = workgroups, 0 = one per CU. */
typedef struct amq_engine_linear { const void* qweight_native; const void* meta_native; int N; int bits; int mode; int reserved; } amq_engine_linear;
typedef struct amq_engine_block { amq_engine_linear lin[7]; const void* ln1; const void* ln2; void* kcache; void* vcache; } amq_engine_block;
size_t amq_decode_engine_image_bytes(int n_block);
size_t amq_decode_engine_scratch_bytes(int hidden, int inter, int n_kv_heads);
size_t amq_decode_engine_sync_bytes(void);
int amq_decode_engine_image(const amq_engine_block* blocks /* host */, int n_block, int hidden, int inter, int n_heads,
                            int n_kv_heads, int head_dim, int group, void* image /* host, out */);
int amq_decode_engine_f16(const void* blocks_dev, int n_block, int hidden, int inter, int n_heads, int n_kv_heads, int head_dim,
                          int max_seq, float eps, void* x, void* scratch, size_t scratch_bytes, const void* step_state,
                          void* sync, size_t sync_bytes, int grid, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* AMQ_HIP_AB_H */
