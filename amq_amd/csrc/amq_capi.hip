// amq_capi.hip -- extern "C" boundary of libamq_hip.so (see include/amq_hip.h).
// Argument validation + dispatch only; no allocation, no synchronisation.
#include "../../include/amq_hip.h"
#ifdef AMQ_AB_ROUTES
#include "../../include/amq_hip_ab.h"
#endif
#include "amq_common.cuh"
#include "amq_kernels.h"

#include <stdarg.h>
#include <stdio.h>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return AMQ_OK;
    return fail(AMQ_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}

int check_shape(int bits, int N, int K, int group) {
    if (bits != 2 && bits != 3 && bits != 4) return fail(AMQ_EINVAL, "bits must be 2, 3 or 4 (got %d)", bits);
    // 128, or a multiple of it that divides K: the repack entry points read such a source format and replicate each group's
    // (scale, zero) into the native layout's per-128 pairs; the compute entry points work on the native layout either way
    // 64 / 32: the native meta holds 128 / group pairs per (row, tile) (amq_common.cuh); served by the repack / dequantize entry points, the GEMV
    // kernel (<= 16 rows, exact math) and the dequantize-once GEMM route -- the other entry points refuse them (check_shape128)
    if (group != 64 && group != 32 && (group < 128 || (group % 128) != 0))
        return fail(AMQ_ESHAPE, "group size must be 32, 64 or a multiple of 128 (got %d)", group);
    if (N <= 0 || K <= 0 || (N % 16) != 0 || (K % 128) != 0)
        return fail(AMQ_ESHAPE, "need N %% 16 == 0 and K %% 128 == 0 (got N=%d K=%d)", N, K);
    if ((K % group) != 0) return fail(AMQ_ESHAPE, "group size %d does not divide K=%d", group, K);
    return AMQ_OK;
}

// entry points whose kernels read ONE (scale, zero) pair per (row, 128-column tile)
int check_shape128(int bits, int N, int K, int group, const char* who) {
    if (group == 64 || group == 32)
        return fail(AMQ_ESHAPE, "%s serves groups of 128 (and multiples); groups of %d run through amq_gemv_f16 / amq_gemv_grouped_f16 (<= 16 rows) and "
                    "amq_gemm_route_f16 / amq_gemm_gated_f16 with a workspace of N * K * 2 bytes", who, group);
    return check_shape(bits, N, K, group);
}

int check_mode(int mode) {
    if (mode != AMQ_MODE_HQQ && mode != AMQ_MODE_FMA && mode != AMQ_MODE_FMA1) return fail(AMQ_EINVAL, "unknown dequant mode %d", mode);
    return AMQ_OK;
}
// AMQ_MODE_FMA1 (scales within amq_fma1_scale_bound: the GEMV kernel's one-op unpack) is AMQ_MODE_FMA to every other kernel: the same results
int kernel_mode(int mode) { return mode == AMQ_MODE_FMA1 ? AMQ_MODE_FMA : mode; }

constexpr size_t LDS_LIMIT = 160 * 1024;

}  // namespace

extern "C" {

int amq_version(void) { return AMQ_VERSION; }
const char* amq_last_error(void) { return g_err; }

int amq_query(int K, int* out, int cap) {
    int vals[6];
    int maxm = 0, maxm_plain = 0, maxm_plain_nonorm = 0;
    for (int m = 1; m <= amq::GEMV_MAX_M; ++m) {
        if (amq::gemv_min_lds_bytes(m, K, false) <= LDS_LIMIT) maxm = m;
        if (amq::gemv_min_lds_bytes(m, K, true, true) <= LDS_LIMIT) maxm_plain = m;
        if (amq::gemv_min_lds_bytes(m, K, true, false) <= LDS_LIMIT) maxm_plain_nonorm = m;
    }
    vals[0] = maxm;                 // largest M amq_gemv_f16 / amq_gemv_grouped_f16 accept for this K whatever the options and group size
    vals[1] = (int)LDS_LIMIT;
    vals[2] = amq::TILE_N;
    vals[3] = amq::TILE_K;
    vals[4] = maxm_plain;           // ... with default options over groups of 128 (2 .. 8 rows run kernels with a smaller cross-wave sum buffer)
    vals[5] = maxm_plain_nonorm;    // ... and no RMSNorm prologue (x may then be staged in two K phases: 8 rows at K = 11008)
    int n = cap < 6 ? cap : 6;
    for (int i = 0; i < n; ++i) out[i] = vals[i];
    return n;
}

size_t amq_native_qweight_bytes(int bits, int N, int K) { return amq::native_qweight_bytes(bits, N, K); }
float amq_fma1_scale_bound(int bits) { return (bits == 2 || bits == 3 || bits == 4) ? amq::fma1_scale_bound(bits) : 0.0f; }
size_t amq_native_meta_bytes(int N, int K, int group) { return amq::native_meta_bytes(N, K, group > 0 ? amq::meta_pairs(group) : 1); }

int amq_repack_from_hqq(int bits, const void* W_q, const void* scale, const void* zero, int N, int K, int group,
                        void* qn, void* mn, void* stream) {
    if (int rc = check_shape(bits, N, K, group)) return rc;
    if (!W_q || !scale || !zero || !qn || !mn) return fail(AMQ_EINVAL, "null pointer");
    return check_hip(amq::launch_repack(amq::FMT_HQQ, bits, W_q, scale, zero, N, K, qn, mn, (hipStream_t)stream, group), "repack_from_hqq");
}

int amq_repack_from_gptq(int bits, const void* qweight, const void* scales, const void* zeros, int N, int K, int group,
                         void* qn, void* mn, void* stream) {
    if (int rc = check_shape(bits, N, K, group)) return rc;
    if (!qweight || !scales || !zeros || !qn || !mn) return fail(AMQ_EINVAL, "null pointer");
    return check_hip(amq::launch_repack(amq::FMT_GPTQ, bits, qweight, scales, zeros, N, K, qn, mn, (hipStream_t)stream, group), "repack_from_gptq");
}

int amq_repack_from_awq(const void* qweight, const void* scales, const void* scaled_zeros, int N, int K, int group,
                        void* qn, void* mn, void* stream) {
    if (int rc = check_shape(4, N, K, group)) return rc;
    if (!qweight || !scales || !scaled_zeros || !qn || !mn) return fail(AMQ_EINVAL, "null pointer");
    return check_hip(amq::launch_repack(amq::FMT_AWQ, 4, qweight, scales, scaled_zeros, N, K, qn, mn, (hipStream_t)stream, group), "repack_from_awq");
}

int amq_dequantize_f16(int bits, int mode, const void* qn, const void* mn, int N, int K, int group, void* W, void* stream) {
    if (int rc = check_shape(bits, N, K, group)) return rc;
    if (int rc = check_mode(mode)) return rc;
    mode = kernel_mode(mode);
    if (!qn || !mn || !W) return fail(AMQ_EINVAL, "null pointer");
    return check_hip(amq::launch_dequantize(bits, mode, qn, mn, N, K, W, (hipStream_t)stream, amq::meta_pairs(group)), "dequantize");
}

int amq_dequantize_hqq_f16(int bits, const void* W_q, const void* scale, const void* zero, int N, int K, int group,
                           void* W, void* stream) {
    if (int rc = check_shape(bits, N, K, group)) return rc;
    if (!W_q || !scale || !zero || !W) return fail(AMQ_EINVAL, "null pointer");
    return check_hip(amq::launch_dequantize_hqq(bits, W_q, scale, zero, N, K, W, (hipStream_t)stream, group), "dequantize_hqq");
}

#ifndef AMQ_GEMV_DEFAULT_MATH
#define AMQ_GEMV_DEFAULT_MATH AMQ_MATH_EXACT
#endif
int amq_default_gemv_math(void) { return AMQ_GEMV_DEFAULT_MATH; }

int amq_gemv_grouped_f16(const amq_segment* segs, int nseg, const void* x, const void* x2, const void* gamma,
                         float eps, int prologue, int M, int K, int group, int x_stride, const amq_gemv_opts* opts,
                         void* stream) {
    if (!segs || nseg < 1 || nseg > AMQ_MAX_SEGMENTS) return fail(AMQ_EINVAL, "nseg must be 1..%d (got %d)", AMQ_MAX_SEGMENTS, nseg);
    if (!x) return fail(AMQ_EINVAL, "null x");
    if (prologue < AMQ_PRO_NONE || prologue > AMQ_PRO_SILU_MUL) return fail(AMQ_EINVAL, "unknown prologue %d", prologue);
    if (prologue == AMQ_PRO_RMSNORM && !gamma) return fail(AMQ_EINVAL, "RMSNorm prologue needs gamma");
    if (prologue == AMQ_PRO_SILU_MUL && !x2) return fail(AMQ_EINVAL, "SiLU*mul prologue needs x2");
    if (M < 1) return fail(AMQ_ESHAPE, "M must be >= 1 (got %d)", M);
    amq_gemv_opts o{};                                   // all zero = defaults
    if (opts) o = *opts;
    if (o.math < AMQ_MATH_DEFAULT || o.math > AMQ_MATH_EXACT)
        return fail(AMQ_EINVAL, "opts.math must be AMQ_MATH_DEFAULT, AMQ_MATH_EXACT, AMQ_MATH_GROUPSCALE or AMQ_MATH_LINEAR");
    if (o.math == AMQ_MATH_DEFAULT) o.math = amq_default_gemv_math();
    if (o.waves != 0 && o.waves != 4 && o.waves != 8 && o.waves != 16) return fail(AMQ_EINVAL, "opts.waves must be 0, 4, 8 or 16");
    if (o.depth != 0 && o.depth != 2 && o.depth != 4) return fail(AMQ_EINVAL, "opts.depth must be 0, 2 or 4");
    if (o.rpt < 0 || o.rpt > 64) return fail(AMQ_EINVAL, "opts.rpt (row-tiles per workgroup) must be 0..64");
    const bool plain_form = !o.dot && o.math != AMQ_MATH_LINEAR && o.depth != 4 && amq::meta_pairs(group) == 1 && o.waves != 4;
    // (rows staged in two K phases -- 7 - 8 rows of K = 11008 -- need dense x rows and no full-row statistic: launch_gemv makes the same decision)
    const bool whole_rows = prologue == AMQ_PRO_RMSNORM || (x_stride != 0 && x_stride != K);
    if (M > amq::GEMV_MAX_M || amq::gemv_min_lds_bytes(M, K, plain_form && o.waves == 0 && o.rpt == 0, whole_rows) > LDS_LIMIT)
        return fail(AMQ_ESHAPE, "M=%d rows of K=%d%s do not fit LDS for the GEMV path; use amq_gemm_f16", M, K, (x_stride != 0 && x_stride != K) ? " (strided x)" : "");
    amq::GemvArgs a{};
    for (int i = 0; i < nseg; ++i) {
        const amq_segment& s = segs[i];
        if (int rc = check_shape(s.bits, s.N, K, group)) return rc;
        if (int rc = check_mode(s.mode)) return rc;
        if (!s.qweight_native || !s.meta_native || !s.y) return fail(AMQ_EINVAL, "segment %d: null pointer", i);
        amq::GemvSeg& d = a.seg[i];
        d.qweight = s.qweight_native; d.meta = s.meta_native; d.bias = s.bias; d.residual = s.residual; d.y = s.y;
        d.N = s.N; d.bits = s.bits; d.mode = s.mode;
        d.y_stride = s.y_stride ? s.y_stride : s.N;
    }
    a.nseg = nseg; a.M = M; a.K = K; a.x_stride = x_stride ? x_stride : K;
    a.x = x; a.x2 = x2; a.gamma = gamma; a.eps = eps; a.prologue = prologue;
    a.flags = (o.dot ? amq::GEMV_FLAG_DOT : 0) | (o.math == AMQ_MATH_LINEAR ? amq::GEMV_FLAG_LINEAR : 0) |
              (o.math == AMQ_MATH_GROUPSCALE ? amq::GEMV_FLAG_GS : 0);
    a.force_waves = o.waves;
    a.force_depth = o.depth;
    a.force_rpt = o.rpt;
    a.gp = amq::meta_pairs(group);
    if (a.gp > 1 && (o.dot || o.math == AMQ_MATH_LINEAR || o.depth == 4))
        return fail(AMQ_EINVAL, "groups of %d: the GEMV kernel's default form only (opts.math = AMQ_MATH_EXACT, opts.dot = 0, opts.depth = 0 / 2)", group);
    return check_hip(amq::launch_gemv(a, (hipStream_t)stream), "gemv");
}

int amq_gemv_grouped_sums_f16(const amq_segment* segs, int nseg, const void* x, const void* gamma, float eps, const float* sums_in,
                              float* sums_out, int M, int K, int group, void* stream) {
    if (!segs || nseg < 1 || nseg > AMQ_MAX_SEGMENTS) return fail(AMQ_EINVAL, "nseg must be 1..%d (got %d)", AMQ_MAX_SEGMENTS, nseg);
    if (!x) return fail(AMQ_EINVAL, "null x");
    if ((sums_in != nullptr) != (gamma != nullptr)) return fail(AMQ_EINVAL, "sums_in (the partial sums of squares of x) and gamma go together");
    if (M < 2 || M > 8) return fail(AMQ_ESHAPE, "the partial-sum forms live in the 2 .. 8-row kernels (got M=%d); one row: amq_gemv_grouped_f16 with AMQ_PRO_RMSNORM", M);
    if (amq::meta_pairs(group) != 1) return fail(AMQ_ESHAPE, "groups of 128 (and multiples) only (got %d)", group);
    if (K < 2048) return fail(AMQ_ESHAPE, "K >= 2048 (the 2 .. 8-row kernels run 8 or 16 waves per workgroup; got K=%d)", K);
    if (sums_out && nseg != 1) return fail(AMQ_EINVAL, "sums_out describes ONE output: nseg must be 1");
    if (sums_out && sums_in) return fail(AMQ_EINVAL, "sums_out is written by launches without a prologue (o_proj, down_proj): not together with sums_in");
    if (sums_in && K > 8192) return fail(AMQ_ESHAPE, "sums_in: K <= 8192 (K / 16 <= 512 partials per row; got K=%d)", K);
    const bool phased = amq::gemv_rows_phased(M, K, true, sums_in != nullptr);
    if (sums_in && phased) return fail(AMQ_ESHAPE, "M=%d rows of K=%d are staged in two K phases: no fused norm there", M, K);
    if (amq::gemv_min_lds_bytes(M, K, true, sums_in != nullptr) > LDS_LIMIT)
        return fail(AMQ_ESHAPE, "M=%d rows of K=%d do not fit LDS for the GEMV path", M, K);
    amq::GemvArgs a{};
    for (int i = 0; i < nseg; ++i) {
        const amq_segment& s = segs[i];
        if (int rc = check_shape(s.bits, s.N, K, group)) return rc;
        if (int rc = check_mode(s.mode)) return rc;
        if (!s.qweight_native || !s.meta_native || !s.y) return fail(AMQ_EINVAL, "segment %d: null pointer", i);
        amq::GemvSeg& d = a.seg[i];
        d.qweight = s.qweight_native; d.meta = s.meta_native; d.bias = s.bias; d.residual = s.residual; d.y = s.y;
        d.N = s.N; d.bits = s.bits; d.mode = s.mode;
        d.y_stride = s.y_stride ? s.y_stride : s.N;
    }
    a.nseg = nseg; a.M = M; a.K = K; a.x_stride = K;
    a.x = x; a.x2 = nullptr; a.gamma = gamma; a.eps = eps;
    a.prologue = sums_in ? amq::PRO_RMSNORM_SUMS : amq::PRO_NONE;
    a.sums_in = sums_in; a.sums_out = sums_out;
    a.gp = 1;
    return check_hip(amq::launch_gemv(a, (hipStream_t)stream), "gemv_sums");
}

int amq_gemv_f16(int bits, int mode, const void* x, const void* qn, const void* mn, const void* bias, void* y,
                 int M, int N, int K, int group, int x_stride, int y_stride, void* stream) {
    amq_segment s{};
    s.qweight_native = qn; s.meta_native = mn; s.bias = bias; s.residual = nullptr; s.y = y;
    s.N = N; s.bits = bits; s.mode = mode; s.y_stride = y_stride;
    return amq_gemv_grouped_f16(&s, 1, x, nullptr, nullptr, 0.f, AMQ_PRO_NONE, M, K, group, x_stride, nullptr, stream);
}

int amq_gemm_f16(int bits, int mode, const void* x, const void* qn, const void* mn, const void* bias, void* y,
                 int M, int N, int K, int group, int x_stride, int y_stride, void* stream) {
    if (int rc = check_shape(bits, N, K, group)) return rc;          // (groups of 64 / 32: the few-row or the tiled kernel -- no workspace here, so never dequantize-once)
    if (int rc = check_mode(mode)) return rc;
    mode = kernel_mode(mode);
    if (!x || !qn || !mn || !y) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1) return fail(AMQ_ESHAPE, "M must be >= 1 (got %d)", M);
    amq::GemmArgs a{x, qn, mn, bias, y, M, N, K, bits, mode, x_stride ? x_stride : K, y_stride ? y_stride : N, nullptr, 1,
                    nullptr, nullptr, nullptr, amq::meta_pairs(group)};
    return check_hip(amq::launch_gemm(a, (hipStream_t)stream), "gemm");
}

size_t amq_gemm_splitk_workspace_bytes(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 128) return 0;
    const int s = amq::gemm_pick_splits(M, N, K);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)N * sizeof(float) : 0;
}

int amq_gemm_splitk_f16(int bits, int mode, const void* x, const void* qn, const void* mn, const void* bias, void* y,
                        int M, int N, int K, int group, int x_stride, int y_stride, void* workspace, size_t workspace_bytes,
                        void* stream) {
    const size_t need = amq_gemm_splitk_workspace_bytes(M, N, K);
    if (need == 0) return amq_gemm_f16(bits, mode, x, qn, mn, bias, y, M, N, K, group, x_stride, y_stride, stream);
    if (int rc = check_shape(bits, N, K, group)) return rc;
    if (int rc = check_mode(mode)) return rc;
    mode = kernel_mode(mode);
    if (!x || !qn || !mn || !y) return fail(AMQ_EINVAL, "null pointer");
    if (!workspace || workspace_bytes < need) return fail(AMQ_EINVAL, "split-K workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    amq::GemmArgs a{x, qn, mn, bias, y, M, N, K, bits, mode, x_stride ? x_stride : K, y_stride ? y_stride : N,
                    (float*)workspace, amq::gemm_pick_splits(M, N, K), nullptr, nullptr, nullptr, amq::meta_pairs(group)};
    return check_hip(amq::launch_gemm(a, (hipStream_t)stream), "gemm_splitk");
}

// the workspace of a route call holds the dequantized fp16 weights (dequantize-once route) or the split-K partials, never both
static bool route_is_deq(int route, int M, int N, int K, int group = 128) {
    if (group == 64 || group == 32)       // few rows: the pair-aware few-row kernel; then the tiled kernel; launches that fill 256 x 256 tiles: dequantize-once
        return route == AMQ_GEMM_DEQ || (route == AMQ_GEMM_AUTO && amq::gemm_fine_takes_deq(M, N, K));
    return route == AMQ_GEMM_DEQ || (route == AMQ_GEMM_AUTO && amq::gemm_takes_deq(M, N, K));
}

size_t amq_gemm_route_workspace_bytes_g(int route, int M, int N, int K, int group) {
    if (M < 1 || N < 1 || K < 128 || route < AMQ_GEMM_AUTO || route > AMQ_GEMM_DEQ) return 0;
    if (route_is_deq(route, M, N, K, group)) return (size_t)N * (size_t)K * 2;       // the dequantized fp16 weights
    if ((group == 64 || group == 32) && amq::gemm_fine_takes_skinny(M)) return 0;     // (their few-row kernel takes no workspace)
    const int s = amq::gemm_pick_splits(M, N, K, (group == 64 || group == 32) && route == AMQ_GEMM_AUTO ? (int)AMQ_GEMM_TILED : route);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)N * sizeof(float) : 0;
}
size_t amq_gemm_route_workspace_bytes(int route, int M, int N, int K) { return amq_gemm_route_workspace_bytes_g(route, M, N, K, 128); }

// groups of 64 / 32 on a route call: the few-row kernel (AUTO / SKINNY) up to gemm_fine_takes_skinny rows, the tiled kernel (AUTO / TILED) or
// dequantize-once (AUTO where it fills the chip, given the workspace; DEQ) beyond; the ring / wave-specialised kernels read one pair per tile
static int check_fine_route(int route, int group, int M, int N, int K, const void* workspace) {
    if (group != 64 && group != 32) return AMQ_OK;
    if (route == AMQ_GEMM_RING || route == AMQ_GEMM_RING128 || route == AMQ_GEMM_WS)
        return fail(AMQ_EINVAL, "groups of %d: the ring / wave-specialised kernels read one (scale, zero) per 128 columns (use AMQ_GEMM_AUTO, _SKINNY, _TILED or the dequantize-once route _DEQ)", group);
    if (route == AMQ_GEMM_DEQ && !workspace) return fail(AMQ_EINVAL, "AMQ_GEMM_DEQ needs a workspace of N * K * 2 bytes for the fp16 weights");
    (void)M; (void)N; (void)K;
    return AMQ_OK;
}

static int gemm_route_impl(int route, int bits, int mode, const void* x, const void* qn, const void* mn, const void* bias,
                           const void* residual, void* y, int M, int N, int K, int group, int x_stride, int y_stride,
                           void* workspace, size_t workspace_bytes, void* stream, amq::GemmNorm* norm) {
    if (route < AMQ_GEMM_AUTO || route > AMQ_GEMM_DEQ) return fail(AMQ_EINVAL, "unknown GEMM route %d", route);
    if (int rc = check_shape(bits, N, K, group)) return rc;
    if (int rc = check_mode(mode)) return rc;
    mode = kernel_mode(mode);
    if (!x || !qn || !mn || !y) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1) return fail(AMQ_ESHAPE, "M must be >= 1 (got %d)", M);
    if (route == AMQ_GEMM_SKINNY && M > 64 && !((group == 64 || group == 32) && amq::gemm_fine_takes_skinny(M)))
        return fail(AMQ_ESHAPE, "the few-row kernel takes at most 64 rows (got %d)", M);
    if (int rc = check_fine_route(route, group, M, N, K, workspace)) return rc;
    const bool fine = group == 64 || group == 32;
    const size_t need = amq_gemm_route_workspace_bytes_g(route, M, N, K, group);
    const bool deq = route_is_deq(route, M, N, K, group);
    if (route == AMQ_GEMM_DEQ && !workspace) return fail(AMQ_EINVAL, "AMQ_GEMM_DEQ needs a workspace of N * K * 2 bytes for the fp16 weights");
    const bool use_ws = need != 0 && workspace != nullptr;         // no workspace: single pass through a fused kernel
    if (use_ws && workspace_bytes < need)
        return fail(AMQ_EINVAL, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    const bool split = use_ws && !deq;
    const int sroute = fine && route == AMQ_GEMM_AUTO ? (int)AMQ_GEMM_TILED : route;       // (groups of 64 / 32 beyond the few-row kernel: the tiled kernel)
    amq::GemmArgs a{x, qn, mn, bias, y, M, N, K, bits, mode, x_stride ? x_stride : K, y_stride ? y_stride : N,
                    split ? (float*)workspace : nullptr, split ? amq::gemm_pick_splits(M, N, K, sroute) : 1, residual, nullptr,
                    use_ws && deq ? workspace : nullptr, amq::meta_pairs(group)};
    if ((route == AMQ_GEMM_DEQ || (fine && deq)) && !amq::gemm_f16w_ok(M, N, K, a.x_stride, a.y_stride))
        return fail(AMQ_ESHAPE, "AMQ_GEMM_DEQ: strides must be multiples of 8 (x) / 4 (y) halves and x, W must each span < 4 GiB");
    return check_hip(amq::launch_gemm(a, (hipStream_t)stream, route, norm), "gemm");
}

int amq_gemm_route_f16(int route, int bits, int mode, const void* x, const void* qn, const void* mn, const void* bias,
                       const void* residual, void* y, int M, int N, int K, int group, int x_stride, int y_stride,
                       void* workspace, size_t workspace_bytes, void* stream) {
    return gemm_route_impl(route, bits, mode, x, qn, mn, bias, residual, y, M, N, K, group, x_stride, y_stride, workspace, workspace_bytes, stream, nullptr);
}

int amq_gemm_res_norm_xfrag_f16(int bits, int mode, const void* x, const void* qn, const void* mn, const void* bias, const void* residual,
                                void* y, int M, int N, int K, int group, int x_stride, void* workspace, size_t workspace_bytes,
                                const void* gamma, float eps, void* xf, void* stream) {
    if (!gamma || !xf) return fail(AMQ_EINVAL, "null pointer");
    if (N < 128 || (N % 128) != 0) return fail(AMQ_ESHAPE, "the normed rows are handed on in fragment order: N %% 128 == 0 (got N=%d)", N);
    amq::GemmNorm norm{gamma, eps, xf, false};
    return gemm_route_impl(AMQ_GEMM_AUTO, bits, mode, x, qn, mn, bias, residual, y, M, N, K, group, x_stride, 0, workspace, workspace_bytes, stream, &norm);
}

int amq_gemm_gated_fused_g(int route, int M, int N, int K, int use_workspace, int group) {
    if (M < 1 || N < 16 || K < 128 || route < AMQ_GEMM_AUTO || route > AMQ_GEMM_DEQ) return 0;
    const bool fine = group == 64 || group == 32;
    const bool deq = route_is_deq(route, M, N, K, group);
    const int sroute = fine && route == AMQ_GEMM_AUTO ? (int)AMQ_GEMM_TILED : route;
    amq::GemmArgs a{nullptr, nullptr, nullptr, nullptr, nullptr, M, N, K, 4, AMQ_MODE_HQQ, K, N, nullptr,
                    use_workspace && !deq ? amq::gemm_pick_splits(M, N, K, sroute) : 1, nullptr, nullptr,
                    use_workspace && deq ? (void*)&a : nullptr,       // (any non-null value: only tested for presence)
                    amq::meta_pairs(group > 0 ? group : 128)};
    return amq::gemm_gate_fused(a, route) ? 1 : 0;
}
int amq_gemm_gated_fused(int route, int M, int N, int K, int use_workspace) { return amq_gemm_gated_fused_g(route, M, N, K, use_workspace, 128); }

int amq_gemm_gated_f16(int route, int bits, int mode, const void* x, const void* qn, const void* mn, const void* bias,
                       const void* gate, void* y, int M, int N, int K, int group, int x_stride, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (route < AMQ_GEMM_AUTO || route > AMQ_GEMM_DEQ) return fail(AMQ_EINVAL, "unknown GEMM route %d", route);
    if (int rc = check_shape(bits, N, K, group)) return rc;
    if (int rc = check_mode(mode)) return rc;
    mode = kernel_mode(mode);
    if (!x || !qn || !mn || !y || !gate) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1) return fail(AMQ_ESHAPE, "M must be >= 1 (got %d)", M);
    if (N % 8) return fail(AMQ_ESHAPE, "the gated product needs N %% 8 == 0 (got %d)", N);
    if (route == AMQ_GEMM_SKINNY && M > 64 && !((group == 64 || group == 32) && amq::gemm_fine_takes_skinny(M)))
        return fail(AMQ_ESHAPE, "the few-row kernel takes at most 64 rows (got %d)", M);
    if (int rc = check_fine_route(route, group, M, N, K, workspace)) return rc;
    const size_t need = amq_gemm_route_workspace_bytes_g(route, M, N, K, group);
    const bool deq = route_is_deq(route, M, N, K, group);
    if (route == AMQ_GEMM_DEQ && !workspace) return fail(AMQ_EINVAL, "AMQ_GEMM_DEQ needs a workspace of N * K * 2 bytes for the fp16 weights");
    const bool use_ws = need != 0 && workspace != nullptr;
    if (use_ws && workspace_bytes < need)
        return fail(AMQ_EINVAL, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    const bool split = use_ws && !deq;
    const int sroute = (group == 64 || group == 32) && route == AMQ_GEMM_AUTO ? (int)AMQ_GEMM_TILED : route;
    amq::GemmArgs a{x, qn, mn, bias, y, M, N, K, bits, mode, x_stride ? x_stride : K, N,
                    split ? (float*)workspace : nullptr, split ? amq::gemm_pick_splits(M, N, K, sroute) : 1, nullptr, gate,
                    use_ws && deq ? workspace : nullptr, amq::meta_pairs(group)};
    if ((group == 64 || group == 32) && deq && !amq::gemm_f16w_ok(M, N, K, a.x_stride, a.y_stride))
        return fail(AMQ_ESHAPE, "groups of %d (dequantize-once route): x_stride must be a multiple of 8 halves, N of 4, x and W must each span < 4 GiB", group);
    if (gate == y && !amq::gemm_gate_fused(a, route))
        return fail(AMQ_EINVAL, "gate may alias y only where the kernel applies it in its epilogue (amq_gemm_gated_fused)");
    return check_hip(amq::launch_gemm(a, (hipStream_t)stream, route), "gemm_gated");
}

int amq_gemm_f16w_f16(const void* x, const void* w, const void* bias, const void* residual, const void* gate, void* y,
                      int M, int N, int K, int x_stride, int y_stride, void* stream) {
    if (!x || !w || !y) return fail(AMQ_EINVAL, "null pointer");
    if (residual && gate) return fail(AMQ_EINVAL, "residual and gate are exclusive");
    const int xs = x_stride ? x_stride : K, ys = y_stride ? y_stride : N;
    if (!amq::gemm_f16w_ok(M, N, K, xs, ys))
        return fail(AMQ_ESHAPE, "need M >= 1, N %% 16 == 0, K %% 128 == 0, x_stride %% 8 == 0, y_stride %% 4 == 0, x and W < 4 GiB each (got M=%d N=%d K=%d)", M, N, K);
    return check_hip(amq::launch_gemm_f16w(x, w, bias, residual, gate, y, M, N, K, xs, ys, (hipStream_t)stream), "gemm_f16w");
}

int amq_gemm_res_f16(int bits, int mode, const void* x, const void* qn, const void* mn, const void* bias, const void* residual,
                     void* y, int M, int N, int K, int group, int x_stride, int y_stride, void* workspace,
                     size_t workspace_bytes, void* stream) {
    return amq_gemm_route_f16(AMQ_GEMM_AUTO, bits, mode, x, qn, mn, bias, residual, y, M, N, K, group, x_stride, y_stride,
                              workspace, workspace_bytes, stream);
}

size_t amq_xfrag_bytes(int M, int K) {
    if (M < 1 || K < 128 || (K % 128) != 0) return 0;
    return (size_t)((M + 63) / 64) * 64 * (size_t)K * 2;
}

int amq_xfrag_f16(const void* src, void* xf, int M, int K, long long stride_m, long long stride_kt, void* stream) {
    if (!src || !xf) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1 || K < 128 || (K % 128) != 0) return fail(AMQ_ESHAPE, "need M >= 1 and K %% 128 == 0 (got M=%d K=%d)", M, K);
    if (stride_m < 0 || stride_kt < 128 || (stride_m % 8) != 0 || (stride_kt % 8) != 0)
        return fail(AMQ_ESHAPE, "strides must be multiples of 8 halves, stride_kt >= 128");
    return check_hip(amq::launch_xfrag(src, xf, M, K, (long)stride_m, (long)stride_kt, (hipStream_t)stream), "xfrag");
}

int amq_rmsnorm_xfrag_f16(const void* x, const void* gamma, void* xf, int M, int K, float eps, void* stream) {
    if (!x || !gamma || !xf) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1 || K < 128 || (K % 128) != 0) return fail(AMQ_ESHAPE, "need M >= 1 and K %% 128 == 0 (got M=%d K=%d)", M, K);
    return check_hip(amq::launch_rmsnorm_xfrag(x, gamma, xf, M, K, eps, (hipStream_t)stream), "rmsnorm_xfrag");
}

int amq_gemm_xfrag_f16(int bits, int mode, const void* xf, const void* qn, const void* mn, const void* bias,
                       const void* gate, const void* residual, void* y, int M, int N, int K, int group, int y_stride,
                       void* stream) {
    if (int rc = check_shape128(bits, N, K, group, "amq_gemm_xfrag_f16")) return rc;
    if (int rc = check_mode(mode)) return rc;
    mode = kernel_mode(mode);
    if (!xf || !qn || !mn || !y) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1) return fail(AMQ_ESHAPE, "M must be >= 1 (got %d)", M);
    if ((M + 63) / 64 > 65535) return fail(AMQ_ESHAPE, "M=%d exceeds one launch", M);
    amq::GemmArgs a{xf, qn, mn, bias, y, M, N, K, bits, mode, K, y_stride ? y_stride : N, nullptr, 1, residual, gate};
    return check_hip(amq::launch_gemm_xfrag(a, (hipStream_t)stream), "gemm_xfrag");
}

int amq_gemm_xfrag_grouped_f16(const amq_segment* segs, int nseg, const void* xf, int M, int K, int group, void* stream) {
    return amq_gemm_xfrag_grouped_form_f16(segs, nseg, xf, M, K, group, AMQ_FEWROW_AUTO, 0, stream);
}

int amq_gemm_xfrag_grouped_form_f16(const amq_segment* segs, int nseg, const void* xf, int M, int K, int group, int form, int blocks_per_wg,
                                    void* stream) {
    if (form < AMQ_FEWROW_AUTO || form > AMQ_FEWROW_STREAM) return fail(AMQ_EINVAL, "unknown few-row form %d", form);
    if (blocks_per_wg != 0 && (form != AMQ_FEWROW_STREAM || blocks_per_wg < 1 || blocks_per_wg > 6 || blocks_per_wg == 5))
        return fail(AMQ_EINVAL, "blocks_per_wg is 0, or 1 / 2 / 3 / 4 / 6 with AMQ_FEWROW_STREAM (got %d)", blocks_per_wg);
    if (!segs || nseg < 1 || nseg > AMQ_MAX_SEGMENTS) return fail(AMQ_EINVAL, "nseg must be 1..%d (got %d)", AMQ_MAX_SEGMENTS, nseg);
    if (!xf) return fail(AMQ_EINVAL, "null xf");
    if (M < 1) return fail(AMQ_ESHAPE, "M must be >= 1 (got %d)", M);
    if ((M + 63) / 64 > 65535) return fail(AMQ_ESHAPE, "M=%d exceeds one launch", M);
    amq::GemvSeg gs[AMQ_MAX_SEGMENTS] = {};
    for (int i = 0; i < nseg; ++i) {
        const amq_segment& s = segs[i];
        if (int rc = check_shape128(s.bits, s.N, K, group, "amq_gemm_xfrag_grouped_f16")) return rc;
        if (int rc = check_mode(s.mode)) return rc;
        if (!s.qweight_native || !s.meta_native || !s.y) return fail(AMQ_EINVAL, "segment %d: null pointer", i);
        amq::GemvSeg& d = gs[i];
        d.qweight = s.qweight_native; d.meta = s.meta_native; d.bias = s.bias; d.residual = s.residual; d.y = s.y;
        d.N = s.N; d.bits = s.bits; d.mode = kernel_mode(s.mode);
        d.y_stride = s.y_stride ? s.y_stride : s.N;
    }
    return check_hip(amq::launch_gemm_xfrag_grouped(xf, M, K, gs, nseg, (hipStream_t)stream, form, blocks_per_wg), "gemm_xfrag_grouped");
}

#ifdef AMQ_AB_ROUTES     /* A/B routes: exported by libamq_hip_ab.so (make ab), declared in include/amq_hip_ab.h */
int amq_gemv_qkv_attn_f16(const amq_segment* segs, const void* x, const void* gamma, float eps, int K, int group, void* kcache,
                          void* vcache, void* out, const void* step_state, int n_heads, int n_kv_heads, int head_dim, int max_seq,
                          void* tickets, void* stream) {
    if (!segs || !x || !gamma || !kcache || !vcache || !out || !step_state || !tickets) return fail(AMQ_EINVAL, "null pointer");
    if (head_dim != 128) return fail(AMQ_ESHAPE, "head_dim must be 128 (got %d)", head_dim);
    if (n_heads < 1 || n_kv_heads < 1 || (n_heads % n_kv_heads) != 0 || n_heads > 255) return fail(AMQ_ESHAPE, "bad head counts %d / %d", n_heads, n_kv_heads);
    if (K > 8192) return fail(AMQ_ESHAPE, "K = %d: the fused launch serves K <= 8192", K);
    if (max_seq < 1 || amq::gemv_qkv_attn_lds_bytes(K, max_seq) > LDS_LIMIT) return fail(AMQ_ESHAPE, "max_seq=%d too long for the single-pass decode attention", max_seq);
    const int Ns[3] = {n_heads * 128, n_kv_heads * 128, n_kv_heads * 128};
    amq::GemvArgs a{};
    for (int i = 0; i < 3; ++i) {
        const amq_segment& s = segs[i];
        if (int rc = check_shape128(s.bits, s.N, K, group, "amq_gemv_qkv_attn_f16")) return rc;
        if (int rc = check_mode(s.mode)) return rc;
        if (s.N != Ns[i]) return fail(AMQ_ESHAPE, "segment %d: N = %d, expected %d", i, s.N, Ns[i]);
        if (!s.qweight_native || !s.meta_native || !s.y) return fail(AMQ_EINVAL, "segment %d: null pointer", i);
        if (s.bias || s.residual) return fail(AMQ_EUNSUPPORTED, "segment %d: bias / residual are not part of the fused q/k/v + attention launch", i);
        amq::GemvSeg& d = a.seg[i];
        d.qweight = s.qweight_native; d.meta = s.meta_native; d.y = s.y; d.N = s.N; d.bits = s.bits; d.mode = kernel_mode(s.mode); d.y_stride = s.N;
    }
    a.nseg = 3; a.M = 1; a.K = K; a.x_stride = K; a.x = x; a.gamma = gamma; a.eps = eps; a.prologue = amq::PRO_RMSNORM;
    amq::AttnArgs t{segs[0].y, segs[1].y, segs[2].y, kcache, vcache, out, nullptr, 0, n_heads, n_kv_heads, max_seq, 10000.0f, nullptr, step_state};
    return check_hip(amq::launch_gemv_qkv_attn(a, t, (int*)tickets, (hipStream_t)stream), "gemv_qkv_attn");
}

size_t amq_decode_engine_image_bytes(int n_block) { return n_block > 0 ? amq::engine_image_bytes(n_block) : 0; }
size_t amq_decode_engine_scratch_bytes(int hidden, int inter, int n_kv_heads) {
    if (hidden < 1 || inter < 1 || n_kv_heads < 1) return 0;
    return amq::engine_scratch_bytes(hidden, inter, n_kv_heads);
}
size_t amq_decode_engine_sync_bytes(void) { return amq::engine_sync_bytes(); }

static int engine_shapes_ok(int hidden, int inter, int n_heads, int n_kv_heads, int head_dim) {
    if (head_dim != 128) return fail(AMQ_ESHAPE, "head_dim must be 128 (got %d)", head_dim);
    if (n_heads < 1 || n_kv_heads < 1 || (n_heads % n_kv_heads) != 0) return fail(AMQ_ESHAPE, "bad head counts %d / %d", n_heads, n_kv_heads);
    if (hidden != n_heads * 128) return fail(AMQ_ESHAPE, "hidden (%d) must equal n_heads * 128", hidden);
    if ((hidden % 128) != 0 || (inter % 128) != 0 || inter < 128) return fail(AMQ_ESHAPE, "hidden and inter must be multiples of 128");
    if (hidden > 32768 || inter > 32768) return fail(AMQ_ESHAPE, "hidden / inter above 32768 are not staged by the engine");
    return AMQ_OK;
}

int amq_decode_engine_image(const amq_engine_block* blocks, int n_block, int hidden, int inter, int n_heads, int n_kv_heads,
                            int head_dim, int group, void* image) {
    if (!blocks || !image || n_block < 1) return fail(AMQ_EINVAL, "null pointer / no blocks");
    if (int rc = engine_shapes_ok(hidden, inter, n_heads, n_kv_heads, head_dim)) return rc;
    const int kvd = n_kv_heads * 128;
    const int Ns[7] = {hidden, kvd, kvd, hidden, inter, inter, hidden};
    std::vector<amq::EngineLinearH> lin((size_t)n_block * 7);
    std::vector<const void*> ln1(n_block), ln2(n_block);
    std::vector<void*> kc(n_block), vc(n_block);
    for (int b = 0; b < n_block; ++b) {
        for (int i = 0; i < 7; ++i) {
            const amq_engine_linear& l = blocks[b].lin[i];
            const int K = i == 6 ? inter : hidden;
            if (l.N != Ns[i]) return fail(AMQ_ESHAPE, "block %d linear %d: N = %d, expected %d", b, i, l.N, Ns[i]);
            if (int rc = check_shape128(l.bits, l.N, K, group, "the decode engine")) return rc;
            if (int rc = check_mode(l.mode)) return rc;
            if (!l.qweight_native || !l.meta_native) return fail(AMQ_EINVAL, "block %d linear %d: null pointer", b, i);
            if (amq::native_qweight_bytes(l.bits, l.N, K) >= (1ull << 32)) return fail(AMQ_ESHAPE, "block %d linear %d spans 4 GiB or more", b, i);
            lin[(size_t)b * 7 + i] = amq::EngineLinearH{l.qweight_native, l.meta_native, l.N, l.bits, kernel_mode(l.mode)};
        }
        if (!blocks[b].ln1 || !blocks[b].ln2 || !blocks[b].kcache || !blocks[b].vcache) return fail(AMQ_EINVAL, "block %d: null pointer", b);
        ln1[b] = blocks[b].ln1; ln2[b] = blocks[b].ln2; kc[b] = blocks[b].kcache; vc[b] = blocks[b].vcache;
    }
    amq::engine_fill_image(image, n_block, lin.data(), ln1.data(), ln2.data(), kc.data(), vc.data(), hidden, inter);
    return AMQ_OK;
}

int amq_decode_engine_f16(const void* blocks_dev, int n_block, int hidden, int inter, int n_heads, int n_kv_heads, int head_dim,
                          int max_seq, float eps, void* x, void* scratch, size_t scratch_bytes, const void* step_state,
                          void* sync, size_t sync_bytes, int grid, void* stream) {
    if (!blocks_dev || !x || !scratch || !step_state || !sync) return fail(AMQ_EINVAL, "null pointer");
    if (n_block < 1) return fail(AMQ_EINVAL, "no blocks");
    if (int rc = engine_shapes_ok(hidden, inter, n_heads, n_kv_heads, head_dim)) return rc;
    if (max_seq < 1) return fail(AMQ_ESHAPE, "max_seq must be >= 1");
    if (grid < 0 || grid > 1024) return fail(AMQ_EINVAL, "grid must be 0 (one workgroup per CU) or 1..1024");
    if (scratch_bytes < amq::engine_scratch_bytes(hidden, inter, n_kv_heads))
        return fail(AMQ_EINVAL, "scratch too small: need %zu bytes, got %zu", amq::engine_scratch_bytes(hidden, inter, n_kv_heads), scratch_bytes);
    if (sync_bytes < amq::engine_sync_bytes()) return fail(AMQ_EINVAL, "sync workspace too small: need %zu bytes", amq::engine_sync_bytes());
    amq::EngineDesc d{};
    d.blocks_dev = blocks_dev; d.n_block = n_block; d.H = hidden; d.I = inter; d.n_heads = n_heads; d.n_kv_heads = n_kv_heads;
    d.max_seq = max_seq; d.eps = eps; d.x = x; d.scratch = scratch; d.state = step_state; d.sync = sync; d.grid = grid;
    amq::StreamDevice sd_((hipStream_t)stream);
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return fail(AMQ_ELAUNCH, "no HIP device");
    const int P = grid > 0 ? grid : cus;
    if (P > cus) return fail(AMQ_EINVAL, "grid %d exceeds the %d CUs of the device: the workgroups must all be resident", P, cus);
    if (n_heads > P) return fail(AMQ_ESHAPE, "the attention stage needs one workgroup per head (%d heads, %d workgroups)", n_heads, P);
    if (amq::engine_lds_bytes(d, P) > LDS_LIMIT)
        return fail(AMQ_ESHAPE, "engine LDS need (%zu bytes: max_seq %d, hidden %d, inter %d) exceeds %zu", amq::engine_lds_bytes(d, P),
                    max_seq, hidden, inter, LDS_LIMIT);
    return check_hip(amq::launch_decode_engine(d, (hipStream_t)stream), "decode_engine");
}
#endif  // AMQ_AB_ROUTES

int amq_rope_cache_f16(void* q, const void* k, const void* v, void* kcache, void* vcache, const void* rope_table,
                       int rope_rows, int pos0, int S, int n_heads, int n_kv_heads, int head_dim, int max_seq, void* stream) {
    if (!q || !k || !v || !kcache || !vcache || !rope_table) return fail(AMQ_EINVAL, "null pointer");
    if (head_dim != 128) return fail(AMQ_ESHAPE, "head_dim must be 128 (got %d)", head_dim);
    if (S < 1 || n_heads < 1 || n_kv_heads < 1 || rope_rows < 1) return fail(AMQ_ESHAPE, "bad sizes");
    if (pos0 < 0 || pos0 + S > max_seq) return fail(AMQ_ESHAPE, "rows %d..%d do not fit the cache (max_seq %d)", pos0, pos0 + S, max_seq);

    return check_hip(amq::launch_rope_cache(q, k, v, kcache, vcache, rope_table, rope_rows, pos0, S, n_heads, n_kv_heads, max_seq,
                                            (hipStream_t)stream), "rope_cache");
}

int amq_rope_cache_batch_f16(void* q, const void* k, const void* v, void* kcache, void* vcache, const void* rope_table,
                             int rope_rows, int pos0, int S, int batch, int n_heads, int n_kv_heads, int head_dim, int max_seq,
                             void* stream) {
    if (!q || !k || !v || !kcache || !vcache || !rope_table) return fail(AMQ_EINVAL, "null pointer");
    if (head_dim != 128) return fail(AMQ_ESHAPE, "head_dim must be 128 (got %d)", head_dim);
    if (S < 1 || batch < 1 || n_heads < 1 || n_kv_heads < 1 || rope_rows < 1) return fail(AMQ_ESHAPE, "bad sizes");
    if (pos0 < 0 || pos0 + S > max_seq) return fail(AMQ_ESHAPE, "rows %d..%d do not fit the cache (max_seq %d)", pos0, pos0 + S, max_seq);
    if ((long long)S * batch * (n_heads + n_kv_heads) > (1ll << 36)) return fail(AMQ_ESHAPE, "too many rows for one launch");
    return check_hip(amq::launch_rope_cache(q, k, v, kcache, vcache, rope_table, rope_rows, pos0, S, n_heads, n_kv_heads, max_seq,
                                            (hipStream_t)stream, batch), "rope_cache_batch");
}

int amq_rope_rows_f16(void* q, void* k, const void* rope_table, int rope_rows, int pos0, int rows, int seq_len, int n_heads,
                      int n_kv_heads, int head_dim, void* stream) {
    if (!q || !k || !rope_table) return fail(AMQ_EINVAL, "null pointer");
    if (head_dim != 128) return fail(AMQ_ESHAPE, "head_dim must be 128 (got %d)", head_dim);
    if (rows < 1 || seq_len < 1 || (rows % seq_len) != 0 || n_heads < 1 || n_kv_heads < 1 || rope_rows < 1 || pos0 < 0)
        return fail(AMQ_ESHAPE, "bad sizes (rows=%d must be a multiple of seq_len=%d)", rows, seq_len);
    if ((long long)rows * (n_heads + n_kv_heads) > (1ll << 36)) return fail(AMQ_ESHAPE, "rows=%d exceeds one launch", rows);
    return check_hip(amq::launch_rope_rows(q, k, rope_table, rope_rows, pos0, rows, seq_len, n_heads, n_kv_heads,
                                           (hipStream_t)stream), "rope_rows");
}

int amq_silu_mul_f16(const void* gate, const void* up, void* out, size_t n, void* stream) {
    if (!gate || !up || !out) return fail(AMQ_EINVAL, "null pointer");
    if (n == 0 || (n & 7)) return fail(AMQ_ESHAPE, "n must be a positive multiple of 8 (got %zu)", n);
    return check_hip(amq::launch_silu_mul(gate, up, out, (long)n, (hipStream_t)stream), "silu_mul");
}

int amq_linear_f16(int bits, int mode, const void* x, const void* qn, const void* mn, const void* bias, void* y,
                   int M, int N, int K, int group, void* stream) {
    // few rows: weight-streaming GEMV family; otherwise the tiled MFMA GEMM
    if (M <= 8 && amq::gemv_lds_bytes(M, K, 1) <= 64 * 1024)
        return amq_gemv_f16(bits, mode, x, qn, mn, bias, y, M, N, K, group, 0, 0, stream);
    if ((group == 64 || group == 32) && M <= amq::GEMV_MAX_M && amq::gemv_lds_bytes(M, K, 1) <= LDS_LIMIT)     // (more rows: amq_gemm_route_f16 + workspace)
        return amq_gemv_f16(bits, mode, x, qn, mn, bias, y, M, N, K, group, 0, 0, stream);
    return amq_gemm_f16(bits, mode, x, qn, mn, bias, y, M, N, K, group, 0, 0, stream);
}

// ---- bfloat16 variants (amq_bf16.hip) -------------------------------------------------------------------
int amq_dequantize_bf16(int bits, const void* qn, const void* mn, int N, int K, int group, void* W, void* stream) {
    if (int rc = check_shape128(bits, N, K, group, "amq_dequantize_bf16")) return rc;
    if (!qn || !mn || !W) return fail(AMQ_EINVAL, "null pointer");
    return check_hip(amq::launch_dequantize_bf16(bits, qn, mn, N, K, W, (hipStream_t)stream), "dequantize_bf16");
}

int amq_dequantize_hqq_bf16(int bits, const void* W_q, const void* scale, const void* zero, int N, int K, int group, void* W, void* stream) {
    if (int rc = check_shape(bits, N, K, group)) return rc;
    if (!W_q || !scale || !zero || !W) return fail(AMQ_EINVAL, "null pointer");
    return check_hip(amq::launch_dequantize_hqq_bf16(bits, W_q, scale, zero, N, K, W, (hipStream_t)stream, group), "dequantize_hqq_bf16");
}

int amq_gemv_bf16(int bits, const void* x, const void* qn, const void* mn, const void* bias, const void* residual, void* y,
                  int M, int N, int K, int group, int x_stride, int y_stride, void* stream) {
    if (int rc = check_shape128(bits, N, K, group, "amq_gemv_bf16")) return rc;
    if (!x || !qn || !mn || !y) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1 || M > 16) return fail(AMQ_ESHAPE, "amq_gemv_bf16 takes 1 .. 16 rows (got %d); use amq_gemm_bf16", M);
    if (x_stride == 0) x_stride = K;
    if (y_stride == 0) y_stride = N;
    if (x_stride < K || y_stride < N) return fail(AMQ_ESHAPE, "row strides shorter than the rows (x_stride=%d K=%d y_stride=%d N=%d)", x_stride, K, y_stride, N);
    if (x_stride & 7) return fail(AMQ_ESHAPE, "amq_gemv_bf16 reads x in 16-byte pieces: x_stride must be a multiple of 8 elements (got %d)", x_stride);
    return check_hip(amq::launch_gemv_bf16(bits, x, qn, mn, bias, residual, y, M, N, K, x_stride, y_stride, (hipStream_t)stream), "gemv_bf16");
}

size_t amq_gemm_bf16_workspace_bytes(int M, int N, int K) {
    if (M <= 16 || N <= 0 || K <= 0) return 0;
    return (size_t)N * (size_t)K * 2;
}

int amq_gemm_bf16(int bits, const void* x, const void* qn, const void* mn, const void* bias, const void* residual, void* y,
                  int M, int N, int K, int group, int x_stride, int y_stride, void* workspace, size_t workspace_bytes, void* stream) {
    if (M >= 1 && M <= 16) return amq_gemv_bf16(bits, x, qn, mn, bias, residual, y, M, N, K, group, x_stride, y_stride, stream);
    if (int rc = check_shape128(bits, N, K, group, "amq_gemm_bf16")) return rc;
    if (!x || !qn || !mn || !y) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1) return fail(AMQ_ESHAPE, "M must be >= 1 (got %d)", M);
    if (x_stride == 0) x_stride = K;
    if (y_stride == 0) y_stride = N;
    if (x_stride < K || y_stride < N) return fail(AMQ_ESHAPE, "row strides shorter than the rows (x_stride=%d K=%d y_stride=%d N=%d)", x_stride, K, y_stride, N);
    const size_t need = amq_gemm_bf16_workspace_bytes(M, N, K);
    if (!workspace || workspace_bytes < need) return fail(AMQ_EINVAL, "amq_gemm_bf16 workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    if (!amq::gemm_f16w_ok(M, N, K, x_stride, y_stride))
        return fail(AMQ_ESHAPE, "amq_gemm_bf16 beyond 16 rows needs x_stride %% 8 == 0, y_stride %% 4 == 0 and x, W spans below 4 GiB (M=%d N=%d K=%d)", M, N, K);
    if (int rc = check_hip(amq::launch_dequantize_bf16(bits, qn, mn, N, K, workspace, (hipStream_t)stream), "dequantize_bf16")) return rc;
    return check_hip(amq::launch_gemm_bf16w(x, workspace, bias, residual, y, M, N, K, x_stride, y_stride, (hipStream_t)stream), "gemm_bf16");
}

// ---- reference-FFI-shaped entry points ---------------------------------------------------------------
namespace {
struct CompatWs { void* qn; void* mn; void* ytmp; };
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
CompatWs carve(void* ws, int bits, int M, int N, int K) {
    char* p = (char*)ws;
    CompatWs c;
    c.qn = p; p += align256(amq::native_qweight_bytes(bits, N, K));
    c.mn = p; p += align256(amq::native_meta_bytes(N, K, 4));            // (room for the finest group the repack writes: 32)
    c.ytmp = p;
    (void)M;
    return c;
}
}  // namespace

size_t amq_compat_workspace_bytes(int bits, int M, int N, int K) {
    if (N <= 0 || K <= 0 || M <= 0) return 0;
    return align256(amq::native_qweight_bytes(bits, N, K)) + align256(amq::native_meta_bytes(N, K, 4)) + align256((size_t)M * N * 2);
}

int amq_vecquantmatmul_faster_old(int bits, const void* vec, const void* mat, void* mul, const void* scales,
                                  const void* zeros, int groupsize, int vec_height, int batch, int height, int width,
                                  void* workspace, size_t workspace_bytes, int workspace_valid, void* stream) {
    if (bits != 2 && bits != 3 && bits != 4) return fail(AMQ_EINVAL, "bits must be 2, 3 or 4 (got %d)", bits);
    if (height <= 0 || (height * 32) % bits) return fail(AMQ_ESHAPE, "mat height %d is not K/32*bits", height);
    const int K = height * 32 / bits, N = width, M = batch;
    if (int rc = check_shape(bits, N, K, groupsize)) return rc;
    if (vec_height != K / 2) return fail(AMQ_ESHAPE, "vec_height must be K/2 = %d (got %d)", K / 2, vec_height);
    if (!vec || !mat || !mul || !scales || !zeros || !workspace) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1) return fail(AMQ_ESHAPE, "batch must be >= 1");
    if (workspace_bytes < amq_compat_workspace_bytes(bits, M, N, K)) return fail(AMQ_EINVAL, "workspace too small");
    const CompatWs w = carve(workspace, bits, M, N, K);
    if (!workspace_valid)
        if (int rc = amq_repack_from_gptq(bits, mat, scales, zeros, N, K, groupsize, w.qn, w.mn, stream)) return rc;
    if (int rc = amq_linear_f16(bits, AMQ_MODE_FMA, vec, w.qn, w.mn, nullptr, w.ytmp, M, N, K, groupsize, stream)) return rc;
    return check_hip(amq::launch_accumulate_f32(mul, w.ytmp, (size_t)M * N, (hipStream_t)stream), "accumulate");
}

static int compat_awq(const void* x, const void* kernel, const void* scales, const void* scaled_zeros, void* y,
                      int m, int n, int k, int group_size, void* workspace, size_t workspace_bytes, int workspace_valid,
                      void* stream, bool gemm) {
    if (int rc = check_shape(4, n, k, group_size)) return rc;
    if (!x || !kernel || !scales || !scaled_zeros || !y || !workspace) return fail(AMQ_EINVAL, "null pointer");
    if (m < 1) return fail(AMQ_ESHAPE, "m must be >= 1");
    if (workspace_bytes < amq_compat_workspace_bytes(4, 1, n, k)) return fail(AMQ_EINVAL, "workspace too small");
    const CompatWs w = carve(workspace, 4, 1, n, k);
    if (!workspace_valid)
        if (int rc = amq_repack_from_awq(kernel, scales, scaled_zeros, n, k, group_size, w.qn, w.mn, stream)) return rc;
    if (gemm) return amq_gemm_f16(4, AMQ_MODE_FMA, x, w.qn, w.mn, nullptr, y, m, n, k, group_size, 0, 0, stream);
    return amq_linear_f16(4, AMQ_MODE_FMA, x, w.qn, w.mn, nullptr, y, m, n, k, group_size, stream);
}

int amq_gemv_4bit(const void* x, const void* kernel, const void* scales, const void* scaled_zeros, void* y,
                  int m, int n, int k, int group_size, void* workspace, size_t workspace_bytes, int workspace_valid, void* stream) {
    return compat_awq(x, kernel, scales, scaled_zeros, y, m, n, k, group_size, workspace, workspace_bytes, workspace_valid, stream, false);
}

int amq_gemm_4bit(const void* x, const void* kernel, const void* scales, const void* scaled_zeros, void* y,
                  int m, int n, int k, int group_size, void* workspace, size_t workspace_bytes, int workspace_valid, void* stream) {
    return compat_awq(x, kernel, scales, scaled_zeros, y, m, n, k, group_size, workspace, workspace_bytes, workspace_valid, stream, true);
}

int amq_rmsnorm_f16(const void* x, const void* gamma, void* y, int M, int K, float eps, void* stream) {
    if (!x || !gamma || !y) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1 || K < 8 || (K % 8) != 0) return fail(AMQ_ESHAPE, "need M >= 1 and K %% 8 == 0 (got M=%d K=%d)", M, K);
    return check_hip(amq::launch_rmsnorm(x, gamma, y, M, K, eps, (hipStream_t)stream), "rmsnorm");
}

int amq_gemv_f16w(const void* x, const void* W, const void* bias, void* y, const void* gamma, float eps,
                  int N, int K, void* stream) {
    if (!x || !W || !y) return fail(AMQ_EINVAL, "null pointer");
    if (N < 1 || K < 8 || (K % 8) != 0) return fail(AMQ_ESHAPE, "need K %% 8 == 0 (got N=%d K=%d)", N, K);
    if ((size_t)K * 2 + 64 > 64 * 1024) return fail(AMQ_ESHAPE, "K=%d too large for the fp16-weight GEMV", K);
    return check_hip(amq::launch_gemv_f16w(x, W, bias, y, gamma, eps, N, K, (hipStream_t)stream), "gemv_f16w");
}

int amq_gemv_f16w_rows(const void* x, const void* W, const void* bias, void* y, const void* gamma, float eps, int M,
                       int N, int K, void* stream) {
    if (!x || !W || !y) return fail(AMQ_EINVAL, "null pointer");
    if (M < 1 || M > 8) return fail(AMQ_ESHAPE, "M must be 1..8 (got %d)", M);
    if (N < 1 || K < 8 || (K % 8) != 0) return fail(AMQ_ESHAPE, "need N >= 1 and K %% 8 == 0 (got %d, %d)", N, K);
    if ((size_t)M * K * 2 + 64 > LDS_LIMIT) return fail(AMQ_ESHAPE, "%d rows of K=%d do not fit LDS", M, K);
    return check_hip(amq::launch_gemv_f16w(x, W, bias, y, gamma, eps, N, K, (hipStream_t)stream, M), "gemv_f16w_rows");
}

int amq_attn_decode_f16(const void* q, const void* k, const void* v, void* kcache, void* vcache, void* out,
                        const int* pos_dev, int pos, int batch, int n_heads, int n_kv_heads, int head_dim,
                        int max_seq, float rope_theta, const void* rope_table, void* stream) {
    if (!q || !k || !v || !kcache || !vcache || !out) return fail(AMQ_EINVAL, "null pointer");
    if (head_dim != 128) return fail(AMQ_ESHAPE, "head_dim must be 128 (got %d)", head_dim);
    if (batch < 1 || n_heads < 1 || n_heads > 255 || n_kv_heads < 1 || (n_heads % n_kv_heads) != 0)
        return fail(AMQ_ESHAPE, "bad head configuration (%d q heads, %d kv heads)", n_heads, n_kv_heads);
    if (max_seq < 1 || (!pos_dev && (pos < 0 || pos >= max_seq))) return fail(AMQ_ESHAPE, "position %d outside the cache (max_seq=%d)", pos, max_seq);
    if (6 * 128 + (size_t)max_seq * 4 + 17 * 1024 > LDS_LIMIT) return fail(AMQ_ESHAPE, "max_seq=%d too long for the single-pass decode attention", max_seq);
    amq::AttnArgs a{q, k, v, kcache, vcache, out, pos_dev, pos, n_heads, n_kv_heads, max_seq, rope_theta, rope_table, nullptr};
    return check_hip(amq::launch_attn_decode(a, batch, (hipStream_t)stream), "attn_decode");
}

int amq_attn_decode_cur_f16(const void* q, const void* k, const void* v, void* kcache, void* vcache, void* out,
                            const void* step_state, int batch, int n_heads, int n_kv_heads, int head_dim, int max_seq,
                            void* stream) {
    if (!q || !k || !v || !kcache || !vcache || !out || !step_state) return fail(AMQ_EINVAL, "null pointer");
    if (n_heads > 255) return fail(AMQ_ESHAPE, "at most 255 heads");
    if (head_dim != 128) return fail(AMQ_ESHAPE, "head_dim must be 128 (got %d)", head_dim);
    if (batch < 1 || n_heads < 1 || n_kv_heads < 1 || (n_heads % n_kv_heads) != 0)
        return fail(AMQ_ESHAPE, "bad head configuration (%d q heads, %d kv heads)", n_heads, n_kv_heads);
    if (max_seq < 1) return fail(AMQ_ESHAPE, "bad max_seq %d", max_seq);
    if (6 * 128 + (size_t)max_seq * 4 + 17 * 1024 > LDS_LIMIT) return fail(AMQ_ESHAPE, "max_seq=%d too long for the single-pass decode attention", max_seq);
    amq::AttnArgs a{q, k, v, kcache, vcache, out, nullptr, 0, n_heads, n_kv_heads, max_seq, 10000.0f, nullptr, step_state};
    return check_hip(amq::launch_attn_decode(a, batch, (hipStream_t)stream), "attn_decode_cur");
}

size_t amq_attn_decode_split_workspace_bytes(int batch, int n_heads, int n_splits) {
    if (batch < 1 || n_heads < 1 || n_splits < 1) return 0;
    return (size_t)batch * n_heads * n_splits * 132 * sizeof(float);
}

int amq_attn_decode_split_f16(const void* q, const void* k, const void* v, void* kcache, void* vcache, void* out,
                              const void* step_state, const int* pos_dev, int pos, int batch, int n_heads, int n_kv_heads,
                              int head_dim, int max_seq, float rope_theta, const void* rope_table, int n_splits,
                              void* workspace, size_t workspace_bytes, void* tickets, void* stream) {
    if (!q || !k || !v || !kcache || !vcache || !out || !workspace || !tickets) return fail(AMQ_EINVAL, "null pointer");
    if (head_dim != 128) return fail(AMQ_ESHAPE, "head_dim must be 128 (got %d)", head_dim);
    if (batch < 1 || batch > 65535 || n_heads < 1 || n_heads > 255 || n_kv_heads < 1 || (n_heads % n_kv_heads) != 0)
        return fail(AMQ_ESHAPE, "bad head configuration (batch %d, %d q heads, %d kv heads)", batch, n_heads, n_kv_heads);
    if (n_splits < 1 || n_splits > 1024) return fail(AMQ_EINVAL, "n_splits must be 1..1024 (got %d)", n_splits);
    if (max_seq < 1 || (!step_state && !pos_dev && (pos < 0 || pos >= max_seq)))
        return fail(AMQ_ESHAPE, "position %d outside the cache (max_seq=%d)", pos, max_seq);
    int chunk = (((max_seq + n_splits - 1) / n_splits) + 31) & ~31;
    chunk = chunk < amq::ATT_MIN_CHUNK ? amq::ATT_MIN_CHUNK : chunk;
    if (6 * 128 + (size_t)chunk * 4 + 17 * 1024 > LDS_LIMIT)
        return fail(AMQ_ESHAPE, "max_seq=%d over %d splits leaves chunks of %d keys: too long", max_seq, n_splits, chunk);
    const size_t need = amq_attn_decode_split_workspace_bytes(batch, n_heads, n_splits);
    if (workspace_bytes < need) return fail(AMQ_EINVAL, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    amq::AttnArgs a{q, k, v, kcache, vcache, out, step_state ? nullptr : pos_dev, pos, n_heads, n_kv_heads, max_seq,
                    step_state ? 10000.0f : rope_theta, step_state ? nullptr : rope_table, step_state};
    return check_hip(amq::launch_attn_decode_split(a, batch, n_splits, workspace, tickets, (hipStream_t)stream), "attn_decode_split");
}

static int amq_attn_prefill_check(const void* q, const void* k, const void* v, const void* out, int batch, int S, int pos0, int n_heads,
                                  int n_kv_heads, int head_dim, long long q_rstride, long long q_bstride, long long k_rstride,
                                  long long k_bstride, long long k_hstride, long long v_rstride, long long v_bstride, long long v_hstride,
                                  long long o_rstride, long long o_bstride) {
    if (!q || !k || !v || !out) return fail(AMQ_EINVAL, "null pointer");
    if (head_dim != 128) return fail(AMQ_ESHAPE, "head_dim must be 128 (got %d)", head_dim);
    if (batch < 1 || batch > 65535 || S < 1 || pos0 < 0 || n_heads < 1 || n_heads > 65535 || n_kv_heads < 1 || (n_heads % n_kv_heads) != 0)
        return fail(AMQ_ESHAPE, "bad sizes (batch %d, S %d, pos0 %d, %d q heads, %d kv heads)", batch, S, pos0, n_heads, n_kv_heads);
    const long long strides[] = {q_rstride, q_bstride, k_rstride, k_bstride, k_hstride, v_rstride, v_bstride, v_hstride, o_rstride, o_bstride};
    for (long long sv : strides)
        if (sv < 0 || (sv % 8) != 0) return fail(AMQ_ESHAPE, "strides must be non-negative multiples of 8 halves (16-byte vector access)");
    if (q_rstride < (long long)n_heads * 128 || o_rstride < (long long)n_heads * 128 || k_rstride < 128 || v_rstride < 128)
        return fail(AMQ_ESHAPE, "row strides smaller than the rows they separate");
    // the kernel forms key * row stride in bytes as a 24 x 24 -> 32-bit product (byte offset inside one sequence's k / v)
    const long long keys = (long long)pos0 + S;
    if (2 * k_rstride >= (1 << 24) || 2 * v_rstride >= (1 << 24) || keys >= (1 << 24) || 2 * keys * k_rstride >= (1ll << 32) || 2 * keys * v_rstride >= (1ll << 32))
        return fail(AMQ_ESHAPE, "k / v of one sequence must span fewer than 2^32 bytes (keys %lld, row strides %lld / %lld halves)", keys, k_rstride, v_rstride);
    return AMQ_OK;
}

int amq_attn_prefill_f16(const void* q, const void* k, const void* v, void* out, int batch, int S, int pos0, int n_heads,
                         int n_kv_heads, int head_dim, long long q_rstride, long long q_bstride, long long k_rstride,
                         long long k_bstride, long long k_hstride, long long v_rstride, long long v_bstride, long long v_hstride,
                         long long o_rstride, long long o_bstride, void* stream) {
    if (int rc = amq_attn_prefill_check(q, k, v, out, batch, S, pos0, n_heads, n_kv_heads, head_dim, q_rstride, q_bstride, k_rstride, k_bstride,
                                        k_hstride, v_rstride, v_bstride, v_hstride, o_rstride, o_bstride)) return rc;
    amq::AttnPrefillArgs a{q, k, v, out, S, pos0, n_heads, n_kv_heads, batch, (long)q_rstride, (long)q_bstride, (long)k_rstride,
                           (long)k_bstride, (long)k_hstride, (long)v_rstride, (long)v_bstride, (long)v_hstride, (long)o_rstride,
                           (long)o_bstride, 0};
    return check_hip(amq::launch_attn_prefill(a, (hipStream_t)stream), "attn_prefill");
}

int amq_attn_prefill_xfrag_f16(const void* q, const void* k, const void* v, void* out_xf, int S, int pos0, int n_heads,
                               int n_kv_heads, int head_dim, long long q_rstride, long long k_rstride, long long k_hstride,
                               long long v_rstride, long long v_hstride, void* stream) {
    if (int rc = amq_attn_prefill_check(q, k, v, out_xf, 1, S, pos0, n_heads, n_kv_heads, head_dim, q_rstride, 0, k_rstride, 0, k_hstride,
                                        v_rstride, 0, v_hstride, (long long)n_heads * 128, 0)) return rc;
    amq::AttnPrefillArgs a{q, k, v, out_xf, S, pos0, n_heads, n_kv_heads, 1, (long)q_rstride, 0, (long)k_rstride, 0, (long)k_hstride,
                           (long)v_rstride, 0, (long)v_hstride, 0, 0, 1};
    return check_hip(amq::launch_attn_prefill(a, (hipStream_t)stream), "attn_prefill_xfrag");
}

int amq_decode_tail_f16(const void* logits, int vocab, const void* embed, int hidden, long long* token, int* pos, void* x,
                        const void* rope_table, void* rope_cur, int rope_rows, void* stream) {
    if (!logits || !embed || !token || !pos || !x) return fail(AMQ_EINVAL, "null pointer");
    if ((rope_table == nullptr) != (rope_cur == nullptr)) return fail(AMQ_EINVAL, "rope_table and rope_cur go together");
    if (rope_cur && rope_rows < 1) return fail(AMQ_EINVAL, "rope_rows must be the number of rows of rope_table");
    if (vocab < 1 || hidden < 8 || (hidden % 8) != 0) return fail(AMQ_ESHAPE, "need vocab >= 1 and hidden %% 8 == 0 (got %d, %d)", vocab, hidden);
    return check_hip(amq::launch_decode_tail(logits, vocab, embed, hidden, token, pos, x, rope_table, rope_cur, rope_rows, (hipStream_t)stream), "decode_tail");
}

int amq_decode_tail_batch_f16(const void* logits, int vocab, const void* embed, int hidden, long long* token, int* pos, void* x,
                              const void* rope_table, void* rope_cur, int rope_rows, int batch, void* stream) {
    if (!logits || !embed || !token || !pos || !x) return fail(AMQ_EINVAL, "null pointer");
    if ((rope_table == nullptr) != (rope_cur == nullptr)) return fail(AMQ_EINVAL, "rope_table and rope_cur go together");
    if (rope_cur && rope_rows < 1) return fail(AMQ_EINVAL, "rope_rows must be the number of rows of rope_table");
    if (vocab < 1 || hidden < 8 || (hidden % 8) != 0) return fail(AMQ_ESHAPE, "need vocab >= 1 and hidden %% 8 == 0 (got %d, %d)", vocab, hidden);
    if (batch < 1 || batch > 65535) return fail(AMQ_ESHAPE, "bad batch %d", batch);
    if (batch > 1 && (vocab % 8) != 0) return fail(AMQ_ESHAPE, "batched rows need vocab %% 8 == 0 (16-byte aligned logits rows)");
    return check_hip(amq::launch_decode_tail(logits, vocab, embed, hidden, token, pos, x, rope_table, rope_cur, rope_rows, (hipStream_t)stream, batch), "decode_tail_batch");
}

int amq_decode_tail_suppress_f16(const void* logits, int vocab, const void* embed, int hidden, long long* token, int* pos, void* x,
                                 const void* rope_table, void* rope_cur, int rope_rows, int batch, const int* suppress_ids, void* stream) {
    if (!logits || !embed || !token || !pos || !x) return fail(AMQ_EINVAL, "null pointer");
    if (!suppress_ids) return fail(AMQ_EINVAL, "suppress_ids: a device array of 8 int32 token ids (-1 = unused slot) is required");
    if ((rope_table == nullptr) != (rope_cur == nullptr)) return fail(AMQ_EINVAL, "rope_table and rope_cur go together");
    if (rope_cur && rope_rows < 1) return fail(AMQ_EINVAL, "rope_rows must be the number of rows of rope_table");
    if (vocab < 1 || hidden < 8 || (hidden % 8) != 0) return fail(AMQ_ESHAPE, "need vocab >= 1 and hidden %% 8 == 0 (got %d, %d)", vocab, hidden);
    if (batch < 1 || batch > 65535) return fail(AMQ_ESHAPE, "bad batch %d", batch);
    if (batch > 1 && (vocab % 8) != 0) return fail(AMQ_ESHAPE, "batched rows need vocab %% 8 == 0 (16-byte aligned logits rows)");
    return check_hip(amq::launch_decode_tail(logits, vocab, embed, hidden, token, pos, x, rope_table, rope_cur, rope_rows, (hipStream_t)stream, batch,
                                             suppress_ids), "decode_tail_suppress");
}

int amq_set_token_f16(const long long* token_in, int n_in, const void* embed, int vocab, int hidden, long long* token, const int* pos, void* x,
                      const void* rope_table, void* rope_cur, int rope_rows, int batch, void* stream) {
    if (!token_in || !embed || !token || !pos || !x) return fail(AMQ_EINVAL, "null pointer");
    if ((rope_table == nullptr) != (rope_cur == nullptr)) return fail(AMQ_EINVAL, "rope_table and rope_cur go together");
    if (rope_cur && rope_rows < 1) return fail(AMQ_EINVAL, "rope_rows must be the number of rows of rope_table");
    if (vocab < 1 || hidden < 8 || (hidden % 8) != 0) return fail(AMQ_ESHAPE, "need vocab >= 1 and hidden %% 8 == 0 (got %d, %d)", vocab, hidden);
    if (batch < 1 || batch > 65535 || (n_in != 1 && n_in != batch)) return fail(AMQ_ESHAPE, "batch %d with %d input ids (1 or one per sequence)", batch, n_in);
    return check_hip(amq::launch_set_token(token_in, n_in, embed, vocab, hidden, token, pos, x, rope_table, rope_cur, rope_rows, batch, (hipStream_t)stream), "set_token");
}

int amq_rope_table_f16(void* table, int max_seq, float rope_theta, void* stream) {
    if (!table || max_seq < 1) return fail(AMQ_EINVAL, "bad rope table request");
    return check_hip(amq::launch_rope_table(table, max_seq, rope_theta, (hipStream_t)stream), "rope_table");
}

int amq_rope_table_freqs_f16(void* table, int max_seq, const float* inv_freq, float scale, void* stream) {
    if (!table || !inv_freq || max_seq < 1) return fail(AMQ_EINVAL, "bad rope table request");
    return check_hip(amq::launch_rope_table_freqs(table, max_seq, inv_freq, scale, (hipStream_t)stream), "rope_table_freqs");
}

}  // extern "C"
