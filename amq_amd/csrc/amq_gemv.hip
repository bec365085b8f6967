// amq_gemv.hip -- host side of the weight-streaming GEMV: launch plan (waves, row-tiles per workgroup, segment ranges) and dispatch to the
// kernel instantiations, which live in one translation unit per fused prologue (amq_gemv_pro{0,1,2}.hip, amq_gemv_fine.hip; kernels and
// launch templates: amq_gemv_body.cuh).
#include "amq_gemv_body.cuh"

namespace amq {


#ifdef AMQ_STAMP
unsigned long long* g_stamp_ptr = nullptr;
extern "C" int amq_debug_set_stamps(void* p) { g_stamp_ptr = (unsigned long long*)p; return 0; }
#endif

// nw: waves per workgroup the launch will use (sizes the double-buffered cross-wave sum); <= 1: the largest (16), a safe bound
size_t gemv_lds_bytes(int M, int K, int nw) {
    const size_t xbytes = (((size_t)M * (K + XPAD) * 2) + 15) & ~(size_t)15;
    const size_t xg = (size_t)(K >> 7) * 64;
    const size_t red = (size_t)2 * (nw > 1 ? nw : 16) * 16 * 16 * 4;
    return xbytes + xg + red;
}

// the <= 8-row kernels (RS = 128): half the cross-wave sum buffer, no per-group sums
size_t gemv_lds_bytes_rows(int M, int K, int nw) {
    const size_t xbytes = (((size_t)M * (K + XPAD) * 2) + 15) & ~(size_t)15;
    return xbytes + (size_t)2 * nw * (M <= 4 ? 64 : 128) * 4;
}
// 5 .. 8 rows whose x does not fit LDS whole (the 7B down_proj at 7 - 8 rows: 8 x 11008 halves = 172 KB): x in two K phases, one 16-wave workgroup
// per row-tile.  Needs no full-row statistic in the prologue (norm = false), an even tile count, >= 2 tiles per wave and phase, one chunk per thread
// and phase.
bool gemv_rows_phased(int M, int K, bool plain, bool norm) {
    const int G = K >> 7;
    return plain && !norm && M > 4 && M <= 8 && (G & 1) == 0 && G / 2 >= 2 * 16 && (K / 2 >> 3) <= 1024 &&
           gemv_lds_bytes_rows(M, K, 16) > 160 * 1024 && gemv_lds_bytes_rows(M, K / 2, 16) <= 160 * 1024;
}
// smallest LDS allocation some launch form of an M-row GEMV needs (plain: default arithmetic and geometry, groups of 128)
size_t gemv_min_lds_bytes(int M, int K, bool plain, bool norm) {
    if (gemv_rows_phased(M, K, plain, norm)) return gemv_lds_bytes_rows(M, K / 2, 16);
    if (plain && M > 1 && M <= 8) return gemv_lds_bytes_rows(M, K, 16);
    return gemv_lds_bytes(M, K, 1);
}

int gemv_pick_waves(int total_rt, int K) {
    // measured on MI355X (tools/sweep1.sh, profiles/r01b_gemv_sweep.txt): 8 waves x 2 tiles in flight is the best or
    // within 3% of it whenever a wave gets >= 4 tiles of a row-tile or there is more than one workgroup per CU; one
    // 16-wave workgroup per CU wins (3-6%) only for long rows (K >= 8192: down_proj) on <= 256 row-tiles; for K = 4096
    // on 256 row-tiles (o_proj) 8 waves are ~10% faster (16-wave workgroups dispatch later); tiny K: 4 waves.
    // K > 4096 always takes 16 waves: the register-held activation staging of the decode prologue covers
    // K <= 8 * XC * threads (4096 for 8 waves, 16384 for 16), and one 16-wave workgroup per CU is within 5% of the
    // 8-wave grids on every shape measured.  (Exception made in launch_gemv: single-row launches with 4096 < K <= 8192
    // go to 8-wave workgroups that stage two chunks per thread, two workgroups per CU.)
    const int G = K >> 7;
    if (G < 16) return 4;
    if (G > 32) return 16;
    return 8;
}

// defined (explicitly instantiated) in amq_gemv_pro{0,1,2}.hip / amq_gemv_fine.hip
#define AMQ_GEMV_EXTERN(PRO_)                                                                                                  \
    extern template hipError_t launch_pro<PRO_>(const GemvKArgs&, int, int, int, int, size_t, hipStream_t);                     \
    extern template hipError_t launch_pro_g<PRO_, 2>(const GemvKArgs&, int, int, size_t, hipStream_t);                          \
    extern template hipError_t launch_pro_g<PRO_, 4>(const GemvKArgs&, int, int, size_t, hipStream_t);
AMQ_GEMV_EXTERN(PRO_NONE)
AMQ_GEMV_EXTERN(PRO_RMSNORM)
AMQ_GEMV_EXTERN(PRO_SILU_MUL)
#undef AMQ_GEMV_EXTERN
hipError_t launch_pro_sums_entry(const GemvKArgs&, int nw, int rs, int, size_t, hipStream_t);     // amq_gemv_pro3.hip

// Fills the per-segment workgroup ranges and launches.  rpt = row-tiles per workgroup.
#ifndef AMQ_RS128_THREE
#define AMQ_RS128_THREE 1
#endif
hipError_t launch_gemv(GemvArgs& a, hipStream_t st) {
    StreamDevice sd_(st);                                  // kernel attributes are per device: the stream's, not the current one
    int total_rt = 0;
    for (int i = 0; i < a.nseg; ++i) { a.seg[i].n_rt = a.seg[i].N / 16; total_rt += a.seg[i].n_rt; }
    int nw = a.force_waves ? a.force_waves : gemv_pick_waves(total_rt, a.K);
    const int gp = a.gp > 1 ? a.gp : 1;                    // meta pairs per tile (groups of 64 / 32: 2 / 4)
    if (gp != 1 && gp != 2 && gp != 4) return hipErrorInvalidValue;
    if (gp > 1 && ((a.flags & (GEMV_FLAG_DOT | GEMV_FLAG_LINEAR)) || a.force_depth == 4)) return hipErrorInvalidValue;   // (the C ABI says so first)
    // Launches of 2 .. 8 rows (sequences decoded together) take the row kernels: x by LDS-DMA ahead of the weight ring, cross-wave sum buffer sized
    // for their rows (RS = 64: 2 .. 4 rows, 78 - 80 VGPRs, three 8-wave workgroups per CU while their LDS fits; RS = 128: 5 .. 8 rows, two per CU;
    // one 16-wave workgroup where not even two fit, e.g. K = 11008).  7 - 8 rows of a K whose rows do not fit LDS whole: two K phases (ph2).
    // Measured against these: rows held in registers, one 16-wave workgroup per CU throughout (profiles/r05_decode_batch.txt).  9 .. 16 rows,
    // A/B options and groups of 64 / 32 keep the generic staging (RS = 256), 16 waves once three 8-wave workgroups no longer fit.
    bool rs128 = false, rs64 = false;
    const bool plain = !(a.flags & (GEMV_FLAG_DOT | GEMV_FLAG_LINEAR)) && a.force_depth != 4 && gp == 1;
    const bool ph2 = !a.force_waves && a.force_rpt <= 0 && gemv_rows_phased(a.M, a.K, plain, a.prologue == PRO_RMSNORM || a.x_stride != a.K);      // (the phased kernel takes dense rows only: LDS-DMA staging)
    if (ph2) {
        rs128 = true;
        nw = 16;
    } else if (a.M > 1 && a.M <= 8 && plain && nw != 4) {
        rs128 = a.M > 4;
        rs64 = !rs128;
        if (!a.force_waves && nw == 8 && 2 * gemv_lds_bytes_rows(a.M, a.K, 8) > 160 * 1024) nw = 16;
    } else if (!a.force_waves && a.M > 1 && nw == 8 && 3 * gemv_lds_bytes(a.M, a.K, 8) > 160 * 1024) {
        nw = 16;
    }
    // 4096 < K <= 8192 at one row (13B / 70B hidden sizes): 8-wave workgroups staging two x chunks per thread, two per CU
    // (~90 VGPRs), instead of one 16-wave workgroup -- 13B 464 -> 485 tokens/s, 70B 126.5 -> 133.  The same trade for
    // 8192 < K <= 16384 (four chunks per thread) loses (13B 484 -> 467; 7B's K = 11008 with three chunks 808 -> 773), as do
    // three workgroups per CU (449 / 122).
    const bool mid_k = gp == 1 && !a.force_waves && nw == 16 && a.M == 1 && (a.K >> 3) > 512 && (a.K >> 3) <= 1024 &&
                       !(a.flags & (GEMV_FLAG_DOT | GEMV_FLAG_LINEAR)) && a.force_depth != 4;    // (only the exact-math body has the two-chunk variant)
    if (mid_k) nw = 8;
    // persistent-style grid: about 24 waves per CU (256 CUs) -- three 8-wave workgroups, but ONE 16-wave workgroup (two do
    // not fit the register file at 78 VGPRs, a second round of workgroups would run on an empty chip); a workgroup walks
    // `rpt` row-tiles
    int rpt = ph2 ? 1 : a.force_rpt;                       // (K phases: one row-tile per workgroup -- its accumulators live across the phases)
    // (the 5 .. 8-row kernels with a fused RMSNorm hold 105 - 128 VGPRs: two 8-wave workgroups per CU; without it 77 - 80, and three where their LDS fits -- 5 rows of K = 4096)
    const bool two_per_cu = nw == 8 && ((rs128 && (a.prologue == PRO_RMSNORM || a.prologue == PRO_RMSNORM_SUMS || !AMQ_RS128_THREE)) || ((rs64 || rs128) && 3 * gemv_lds_bytes_rows(a.M, a.K, 8) > 160 * 1024));
    const int target = (mid_k || two_per_cu) ? 512 : nw == 16 ? 256 : 256 * 24 / nw;
    if (rpt <= 0) {
        rpt = (total_rt + target - 1) / target;
        if (rpt < 1) rpt = 1;
    }
    // several workgroups per CU that each hold a large x (the 5 .. 8-row kernels: two per CU): the launch's 512 workgroups are dealt to the segments in
    // proportion to their row-tiles, so that every CU ends up with the same load -- q/k/v of 7B at 8 rows: 3 x 170 workgroups of 2 or 1 row-tiles
    // (3 per CU) instead of 3 x 128 of 2 (4 on half of the CUs, 2 on the others; profiles/r05_decode_batch.txt)
    // ... and the one-row launches of 4096 < K <= 8192 (two 8-wave workgroups per CU as well): 70B q/k/v = 640 row-tiles ran as 320 workgroups of 2 (4 on a
    // quarter of the CUs), dealt 512 of 2 or 1: 13.4 -> 11.0 us per launch, 70B 135.0 -> 136.8 tokens/s.  Every launch dealt this way (A/B: scope 2): 7B -0.4 %.
#ifndef AMQ_DEAL_SCOPE
#define AMQ_DEAL_SCOPE 1
#endif
    const bool deal = a.force_rpt <= 0 && !ph2 && total_rt > target &&
                      ((rs128 && nw == 8) || (AMQ_DEAL_SCOPE >= 1 && mid_k) || AMQ_DEAL_SCOPE >= 2);
    int wg = 0, mask = 0;
    for (int i = 0; i < a.nseg; ++i) {
        a.seg[i].wg_begin = wg;
        a.seg[i].wg_count = (a.seg[i].n_rt + rpt - 1) / rpt;
        if (deal) {
            const int share = (int)((long long)target * a.seg[i].n_rt / total_rt);
            a.seg[i].wg_count = share < 1 ? 1 : share > a.seg[i].n_rt ? a.seg[i].n_rt : share;
        }
        if (a.seg[i].n_rt / a.seg[i].wg_count > GEMV_SPLIT_BASE_MASK || a.seg[i].wg_count >= (1 << 19)) return hipErrorInvalidValue;      // (N beyond 50 M rows)
        wg += a.seg[i].wg_count;
        mask |= 1 << a.seg[i].bits;
    }
    const bool lin = (a.flags & GEMV_FLAG_LINEAR) && !((a.flags & GEMV_FLAG_DOT) && a.M == 1);
    (void)lin; (void)mask;
    const size_t lds = ph2 ? gemv_lds_bytes_rows(a.M, a.K / 2, 16) : (rs128 || rs64) ? gemv_lds_bytes_rows(a.M, a.K, nw) : gemv_lds_bytes(a.M, a.K, a.M > 1 ? nw : 16);     // (one-row launches keep the allocation they were tuned with)
    // the partial-sum forms exist in the 2 .. 8-row kernels only (amq_gemv_grouped_sums_f16 checks the same and says why)
    if ((a.prologue == PRO_RMSNORM_SUMS || a.sums_out) && !(rs128 || rs64)) return hipErrorInvalidValue;
    if (a.prologue == PRO_RMSNORM_SUMS && (ph2 || !a.sums_in || (a.K >> 4) > 64 * SUMS_PER_LANE || a.x_stride != a.K)) return hipErrorInvalidValue;
    if (a.sums_out && (a.nseg != 1 || a.prologue != PRO_NONE)) return hipErrorInvalidValue;
    GemvKArgs k{};
    k.x = a.x; k.x2 = a.prologue == PRO_RMSNORM_SUMS ? a.sums_in : a.x2; k.gamma = a.gamma;
    k.sums_out = (float*)a.sums_out; k.sums_stride = a.seg[0].n_rt;
    k.M = a.M; k.K = a.K; k.x_stride = a.x_stride; k.nseg = a.nseg;
    k.eps = a.eps; k.rpt = rpt;
    for (int i = 0; i < a.nseg; ++i) {
        const GemvSeg& s = a.seg[i];
        // (the one-op unpack of MODE_FMA1 exists in the exact-math, group-128 bodies; elsewhere such buffers run as MODE_FMA: same results)
        const bool has_fma1 = gp == 1 && !(a.flags & (GEMV_FLAG_DOT | GEMV_FLAG_LINEAR));
        k.wg_begin[i] = s.wg_begin; k.n_rt[i] = gemv_split(s.n_rt, s.wg_count); k.key[i] = s.bits * 4 + (s.mode == MODE_FMA1 && !has_fma1 ? (int)MODE_FMA : s.mode);
        k.qweight[i] = s.qweight; k.meta[i] = s.meta;
        k.bias[i] = s.bias; k.residual[i] = s.residual; k.y[i] = s.y; k.y_stride[i] = s.y_stride;
    }
#ifdef AMQ_STAMP
    k.stamps = g_stamp_ptr;
#endif
    if (gp == 2) {
        switch (a.prologue) {
            case PRO_NONE: return launch_pro_g<PRO_NONE, 2>(k, nw, wg, lds, st);
            case PRO_RMSNORM: return launch_pro_g<PRO_RMSNORM, 2>(k, nw, wg, lds, st);
            default: return launch_pro_g<PRO_SILU_MUL, 2>(k, nw, wg, lds, st);
        }
    }
    if (gp == 4) {
        switch (a.prologue) {
            case PRO_NONE: return launch_pro_g<PRO_NONE, 4>(k, nw, wg, lds, st);
            case PRO_RMSNORM: return launch_pro_g<PRO_RMSNORM, 4>(k, nw, wg, lds, st);
            default: return launch_pro_g<PRO_SILU_MUL, 4>(k, nw, wg, lds, st);
        }
    }
    if (a.prologue == PRO_RMSNORM_SUMS) return launch_pro_sums_entry(k, nw, rs64 ? 64 : 128, wg, lds, st);
    switch (a.prologue) {
        case PRO_NONE: return launch_pro<PRO_NONE>(k, a.flags | (rs128 ? GEMV_FLAG_RS128 : 0) | (rs64 ? GEMV_FLAG_RS64 : 0) | (ph2 ? GEMV_FLAG_PH2 : 0), a.force_depth, nw, wg, lds, st);
        case PRO_RMSNORM: return launch_pro<PRO_RMSNORM>(k, a.flags | (rs128 ? GEMV_FLAG_RS128 : 0) | (rs64 ? GEMV_FLAG_RS64 : 0) | (ph2 ? GEMV_FLAG_PH2 : 0), a.force_depth, nw, wg, lds, st);
        default: return launch_pro<PRO_SILU_MUL>(k, a.flags | (rs128 ? GEMV_FLAG_RS128 : 0) | (rs64 ? GEMV_FLAG_RS64 : 0) | (ph2 ? GEMV_FLAG_PH2 : 0), a.force_depth, nw, wg, lds, st);
    }
}

}  // namespace amq
