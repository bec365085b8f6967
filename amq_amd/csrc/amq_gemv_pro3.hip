// amq_gemv_pro3.hip -- the 2 .. 8-row GEMV kernels whose RMSNorm takes its sums of squares as per-row-tile partials from the launch that produced x
// (PRO_RMSNORM_SUMS; amq_gemv_body.cuh, amq_gemv_grouped_sums_f16): RS = 64 (2 .. 4 rows) / 128 (5 .. 8), default arithmetic and geometry, 8 or 16 waves
#include "amq_gemv_body.cuh"
namespace amq {
hipError_t launch_pro_sums_entry(const GemvKArgs& a, int nw, int rs, int total_wg, size_t lds, hipStream_t st) {
    if (rs == 64) {
        if (nw == 16) return launch_one<PRO_RMSNORM_SUMS, 16, 2, MATH_EXACT, XCfg<16>::XC, 64>(a, total_wg, lds, st);
        return launch_one<PRO_RMSNORM_SUMS, 8, 2, MATH_EXACT, XCfg<8>::XC, 64>(a, total_wg, lds, st);
    }
    if (nw == 16) return launch_one<PRO_RMSNORM_SUMS, 16, 2, MATH_EXACT, XCfg<16>::XC, 128>(a, total_wg, lds, st);
    return launch_one<PRO_RMSNORM_SUMS, 8, 2, MATH_EXACT, XCfg<8>::XC, 128>(a, total_wg, lds, st);
}
}
