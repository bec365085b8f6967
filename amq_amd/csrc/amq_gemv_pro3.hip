// amq_gemv_pro3.hip -- the 5 .. 8-row GEMV kernels whose RMSNorm takes its sums of squares as per-row-tile partials from the launch that produced x
// (PRO_RMSNORM_SUMS; amq_gemv_body.cuh, amq_gemv_grouped_sums_f16): RS = 128, default arithmetic and geometry, 8 or 16 waves
#include "amq_gemv_body.cuh"
namespace amq {
hipError_t launch_pro_sums_entry(const GemvKArgs& a, int nw, int total_wg, size_t lds, hipStream_t st) {
    if (nw == 16) return launch_one<PRO_RMSNORM_SUMS, 16, 2, MATH_EXACT, XCfg<16>::XC, 128>(a, total_wg, lds, st);
    return launch_one<PRO_RMSNORM_SUMS, 8, 2, MATH_EXACT, XCfg<8>::XC, 128>(a, total_wg, lds, st);
}
}
