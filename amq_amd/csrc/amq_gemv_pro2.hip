// amq_gemv_pro2.hip -- the GEMV kernels with prologue PRO_SILU_MUL, groups of 128 (amq_gemv_body.cuh)
#include "amq_gemv_body.cuh"
namespace amq {
template hipError_t launch_pro<PRO_SILU_MUL>(const GemvKArgs&, int, int, int, int, size_t, hipStream_t);
}
