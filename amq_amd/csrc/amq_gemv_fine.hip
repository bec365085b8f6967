// amq_gemv_fine.hip -- the GEMV kernels for groups of 64 / 32 (two / four (scale, zero) pairs per tile), every prologue (amq_gemv_body.cuh)
#include "amq_gemv_body.cuh"
namespace amq {
#define AMQ_INST(PRO_)                                                                               \
    template hipError_t launch_pro_g<PRO_, 2>(const GemvKArgs&, int, int, size_t, hipStream_t);       \
    template hipError_t launch_pro_g<PRO_, 4>(const GemvKArgs&, int, int, size_t, hipStream_t);
AMQ_INST(PRO_NONE)
AMQ_INST(PRO_RMSNORM)
AMQ_INST(PRO_SILU_MUL)
#undef AMQ_INST
}
