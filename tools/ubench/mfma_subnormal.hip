// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs (default kernel mode)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, uint16_t bbits, uint16_t abits) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, abits); b[i] = __builtin_bit_cast(_Float16, bbits); }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 256); float h[64];
    struct { uint16_t b, a; const char* what; double expect; } cs[] = {
        {0x0003, 0x3C00, "B = 3*2^-24 (subnormal), A = 1.0", 32 * 3.0 / 16777216.0},
        {0x03C0, 0x3C00, "B = 960*2^-24 (subnormal), A = 1.0", 32 * 960.0 / 16777216.0},
        {0x0003, 0x0100, "B = 3*2^-24, A = 256*2^-24 (both subnormal)", 32 * 3.0 * 256.0 / 16777216.0 / 16777216.0},
        {0x3C00, 0x0010, "B = 1.0, A = 16*2^-24 (subnormal)", 32 * 16.0 / 16777216.0},
    };
    for (auto& c : cs) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c.b, c.a);
        hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("%-48s got %.9e expect %.9e %s\n", c.what, h[0], c.expect, (double)h[0] == c.expect ? "EXACT" : "DIFF");
    }
    return 0;
}
