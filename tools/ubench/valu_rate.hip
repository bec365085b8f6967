// Instruction issue-rate probe for gfx950: cycles per wave-instruction for the
// ops the dequant path is built from, at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP16(X) X X X X X X X X X X X X X X X X
#define LOOPS 256

#define KERNEL(NAME, ASM)                                                          \
__global__ void NAME(uint64_t* out, uint32_t seed) {                               \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;       \
    uint32_t b0 = seed ^ 0x3c003c00u, b1 = 0x3c003c00u, c0 = 0x00070007u;          \
    float f0 = 1.0f, f1 = 2.0f, f2 = 3.f, f3 = 4.f;                                \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                    \
    for (int i = 0; i < LOOPS; ++i) {                                              \
        asm volatile(REP16(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) \
                     : "v"(b0), "v"(b1), "s"(c0));                                 \
    }                                                                              \
    uint64_t t1 = __builtin_amdgcn_s_memtime();                                    \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    if (a0 + a1 + a2 + a3 == 0x12345 && f0 + f1 + f2 + f3 == 1.2345f) out[0] = 0;  \
}

// 4 independent chains per group so dependency latency does not limit issue
KERNEL(k_pk_add,  "v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_add_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n")
KERNEL(k_pk_mul,  "v_pk_mul_f16 %0, %0, %9\n v_pk_mul_f16 %1, %1, %9\n v_pk_mul_f16 %2, %2, %9\n v_pk_mul_f16 %3, %3, %9\n")
KERNEL(k_pk_fma,  "v_pk_fma_f16 %0, %0, %9, %8\n v_pk_fma_f16 %1, %1, %9, %8\n v_pk_fma_f16 %2, %2, %9, %8\n v_pk_fma_f16 %3, %3, %9, %8\n")
KERNEL(k_dot2c,   "v_dot2c_f32_f16 %4, %0, %8\n v_dot2c_f32_f16 %5, %1, %8\n v_dot2c_f32_f16 %6, %2, %8\n v_dot2c_f32_f16 %7, %3, %8\n")
KERNEL(k_dot2,    "v_dot2_f32_f16 %4, %0, %8, %4\n v_dot2_f32_f16 %5, %1, %8, %5\n v_dot2_f32_f16 %6, %2, %8, %6\n v_dot2_f32_f16 %7, %3, %8, %7\n")
KERNEL(k_and_or,  "v_and_or_b32 %0, %0, %10, %8\n v_and_or_b32 %1, %1, %10, %8\n v_and_or_b32 %2, %2, %10, %8\n v_and_or_b32 %3, %3, %10, %8\n")
KERNEL(k_and,     "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n")
KERNEL(k_and_lit, "v_and_b32 %0, 0x3c003c0, %0\n v_and_b32 %1, 0x3c003c0, %1\n v_and_b32 %2, 0x3c003c0, %2\n v_and_b32 %3, 0x3c003c0, %3\n")
KERNEL(k_and_sgpr,"v_and_b32 %0, %10, %0\n v_and_b32 %1, %10, %1\n v_and_b32 %2, %10, %2\n v_and_b32 %3, %10, %3\n")
KERNEL(k_pkmul_sgpr,"v_pk_mul_f16 %0, %0, %10 op_sel_hi:[1,0]\n v_pk_mul_f16 %1, %1, %10 op_sel_hi:[1,0]\n v_pk_mul_f16 %2, %2, %10 op_sel_hi:[1,0]\n v_pk_mul_f16 %3, %3, %10 op_sel_hi:[1,0]\n")
KERNEL(k_perm,    "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n")
KERNEL(k_mad24,   "v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n")
KERNEL(k_lshr,    "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n")
KERNEL(k_fma32,   "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %4\n v_fma_f32 %7, %7, %4, %5\n")
KERNEL(k_fmac32,  "v_fmac_f32 %4, %0, %8\n v_fmac_f32 %5, %1, %8\n v_fmac_f32 %6, %2, %8\n v_fmac_f32 %7, %3, %8\n")
KERNEL(k_add16,   "v_add_f16 %0, %0, %8\n v_add_f16 %1, %1, %8\n v_add_f16 %2, %2, %8\n v_add_f16 %3, %3, %8\n")
KERNEL(k_fmamix,  "v_fma_mix_f32 %4, %0, %8, %4 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %5, %1, %8, %5 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %6, %2, %8, %6 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %7, %3, %8, %7 op_sel_hi:[1,1,0]\n")
KERNEL(k_cvtu16,  "v_cvt_f16_u16 %0, %0\n v_cvt_f16_u16 %1, %1\n v_cvt_f16_u16 %2, %2\n v_cvt_f16_u16 %3, %3\n")
KERNEL(k_bfe,     "v_bfe_u32 %0, %0, 4, 4\n v_bfe_u32 %1, %1, 4, 4\n v_bfe_u32 %2, %2, 4, 4\n v_bfe_u32 %3, %3, 4, 4\n")
KERNEL(k_lshlor,  "v_lshl_or_b32 %0, %0, 4, %8\n v_lshl_or_b32 %1, %1, 4, %8\n v_lshl_or_b32 %2, %2, 4, %8\n v_lshl_or_b32 %3, %3, 4, %8\n")
KERNEL(k_pkmad16, "v_pk_mad_u16 %0, %0, %8, %9\n v_pk_mad_u16 %1, %1, %8, %9\n v_pk_mad_u16 %2, %2, %8, %9\n v_pk_mad_u16 %3, %3, %8, %9\n")
KERNEL(k_pkmul16, "v_pk_mul_lo_u16 %0, %0, %8\n v_pk_mul_lo_u16 %1, %1, %8\n v_pk_mul_lo_u16 %2, %2, %8\n v_pk_mul_lo_u16 %3, %3, %8\n")

__global__ void k_mfma(uint64_t* out, uint32_t seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)((seed + threadIdx.x + i) & 7); b[i] = (_Float16)1.0f; }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < LOOPS; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (c0[0] + c1[1] + c2[2] + c3[3] == 1.2345f) out[0] = 0;
}

__global__ void k_mfma_dep(uint64_t* out, uint32_t seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)((seed + threadIdx.x + i) & 7); b[i] = (_Float16)1.0f; }
    f4 c0 = {0, 0, 0, 0};
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < LOOPS; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (c0[0] == 1.2345f) out[0] = 0;
}
typedef void (*kern_t)(uint64_t*, uint32_t);
struct Case { const char* name; kern_t k; int per_loop; };

int main() {
    uint64_t* d; hipMalloc(&d, 1 << 20);
    std::vector<uint64_t> h(4096);
    Case cases[] = {
        {"v_pk_add_f16", k_pk_add, 64}, {"v_pk_mul_f16", k_pk_mul, 64}, {"v_pk_fma_f16", k_pk_fma, 64},
        {"v_dot2c_f32_f16", k_dot2c, 64}, {"v_dot2_f32_f16", k_dot2, 64}, {"v_and_or_b32", k_and_or, 64},
        {"v_and_b32", k_and, 64}, {"v_and_b32 literal", k_and_lit, 64}, {"v_and_b32 sgpr", k_and_sgpr, 64}, {"v_pk_mul_f16 sgpr opsel", k_pkmul_sgpr, 64}, {"v_perm_b32", k_perm, 64}, {"v_mad_u32_u24", k_mad24, 64}, {"v_lshrrev_b32", k_lshr, 64},
        {"v_fma_f32", k_fma32, 64}, {"v_fmac_f32", k_fmac32, 64}, {"v_add_f16", k_add16, 64}, {"v_fma_mix_f32", k_fmamix, 64},
        {"v_cvt_f16_u16", k_cvtu16, 64}, {"v_bfe_u32", k_bfe, 64}, {"v_lshl_or_b32", k_lshlor, 64},
        {"v_pk_mad_u16", k_pkmad16, 64}, {"v_pk_mul_lo_u16", k_pkmul16, 64},
        {"mfma_16x16x32_f16", k_mfma, 16}, {"mfma_16x16x32 dependent", k_mfma_dep, 16},
    };
    printf("%-20s %10s %10s %10s   (cycles per wave-instruction per SIMD; s_memtime ticks)\n", "op", "1w/SIMD", "2w/SIMD", "4w/SIMD");
    for (auto& c : cases) {
        printf("%-20s", c.name);
        for (int wps : {1, 2, 4}) {
            int threads = 64 * 4 * wps;      // one workgroup on one CU: wps waves on each of 4 SIMDs
            hipMemset(d, 0, 1 << 20);
            hipLaunchKernelGGL(c.k, dim3(1), dim3(threads), 0, 0, d, 12345u);
            hipLaunchKernelGGL(c.k, dim3(1), dim3(threads), 0, 0, d, 12345u);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 8 * (threads / 64), hipMemcpyDeviceToHost);
            double mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
            // per SIMD: wps waves each issue LOOPS*per_loop instructions
            printf(" %10.2f", mx / (double)(LOOPS * c.per_loop * wps));
        }
        printf("\n");
    }
    return 0;
}
