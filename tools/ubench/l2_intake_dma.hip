// Per-CU intake of L2-resident data every workgroup reads ONCE (each wave its own K tiles: no line is re-read inside a CU -- the x of a few-row GEMM),
// register loads against LDS-DMA: W waves per workgroup, each wave streams its share of `bytes` with U 1 KiB transfers in flight,
//   mode 0: global_load_dwordx4 into registers (xor-consumed)      mode 1: global_load_lds_dwordx4 into a per-wave LDS ring of U KiB, read back with ds_read_b128
// hipcc --offload-arch=gfx950 -O3 -o l2_intake_dma l2_intake_dma.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

template <int U, int MODE>
__global__ __launch_bounds__(512) void k(const u4* __restrict__ src, uint32_t* out, int pieces, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    u4 acc = {0, 0, 0, 0};
    const int nt = (pieces - wave + nw - 1) / nw;          // pieces wave, wave + nw, ...
    if (MODE == 0) {
        u4 buf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) buf[u] = src[(size_t)(wave + (u < nt ? u : 0) * nw) * 64 + lane];
        for (int i0 = 0; i0 + U <= nt; i0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc ^= buf[u];
                const int nx = i0 + u + U < nt ? i0 + u + U : nt - 1;
                buf[u] = src[(size_t)(wave + nx * nw) * 64 + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= buf[u];
    } else {
        unsigned char* ring = smem + wave * (U * 1024);
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
#pragma unroll
        for (int u = 0; u < U; ++u) glds16(src, (unsigned)((wave + (u < nt ? u : 0) * nw) * 1024 + lane * 16), lds0 + u * 1024);
        for (int i0 = 0; i0 + U <= nt; i0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(U - 1) : "memory");
                acc ^= *(const u4*)(ring + u * 1024 + lane * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int nx = i0 + u + U < nt ? i0 + u + U : nt - 1;
                glds16(src, (unsigned)((wave + nx * nw) * 1024 + lane * 16), lds0 + u * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

template <int U, int MODE>
void run(const u4* src, uint32_t* out, unsigned long long* cyc, int grid, int waves, size_t bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int pieces = (int)(bytes / 1024);
    const size_t lds = MODE ? (size_t)waves * U * 1024 : 0;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k<U, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<U, MODE>), dim3(grid), dim3(waves * 64), lds, 0, src, out, pieces, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<U, MODE>), dim3(grid), dim3(waves * 64), lds, 0, src, out, pieces, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; double mean = 0;
    for (int i = 0; i < grid; ++i) { mean += (double)h[i] / grid; if (h[i] > mx) mx = h[i]; }
    printf("%s grid %4d  waves %2d  U %2d  %5.0f KiB per workgroup: %7.2f us per launch; in-kernel %8.0f cycles mean (%8llu max) = %5.1f B/clk per workgroup\n",
           MODE ? "lds-dma " : "register", grid, waves, U, bytes / 1024.0, ms * 1e3 / it, mean, mx, (double)bytes / mean);
}

int main() {
    const size_t bytes = 512 << 10;
    u4* src; uint32_t* out; unsigned long long* cyc;
    hipMalloc(&src, 4 << 20); hipMemset(src, 1, 4 << 20); hipMalloc(&out, 4); hipMalloc(&cyc, 8 * 1024);
    for (int grid : {1, 64, 256}) {
        run<4, 0>(src, out, cyc, grid, 8, bytes);
        run<8, 0>(src, out, cyc, grid, 8, bytes);
        run<16, 0>(src, out, cyc, grid, 8, bytes);
        run<4, 1>(src, out, cyc, grid, 8, bytes);
        run<8, 1>(src, out, cyc, grid, 8, bytes);
        run<16, 1>(src, out, cyc, grid, 8, bytes);
    }
    return 0;
}
