// What does ONE wave per SIMD pay for an instruction placed between its own back-to-back MFMAs?  (the question behind a one-wave-per-SIMD
// GEMM loop: amq_gemm_f16.hip's four-wave experiment.)  256 threads = 4 waves, one workgroup per CU (LDS), a loop of 64 independent
// v_mfma_f32_16x16x32_f16 (16 cycles each) with one extra operation after every EVERY-th MFMA:
//   none | ds_read_b128 | LDS-DMA piece (s_mov m0 + s_nop + buffer_load_dwordx4 ... lds) from an L2-resident or an HBM-streamed source |
//   the piece without its M0 write | buffer_load_dwordx4 into VGPRs | 3 SALU fillers | global_load_lds_dwordx4
// Prints shader cycles per MFMA (s_memtime around the loop, wave 0 of every workgroup: mean and max), for 1 workgroup and for 256.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_shadow mfma_shadow.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <utility>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int i4 __attribute__((ext_vector_type(4)));

enum { M_NONE = 0, M_DSREAD = 1, M_PIECE = 2, M_PIECE_NOM0 = 3, M_VLOAD = 4, M_PIECE_HBM = 5, M_SALU = 6, M_READ_AND_PIECE = 7, M_GLOBAL_LDS = 8 };

template <int... I, class F> __device__ __forceinline__ void sfor(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

template <int MODE, int EVERY>
__global__ __launch_bounds__(256) void k(const unsigned char* src, unsigned span_mask, int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    f4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    h8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    const unsigned long long base = (unsigned long long)src;
    i4 rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
    rsrc.z = -1;
    rsrc.w = 0x00020000;
    const unsigned voff = (unsigned)lane * 16u + (unsigned)wave * 1024u;
    unsigned soff = ((unsigned)blockIdx.x * 65536u) & span_mask;            // per-workgroup start; advances 4 KiB per operation
    const unsigned lds_w = lds0 + wave * 1024;
    const unsigned rd_addr = (unsigned)lane * 16u + (unsigned)wave * 1024u;
    i4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0}, r3 = {0, 0, 0, 0};
    constexpr int OPS = (64 + EVERY - 1) / EVERY;                           // operations per loop iteration
    constexpr int KEEP = OPS < 60 ? OPS : 60;

    auto ds_read = [&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        if constexpr ((n & 3) == 0) { i4& q_ = r0; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q_) : "v"(rd_addr), "n"((n & 7) * 4096)); }
        if constexpr ((n & 3) == 1) { i4& q_ = r1; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q_) : "v"(rd_addr), "n"((n & 7) * 4096)); }
        if constexpr ((n & 3) == 2) { i4& q_ = r2; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q_) : "v"(rd_addr), "n"((n & 7) * 4096)); }
        if constexpr ((n & 3) == 3) { i4& q_ = r3; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q_) : "v"(rd_addr), "n"((n & 7) * 4096)); }
    };
    auto piece = [&](auto n_c, bool m0) {
        constexpr int n = decltype(n_c)::value;
        const unsigned dst = lds_w + 32768u + (unsigned)(n & 7) * 4096u;
        if (m0) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(rsrc), "s"(soff), "s"(dst) : "memory", "m0");
        else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(rsrc), "s"(soff) : "memory");
        soff = (soff + 4096u) & span_mask;
    };

    // M0 for the no-M0 form
    asm volatile("s_mov_b32 m0, %0" :: "s"(lds_w + 32768u) : "m0");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        sfor(std::make_integer_sequence<int, 64>{}, [&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            f4& c = acc[i & 15];
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
            if constexpr ((i % EVERY) == 0) {
                constexpr int n = i / EVERY;
                using N = std::integral_constant<int, n>;
                if constexpr (MODE == M_DSREAD) ds_read(N{});
                if constexpr (MODE == M_PIECE || MODE == M_PIECE_HBM) piece(N{}, true);
                if constexpr (MODE == M_PIECE_NOM0) piece(N{}, false);
                if constexpr (MODE == M_READ_AND_PIECE) { ds_read(N{}); piece(N{}, true); }
                if constexpr (MODE == M_SALU) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0");
                if constexpr (MODE == M_VLOAD) {
                    i4& q_ = (n & 1) ? r1 : r0;
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(q_) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
                    soff = (soff + 4096u) & span_mask;
                }
                if constexpr (MODE == M_GLOBAL_LDS) {
                    const unsigned dst = lds_w + 32768u + (unsigned)(n & 7) * 4096u;
                    const unsigned char* p = src + soff + voff;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory", "m0");
                    soff = (soff + 4096u) & span_mask;
                }
            }
        });
        // the previous iteration's operations have completed (this iteration's stay in flight)
        if constexpr (MODE == M_DSREAD || MODE == M_READ_AND_PIECE) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(KEEP < 15 ? KEEP : 15) : "memory");
        if constexpr (MODE == M_PIECE || MODE == M_PIECE_NOM0 || MODE == M_VLOAD || MODE == M_PIECE_HBM || MODE == M_READ_AND_PIECE || MODE == M_GLOBAL_LDS)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += (float)(r0.x ^ r1.x ^ r2.x ^ r3.x);
    if (s == 1.2345f) sink[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int EVERY>
void run(const char* name, const unsigned char* src, size_t span, int grid, unsigned long long* cyc, float* sink) {
    const int iters = 2000;
    static bool attr = false;
    (void)attr;
    hipFuncSetAttribute((const void*)k<MODE, EVERY>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, EVERY>), dim3(grid), dim3(256), 100 << 10, 0, src, (unsigned)(span - 1), iters, cyc, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double mean = 0; unsigned long long mx = 0;
    for (int i = 0; i < grid; ++i) { mean += (double)h[i] / grid; if (h[i] > mx) mx = h[i]; }
    const double per = mean / (iters * 64.0), permax = (double)mx / (iters * 64.0);
    constexpr int OPS = (64 + EVERY - 1) / EVERY;
    printf("%-44s every %2d  grid %3d  cycles per MFMA %6.2f (max %6.2f)   extra per operation %6.1f cycles\n", name, EVERY, grid, per, permax,
           MODE == M_NONE ? 0.0 : (per - 16.0) * 64.0 / OPS);
}

#define SWEEP(MODE, NAME, SRC, SPAN)                                                          \
    for (int grid : {1, 256}) {                                                               \
        run<MODE, 1>(NAME, SRC, SPAN, grid, cyc, sink); run<MODE, 2>(NAME, SRC, SPAN, grid, cyc, sink); \
        run<MODE, 4>(NAME, SRC, SPAN, grid, cyc, sink); run<MODE, 8>(NAME, SRC, SPAN, grid, cyc, sink); \
    }

int main() {
    unsigned char* src; unsigned long long* cyc; float* sink;
    const size_t big = 1ull << 30;
    hipMalloc(&src, big); hipMemset(src, 0, big); hipMalloc(&cyc, 8 * 256); hipMalloc(&sink, 4);
    run<M_NONE, 64>("none", src, 1 << 20, 1, cyc, sink);
    run<M_NONE, 64>("none", src, 1 << 20, 256, cyc, sink);
    SWEEP(M_SALU, "3 x s_nop", src, 1 << 20)
    SWEEP(M_DSREAD, "ds_read_b128", src, 1 << 20)
    SWEEP(M_PIECE, "LDS-DMA piece, 1 MiB source (L2)", src, 1 << 20)
    SWEEP(M_PIECE_NOM0, "  ... without the M0 write", src, 1 << 20)
    SWEEP(M_GLOBAL_LDS, "global_load_lds_dwordx4, 1 MiB source", src, 1 << 20)
    SWEEP(M_VLOAD, "buffer_load_dwordx4 into VGPRs, 1 MiB source", src, 1 << 20)
    SWEEP(M_PIECE_HBM, "LDS-DMA piece, 1 GiB source (HBM)", src, big)
    SWEEP(M_READ_AND_PIECE, "ds_read_b128 + piece, 1 MiB source", src, 1 << 20)
    return 0;
}
