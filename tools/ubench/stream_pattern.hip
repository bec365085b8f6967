// Pure weight-streaming access-pattern probe (no compute): what HBM rate does a
// given workgroup->address mapping reach on gfx950?
//   mode 0: "rowtile"  WG b streams chunk b (CH bytes contiguous); its NW waves interleave 1 KiB pieces, U in flight
//   mode 1: "linear"   classic grid-stride: piece index = (iter*gridDim + b)*NW + wave   (compact frontier)
//   mode 2: "persist"  grid = 256*k WGs, WG walks chunks b, b+grid, ... each streamed as in mode 0
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int NW, int U, bool NT>
__global__ __launch_bounds__(NW * 64) void k_stream(const u4* __restrict__ src, uint32_t* out, int mode, int pieces_per_chunk, int nchunks) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u4 acc = {0, 0, 0, 0};
    auto ld = [&](size_t piece) { const u4* p = src + piece * 64 + lane; return NT ? __builtin_nontemporal_load(p) : *p; };
    if (mode == 1) {
        const size_t total = (size_t)nchunks * pieces_per_chunk;
        const size_t stride = (size_t)gridDim.x * NW;
        size_t i = (size_t)blockIdx.x * NW + wave;
        u4 buf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) buf[u] = (i + u * stride < total) ? ld(i + u * stride) : acc;
        for (; i < total; i += U * stride) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc ^= buf[u];
                const size_t nx = i + (size_t)(u + U) * stride;
                if (nx < total) buf[u] = ld(nx);
            }
        }
    } else {
        for (int chunk = blockIdx.x; chunk < nchunks; chunk += (mode == 2 ? gridDim.x : nchunks)) {
            const size_t base = (size_t)chunk * pieces_per_chunk;
            const int nt = (pieces_per_chunk - wave + NW - 1) / NW;
            u4 buf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) if (u < nt) buf[u] = ld(base + wave + u * NW);
            for (int i0 = 0; i0 < nt; i0 += U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (i0 + u < nt) {
                        acc ^= buf[u];
                        if (i0 + u + U < nt) buf[u] = ld(base + wave + (size_t)(i0 + u + U) * NW);
                    }
                }
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int NW, int U, bool NT>
double run(const u4* src, uint32_t* out, int mode, int grid, int ppc, int nchunks, size_t total_bytes, int nbuf, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t stride_u4 = total_bytes / 16;
    for (int i = 0; i < nbuf; ++i) hipLaunchKernelGGL((k_stream<NW, U, NT>), dim3(grid), dim3(NW * 64), 0, 0, src + i * stride_u4, out, mode, ppc, nchunks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_stream<NW, U, NT>), dim3(grid), dim3(NW * 64), 0, 0, src + (i % nbuf) * stride_u4, out, mode, ppc, nchunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
    const size_t layer = 48ull << 20;           // one "layer": 48 MiB (22016 x 4096 @ 4 bit)
    const int nbuf = 16;                        // rotate 768 MiB
    u4* src; uint32_t* out;
    hipMalloc(&src, layer * nbuf); hipMalloc(&out, 4);
    hipMemset(src, 1, layer * nbuf);
    const int iters = 64;
    printf("%-44s %9s %9s\n", "pattern", "us", "TB/s");
    auto rep = [&](const char* name, double us) { printf("%-44s %9.2f %9.3f\n", name, us, layer / us / 1e6); };
    const int chunk = 32 << 10, ppc = chunk / 1024, nchunks = (int)(layer / chunk);
    rep("rowtile 32KB/WG NW4 U4 nt (current gemv)", run<4, 4, true>(src, out, 0, nchunks, ppc, nchunks, layer, nbuf, iters));
    rep("rowtile 32KB/WG NW8 U4 nt", run<8, 4, true>(src, out, 0, nchunks, ppc, nchunks, layer, nbuf, iters));
    rep("rowtile 32KB/WG NW4 U8 nt", run<4, 8, true>(src, out, 0, nchunks, ppc, nchunks, layer, nbuf, iters));
    rep("rowtile 32KB/WG NW4 U4 plain", run<4, 4, false>(src, out, 0, nchunks, ppc, nchunks, layer, nbuf, iters));
    rep("linear grid=1024 NW4 U4 nt", run<4, 4, true>(src, out, 1, 1024, ppc, nchunks, layer, nbuf, iters));
    rep("linear grid=2048 NW4 U4 nt", run<4, 4, true>(src, out, 1, 2048, ppc, nchunks, layer, nbuf, iters));
    rep("linear grid=1024 NW4 U8 nt", run<4, 8, true>(src, out, 1, 1024, ppc, nchunks, layer, nbuf, iters));
    rep("linear grid=512 NW8 U8 nt", run<8, 8, true>(src, out, 1, 512, ppc, nchunks, layer, nbuf, iters));
    rep("linear grid=1024 NW4 U4 plain", run<4, 4, false>(src, out, 1, 1024, ppc, nchunks, layer, nbuf, iters));
    rep("linear grid=4096 NW4 U2 nt", run<4, 2, true>(src, out, 1, 4096, ppc, nchunks, layer, nbuf, iters));
    rep("persist grid=512 32KB chunks NW4 U4 nt", run<4, 4, true>(src, out, 2, 512, ppc, nchunks, layer, nbuf, iters));
    rep("persist grid=1024 32KB chunks NW4 U4 nt", run<4, 4, true>(src, out, 2, 1024, ppc, nchunks, layer, nbuf, iters));
    rep("persist grid=256 32KB chunks NW8 U4 nt", run<8, 4, true>(src, out, 2, 256, ppc, nchunks, layer, nbuf, iters));
    rep("persist grid=512 32KB chunks NW8 U8 nt", run<8, 8, true>(src, out, 2, 512, ppc, nchunks, layer, nbuf, iters));
    // small layer: 8 MiB (4096 x 4096 @ 4 bit)
    const size_t small = 8ull << 20;
    auto rep2 = [&](const char* name, double us) { printf("%-44s %9.2f %9.3f\n", name, us, small / us / 1e6); };
    const int nch2 = (int)(small / chunk);
    rep2("8MiB rowtile NW16 U2 nt", run<16, 2, true>(src, out, 0, nch2, ppc, nch2, small, 64, iters));
    rep2("8MiB rowtile NW8 U4 nt", run<8, 4, true>(src, out, 0, nch2, ppc, nch2, small, 64, iters));
    rep2("8MiB linear grid=512 NW4 U4 nt", run<4, 4, true>(src, out, 1, 512, ppc, nch2, small, 64, iters));
    rep2("8MiB linear grid=1024 NW4 U2 nt", run<4, 2, true>(src, out, 1, 1024, ppc, nch2, small, 64, iters));
    rep2("8MiB linear grid=2048 NW4 U1 nt", run<4, 1, true>(src, out, 1, 2048, ppc, nch2, small, 64, iters));
    return 0;
}
