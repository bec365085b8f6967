// grid_barrier.hip -- what does a device-wide barrier between dependent stages cost on MI355X, compared with the
// ~1.3 us kernel boundary of a hipGraph replay?  (Input for a persistent per-token decode kernel: DESIGN.md section 7.)
//
// All workgroups are co-resident (<= 1 per CU x mult); a barrier = every workgroup's thread 0 does
//   release fence -> atomicAdd(counter) -> last arriver publishes the generation -> others poll it -> acquire fence.
// The poll is bounded: a workgroup that does not see the generation after SPIN_LIMIT polls sets an error flag and leaves.
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ; run: ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr unsigned SPIN_LIMIT = 4u << 20;

template <int FENCE, int WORK>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* ctr, unsigned* gen, unsigned* err, float* data, int iters) {
    const unsigned nwg = gridDim.x;
    float v = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (WORK) {   // a token amount of dependent global traffic per stage: write own slot, read a neighbour's after the barrier
            data[((size_t)(it & 1) * nwg + blockIdx.x) * 256 + threadIdx.x] = v;       // double-buffered by parity
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FENCE) __atomic_thread_fence(__ATOMIC_RELEASE);                 // agent scope by default for HIP device code
            const unsigned prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1);
            if (prev == nwg * target - 1) {
                __hip_atomic_store(gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                unsigned spins = 0;
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > SPIN_LIMIT) { *err = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (FENCE) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        if (WORK) {
            const unsigned nb = (blockIdx.x + 97) % nwg;                        // a workgroup on (most likely) another XCD
            v = __builtin_nontemporal_load(&data[((size_t)(it & 1) * nwg + nb) * 256 + threadIdx.x]) + 1.0f;
        }
    }
    if (WORK) data[((size_t)2 * nwg + blockIdx.x) * 256 + threadIdx.x] = v;
}

template <int FENCE, int WORK>
static void run(const char* name, int nwg, int iters) {
    unsigned *ctr, *gen, *err;
    float* data;
    CHECK(hipMalloc(&ctr, 256)); CHECK(hipMalloc(&gen, 256)); CHECK(hipMalloc(&err, 256));
    CHECK(hipMalloc(&data, (size_t)3 * nwg * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemset(ctr, 0, 256)); CHECK(hipMemset(gen, 0, 256)); CHECK(hipMemset(err, 0, 256));
        CHECK(hipMemset(data, 0, (size_t)3 * nwg * 256 * 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((barrier_kernel<FENCE, WORK>), dim3(nwg), dim3(256), 0, 0, ctr, gen, err, data, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    unsigned herr = 0; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<float> h((size_t)nwg * 256);
    CHECK(hipMemcpy(h.data(), data + (size_t)2 * nwg * 256, h.size() * 4, hipMemcpyDeviceToHost));
    bool ok = true;
    if (WORK) for (size_t i = 0; i < h.size(); ++i) if (h[i] != (float)(i % 256) + iters) { ok = false; break; }
    printf("%-34s wg %4d  %.3f us per barrier  %s%s\n", name, nwg, best * 1e3 / iters, herr ? "SPIN LIMIT HIT " : "", ok ? "" : "DATA MISMATCH (stale read)");
    CHECK(hipFree(ctr)); CHECK(hipFree(gen)); CHECK(hipFree(err)); CHECK(hipFree(data));
}

int main() {
    const int iters = 2000;
    for (int nwg : {256, 512, 768}) {
        run<0, 0>("atomics only", nwg, iters);
        run<1, 0>("release/acquire fences", nwg, iters);
        run<1, 1>("fences + dependent traffic", nwg, iters);
    }
    return 0;
}
