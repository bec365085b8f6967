// grid_barrier2.hip -- second look at a device-wide barrier for a persistent decode kernel (first look: grid_barrier.hip,
// profiles/r01b_grid_barrier.txt: 3.8 us for 256 workgroups on one counter, +2 us for the agent-scope fence pair, +3.5 us
// until another XCD's data was readable).  Two changes, both suggested by round-2 measurements:
//   * arrivals spread over NG counters (group = workgroup id % NG, i.e. most likely one XCD each) + one top-level counter:
//     one address takes ~13 ns per arrival, eight take them in parallel;
//   * no cache-wide fences: the exchanged data are agent-scope relaxed atomic stores / loads (write-through / L2-bypassing
//     per ACCESS), ordered by vmcnt + the barrier's own atomics.  (A device-scope __threadfence() per workgroup writes
//     back and invalidates the whole L2: profiles/r02_skinny_split_negative.txt.)
// Every poll loop is bounded (error flag + leave).  build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier2 grid_barrier2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr unsigned SPIN_LIMIT = 1u << 20;
constexpr int PAD = 64;                      // counters 256 bytes apart

// NG = 0: flat (one counter).  GEN_PER_GROUP: pollers of group g poll gen[g] instead of one word.  WORK: floats per thread exchanged.
template <int NG, int GEN_PER_GROUP, int WORK>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* ctr, unsigned* gen, unsigned* err, float* data, int iters) {
    const unsigned nwg = gridDim.x;
    const unsigned g = NG ? blockIdx.x % NG : 0;
    const unsigned gsize = NG ? (nwg - g + NG - 1) / NG : nwg;         // workgroups in my group
    float v = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (WORK) {
            float* p = &data[((size_t)(it & 1) * nwg + blockIdx.x) * 256 + threadIdx.x];
            __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                  // vmcnt(0): the stores have completed
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned)(it + 1);
            bool released = false;
            if (NG) {
                const unsigned prev = __hip_atomic_fetch_add(ctr + (1 + g) * PAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prev == gsize * target - 1) {
                    const unsigned top = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (top == (unsigned)NG * target - 1) {
                        if (GEN_PER_GROUP) {
                            for (int i = 0; i < NG; ++i) __hip_atomic_store(gen + i * PAD, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            __hip_atomic_store(gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        released = true;
                    }
                }
            } else {
                const unsigned prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prev == nwg * target - 1) {
                    __hip_atomic_store(gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    released = true;
                }
            }
            if (!released) {
                unsigned* const mine = gen + (GEN_PER_GROUP ? g * PAD : 0);
                unsigned spins = 0;
                while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > SPIN_LIMIT) { *err = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        __syncthreads();
        if (WORK) {
            const unsigned nb = (blockIdx.x + 97) % nwg;                        // a workgroup on (most likely) another XCD
            v = __hip_atomic_load(&data[((size_t)(it & 1) * nwg + nb) * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1.0f;
        }
    }
    if (WORK) data[((size_t)2 * nwg + blockIdx.x) * 256 + threadIdx.x] = v;
}

template <int NG, int GPG, int WORK>
static void run(const char* name, int nwg, int iters) {
    unsigned *ctr, *gen, *err;
    float* data;
    CHECK(hipMalloc(&ctr, 65 * PAD * 4)); CHECK(hipMalloc(&gen, 65 * PAD * 4)); CHECK(hipMalloc(&err, 256));
    CHECK(hipMalloc(&data, (size_t)3 * nwg * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemset(ctr, 0, 65 * PAD * 4)); CHECK(hipMemset(gen, 0, 65 * PAD * 4)); CHECK(hipMemset(err, 0, 256));
        CHECK(hipMemset(data, 0, (size_t)3 * nwg * 256 * 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((barrier_kernel<NG, GPG, WORK>), dim3(nwg), dim3(256), 0, 0, ctr, gen, err, data, iters);
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
    printf("%-52s wg %4d  %.3f us per barrier  %s%s\n", name, nwg, best * 1e3 / iters, herr ? "SPIN LIMIT HIT " : "", ok ? "" : "DATA MISMATCH (stale read)");
    fflush(stdout);
    CHECK(hipFree(ctr)); CHECK(hipFree(gen)); CHECK(hipFree(err)); CHECK(hipFree(data));
}

int main() {
    const int iters = 2000;
    for (int nwg : {256, 512, 768}) {
        run<0, 0, 0>("flat, one counter", nwg, iters);
        run<8, 0, 0>("8 group counters, one generation word", nwg, iters);
        run<8, 1, 0>("8 group counters, 8 generation words", nwg, iters);
        run<32, 1, 0>("32 group counters, 32 generation words", nwg, iters);
        run<8, 1, 1>("8 + 8, data exchanged by agent-scope accesses", nwg, iters);
        run<32, 1, 1>("32 + 32, data exchanged by agent-scope accesses", nwg, iters);
    }
    return 0;
}
