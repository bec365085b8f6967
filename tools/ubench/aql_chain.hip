// aql_chain.hip -- can DEPENDENT launches overlap on gfx950?  (VERDICT r3 item 1, steps a and b.)
//
// HIP sets the AQL barrier bit on every kernel packet of a stream, so packet k+1 is not even looked at before packet k has
// drained, and every boundary pays ~1.4 us + the successor's launch ramp.  This program submits the same kernels
//   (0) through a HIP stream (control),
//   (1) through hipExtLaunchKernelGGL(..., flags = hipExtAnyOrderLaunch)    (hip_ext.h:67 says: unsupported on GFX9xx),
//   (2) through its OWN HSA user queue: code object loaded with the HSA loader, AQL KERNEL_DISPATCH packets written by hand,
//       header barrier bit 0 / 1, acquire + release fence scope NONE / AGENT, one doorbell for the whole batch,
// and measures with s_memrealtime stamps (100 MHz) taken by the kernels themselves:
//   a. start-to-start and end-to-start distance of two trivial 768-workgroup kernels;
//   b. the SAFETY property a flag-ordered chain needs -- "a queue launches ALL workgroups of packet k before the first of
//      packet k+1" -- kernel A has 4x the chip's resident capacity, every A workgroup bumps `started` at entry; every B workgroup
//      records the value it sees at ITS entry (min over B); B then spins (bounded, error word) until all of A has finished.
//      min == A's grid for every trial <=> in-order launch; and if a chip-filling B ever got in ahead of A's tail the bounded
//      spin would trip.
//   c. a chain of N kernels with a body of T us each, ordered (i) by the queue barrier, (ii) by in-memory epoch counters only
//      (wait AFTER a prologue that stands for weight priming): the per-stage cost of both.
//
// build:  hipcc --offload-arch=gfx950 -O3 -o aql_chain aql_chain.hip -lhsa-runtime64
//         hipcc --offload-arch=gfx950 -O3 --cuda-device-only --no-gpu-bundle-output -o aql_chain.hsaco aql_chain.hip
// run:    ./aql_chain [aql_chain.hsaco]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define HCHECK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); printf("HSA error %s (0x%x) at line %d\n", m_ ? m_ : "?", (unsigned)s_, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------------------------- kernels
struct StageArgs {
    unsigned long long* stamps;   // [slot][workgroup][2]: entry, exit (reduced on the host: 3072 same-address atomics are 40 us)
    unsigned* counters;           // per stage 80 words of 256 B: [0..31] arrival groups (workgroup % 32), [32] top, [40..71] done words
                                  // (one flat counter costs 768 x 13 ns = 10 us of same-address atomics per stage: first run of this file)
    unsigned* err;
    int slot;                     // stage index
    int wait_on;                  // stage index whose counter to wait for (-1: none)
    unsigned wait_target;         // value that counter must reach
    int body_ticks;               // busy time of the body in 10 ns ticks
    int pro_ticks;                // prologue (stands for weight priming) before the wait
    int signal;                   // 1: drain + bump counters[64 * slot] once per workgroup
    int nwg;                      // grid size (gridDim would need the hidden kernarg block a hand-written packet does not carry)
};

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

__device__ __forceinline__ void spin_ticks(int ticks) {
    const unsigned long long t0 = now();
    while ((long long)(now() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
}

extern "C" __global__ __launch_bounds__(512) void stage_kernel(StageArgs a) {
    const unsigned long long t_in = now();
    if (a.pro_ticks) spin_ticks(a.pro_ticks);
    if (a.wait_on >= 0) {
        if (threadIdx.x == 0) {
            unsigned* c = a.counters + 64 * (80 * (size_t)a.wait_on + 40 + (blockIdx.x & 31));
            const unsigned long long t0 = now();
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.wait_target) {
                if ((long long)(now() - t0) > 20000000) { *a.err = 1u + a.slot; break; }       // 200 ms: never a hang
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    if (a.body_ticks) spin_ticks(a.body_ticks);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (a.signal) {
            unsigned* base = a.counters + 64 * 80 * (size_t)a.slot;
            const unsigned g = blockIdx.x & 31, gsize = (a.nwg - g + 31) / 32;
            if (__hip_atomic_fetch_add(base + 64 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1)
                if (__hip_atomic_fetch_add(base + 64 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 31)
                    for (int i = 0; i < 32; ++i) __hip_atomic_store(base + 64 * (40 + i), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t_out = now();
        unsigned long long* s = a.stamps + 2 * ((size_t)a.slot * a.nwg + blockIdx.x);
        const unsigned long long xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;      // HW_REG_XCC_ID[3:0]
        s[0] = t_in | (xcc << 60); s[1] = t_out;
    }
}

struct OrderArgs {
    unsigned* started;     // A: bumped at entry; started[64 + 16 * xcc]: the same per XCD
    unsigned* finished;    // A: bumped at exit
    unsigned* seen_min;    // B: min over workgroups of `started` at entry; seen_min[64 + 16 * xcc]: of the XCD's own counter
    unsigned* err;
    unsigned a_grid;
    int role;              // 0 = A, 1 = B
    int body_ticks;
};

extern "C" __global__ __launch_bounds__(256) void order_kernel(OrderArgs a) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;      // HW_REG_XCC_ID[3:0]
    if (a.role == 0) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(a.started + 64 + 16 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        spin_ticks(a.body_ticks);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (threadIdx.x == 0) {
            const unsigned sx = __hip_atomic_load(a.started + 64 + 16 * xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned s = __hip_atomic_load(a.started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMin(a.seen_min, s);
            atomicMin(a.seen_min + 64 + 16 * xcc, sx);
            const unsigned long long t0 = now();
            while (__hip_atomic_load(a.finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.a_grid) {
                if ((long long)(now() - t0) > 20000000) { *a.err = 1; break; }                  // 200 ms
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ------------------------------------------------------------------------------------------------------------- HSA plumbing
static hsa_agent_t g_gpu; static bool g_have_gpu = false;
static hsa_status_t agent_cb(hsa_agent_t ag, void*) {
    hsa_device_type_t t; hsa_agent_get_info(ag, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = ag; g_have_gpu = true; }
    return HSA_STATUS_SUCCESS;
}

struct Kernel { uint64_t object; uint32_t kernarg, group, priv; };

static Kernel get_kernel(hsa_executable_t exe, const char* name) {
    hsa_executable_symbol_t sym;
    HCHECK(hsa_executable_get_symbol_by_name(exe, name, &g_gpu, &sym));
    Kernel k;
    HCHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HCHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    HCHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    HCHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    printf("  %s: object 0x%llx kernarg %u B group %u B private %u B\n", name, (unsigned long long)k.object, k.kernarg, k.group, k.priv);
    return k;
}

static char* g_khost = nullptr; static char* g_kdev = nullptr;      // kernarg staging (host-coherent) and its device copy (argv[2] = "dev")
struct Queue {
    hsa_queue_t* q = nullptr;
    hsa_signal_t done{};
    void create() {
        HCHECK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
        HCHECK(hsa_signal_create(1, 0, nullptr, &done));
    }
    // Writes the packets behind the current write index, one doorbell, waits for the LAST packet's completion signal.
    // barrier / fence apply to every packet but the first (barrier 0) -- the queue is idle when a batch starts -- and the
    // tail: a BARRIER_AND packet with the barrier bit, so `done` means "everything before it has completed".
    double submit_and_wait(const std::vector<hsa_kernel_dispatch_packet_t>& pk, bool barrier, int fence_scope) {
        const uint64_t base = hsa_queue_load_write_index_relaxed(q);
        const uint32_t mask = q->size - 1;
        auto* ring = (hsa_kernel_dispatch_packet_t*)q->base_address;
        hsa_signal_store_relaxed(done, 1);
        const size_t n = pk.size();
        std::vector<uint32_t> first(n + 1);
        if (g_kdev) { CHECK(hipMemcpy(g_kdev, g_khost, 256 * 1024, hipMemcpyHostToDevice)); CHECK(hipDeviceSynchronize()); }
        for (size_t i = 0; i < n; ++i) {
            hsa_kernel_dispatch_packet_t p = pk[i];
            if (g_kdev) p.kernarg_address = g_kdev + ((char*)p.kernarg_address - g_khost);
            uint16_t hdr = HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE;
            hdr |= (uint16_t)((barrier && i > 0) ? 1 : 0) << HSA_PACKET_HEADER_BARRIER;
            hdr |= (uint16_t)fence_scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE;
            hdr |= (uint16_t)fence_scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE;
            first[i] = hdr | ((uint32_t)p.setup << 16);
            auto* slot = ring + ((base + i) & mask);
            p.header = HSA_PACKET_TYPE_INVALID << HSA_PACKET_HEADER_TYPE;
            *slot = p;
        }
        {   // tail barrier packet
            hsa_barrier_and_packet_t b; memset(&b, 0, sizeof b);
            b.completion_signal = done;
            uint16_t hdr = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                           (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            first[n] = hdr;
            b.header = HSA_PACKET_TYPE_INVALID << HSA_PACKET_HEADER_TYPE;
            memcpy(ring + ((base + n) & mask), &b, sizeof b);
        }
        timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
        for (size_t i = n + 1; i-- > 0;)                                       // headers last-to-first: the CP stops at an INVALID header
            __atomic_store_n((uint32_t*)(ring + ((base + i) & mask)), first[i], __ATOMIC_RELEASE);
        hsa_queue_store_write_index_screlease(q, base + n + 1);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(base + n));
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) { printf("  (still waiting)\n"); }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        return (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
    }
};

static unsigned g_dyn_lds = 0;
// Packet i goes to queue i % nq: different HSA queues sit on different hardware queues, which do run side by side.
static double submit_multi(std::vector<Queue*> qs, const std::vector<hsa_kernel_dispatch_packet_t>& pk, int fence_scope) {
    const size_t nq = qs.size();
    if (g_kdev) { CHECK(hipMemcpy(g_kdev, g_khost, 256 * 1024, hipMemcpyHostToDevice)); CHECK(hipDeviceSynchronize()); }
    std::vector<uint64_t> base(nq), cnt(nq, 0);
    for (size_t j = 0; j < nq; ++j) { base[j] = hsa_queue_load_write_index_relaxed(qs[j]->q); hsa_signal_store_relaxed(qs[j]->done, 1); }
    std::vector<std::pair<uint32_t*, uint32_t>> hdrs;
    for (size_t i = 0; i < pk.size(); ++i) {
        const size_t j = i % nq; Queue* Q = qs[j];
        hsa_kernel_dispatch_packet_t p = pk[i];
        if (g_kdev) p.kernarg_address = g_kdev + ((char*)p.kernarg_address - g_khost);
        uint16_t hdr = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (uint16_t)(fence_scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (uint16_t)(fence_scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        const uint32_t first = hdr | ((uint32_t)p.setup << 16);
        auto* slot = (hsa_kernel_dispatch_packet_t*)Q->q->base_address + ((base[j] + cnt[j]++) & (Q->q->size - 1));
        p.header = HSA_PACKET_TYPE_INVALID << HSA_PACKET_HEADER_TYPE; *slot = p;
        hdrs.push_back({(uint32_t*)slot, first});
    }
    for (size_t j = 0; j < nq; ++j) {
        hsa_barrier_and_packet_t bp; memset(&bp, 0, sizeof bp); bp.completion_signal = qs[j]->done;
        const uint16_t hdr = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                             (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        auto* slot = (hsa_kernel_dispatch_packet_t*)qs[j]->q->base_address + ((base[j] + cnt[j]++) & (qs[j]->q->size - 1));
        bp.header = HSA_PACKET_TYPE_INVALID << HSA_PACKET_HEADER_TYPE; memcpy(slot, &bp, sizeof bp);
        hdrs.push_back({(uint32_t*)slot, (uint32_t)hdr});
    }
    timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t i = hdrs.size(); i-- > 0;) __atomic_store_n(hdrs[i].first, hdrs[i].second, __ATOMIC_RELEASE);
    for (size_t j = 0; j < nq; ++j) {
        hsa_queue_store_write_index_screlease(qs[j]->q, base[j] + cnt[j]);
        hsa_signal_store_screlease(qs[j]->q->doorbell_signal, (hsa_signal_value_t)(base[j] + cnt[j] - 1));
    }
    for (size_t j = 0; j < nq; ++j)
        while (hsa_signal_wait_scacquire(qs[j]->done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) { printf("  (still waiting)\n"); }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
}

static hsa_kernel_dispatch_packet_t make_packet(const Kernel& k, unsigned grid_wg, unsigned wg, void* kernarg) {
    hsa_kernel_dispatch_packet_t p; memset(&p, 0, sizeof p);
    p.setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    p.workgroup_size_x = (uint16_t)wg; p.workgroup_size_y = 1; p.workgroup_size_z = 1;
    p.grid_size_x = grid_wg * wg; p.grid_size_y = 1; p.grid_size_z = 1;
    p.private_segment_size = k.priv; p.group_segment_size = k.group + g_dyn_lds;
    p.kernel_object = k.object; p.kernarg_address = kernarg;
    return p;
}

// ------------------------------------------------------------------------------------------------------------------ main
struct Buffers {
    unsigned long long* stamps; unsigned* counters; unsigned* err; unsigned* misc;
    char* kernarg;    // host-coherent, 512 B per packet
};
static constexpr int MAXST = 256;
static constexpr int MAXWG = 768;
static int g_wg = 768;      // workgroups per stage of the current configuration (the kernels index stamps with it)

static void reset(Buffers& b) {
    CHECK(hipMemset(b.stamps, 0, (size_t)8 * 2 * MAXWG * MAXST));
    CHECK(hipMemset(b.counters, 0, (size_t)64 * 4 * 80 * MAXST));
    CHECK(hipMemset(b.err, 0, 256));
    std::vector<unsigned> m(1024, 0);                                   // [0..255] started (+ per XCD), [256..511] finished, [512..767] seen_min (+ per XCD)
    for (int i = 512; i < 768; ++i) m[i] = ~0u;
    CHECK(hipMemcpy(b.misc, m.data(), 4096, hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
}

struct StageTimes { double start_to_start, end_to_start, skew_in, span; };
static std::vector<unsigned long long> g_xs;    // per (stage, XCD): first start, last end
static std::vector<unsigned long long> read_stamps(Buffers& b, int n) {
    g_xs.assign((size_t)n * 16, 0); for (int i = 0; i < n * 8; ++i) g_xs[2 * i] = ~0ull;
    std::vector<unsigned long long> raw((size_t)2 * MAXWG * n), s(4 * n);
    CHECK(hipMemcpy(raw.data(), b.stamps, raw.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {                                      // min start, max start, min end, max end over the stage's workgroups
        unsigned long long a0 = ~0ull, a1 = 0, e0 = ~0ull, e1 = 0;
        for (int w = 0; w < g_wg; ++w) {
            const unsigned long long t_raw = raw[2 * ((size_t)i * g_wg + w)], t_out = raw[2 * ((size_t)i * g_wg + w) + 1];
            const unsigned long long t_in = t_raw & ((1ull << 60) - 1); const int x = (int)(t_raw >> 60) & 7;
            g_xs[2 * (i * 8 + x)] = std::min(g_xs[2 * (i * 8 + x)], t_in); g_xs[2 * (i * 8 + x) + 1] = std::max(g_xs[2 * (i * 8 + x) + 1], t_out);
            a0 = std::min(a0, t_in); a1 = std::max(a1, t_in); e0 = std::min(e0, t_out); e1 = std::max(e1, t_out);
        }
        s[4 * i] = a0; s[4 * i + 1] = a1; s[4 * i + 2] = e0; s[4 * i + 3] = e1;
    }
    return s;
}
static unsigned read_err(Buffers& b) { unsigned e; CHECK(hipMemcpy(&e, b.err, 4, hipMemcpyDeviceToHost)); return e; }

static void report_pair(const char* name, const std::vector<unsigned long long>& s) {
    // ticks of 10 ns
    printf("  %-46s  A: start skew %5.2f us, first start -> last end %6.2f us | B first start - A first start %6.2f us | B first start - A last end %+6.2f us\n",
           name, (s[1] - s[0]) * 0.01, (s[3] - s[0]) * 0.01, ((long long)(s[4] - s[0])) * 0.01, ((long long)(s[4] - s[3])) * 0.01);
}

static void report_chain(const char* name, const std::vector<unsigned long long>& s, int n, double host_us, unsigned err) {
    std::vector<double> s2s, gap;
    for (int i = 1; i < n; ++i) {
        s2s.push_back(((long long)(s[4 * i] - s[4 * (i - 1)])) * 0.01);
        gap.push_back(((long long)(s[4 * i] - s[4 * (i - 1) + 3])) * 0.01);
    }
    std::vector<double> sk, lastst, esk;
    for (int i = 1; i < n; ++i) {
        sk.push_back((s[4 * i + 1] - s[4 * i]) * 0.01);                                  // start skew inside the stage
        lastst.push_back(((long long)(s[4 * i + 1] - s[4 * (i - 1) + 3])) * 0.01);       // LAST start - previous last end
        esk.push_back((s[4 * i + 3] - s[4 * i + 2]) * 0.01);                              // end skew
    }
    std::sort(sk.begin(), sk.end()); std::sort(lastst.begin(), lastst.end()); std::sort(esk.begin(), esk.end());
    std::sort(s2s.begin(), s2s.end()); std::sort(gap.begin(), gap.end());
    std::vector<double> xg;                                                               // SAME-XCD first start of stage k+1 - last end of stage k
    for (int i = 1; i < n; ++i) for (int x = 0; x < 8; ++x)
        if (g_xs[2 * (i * 8 + x)] != ~0ull && g_xs[2 * ((i - 1) * 8 + x) + 1]) xg.push_back(((long long)(g_xs[2 * (i * 8 + x)] - g_xs[2 * ((i - 1) * 8 + x) + 1])) * 0.01);
    std::sort(xg.begin(), xg.end());
    const double total = (s[4 * (n - 1) + 3] - s[0]) * 0.01;
    printf("  %-46s  %3d stages: device span %8.2f us = %6.3f us/stage | start-to-start p50 %6.2f | first-start - prev-last-end p50 %+6.2f (min %+6.2f max %+6.2f) | host wall %8.1f us | err %u\n"
           "  %-46s      p50: start skew %5.2f | LAST start - prev last end %+5.2f | end skew %5.2f || on the SAME XCD, first start - prev last end: min %+5.2f p50 %+5.2f max %+5.2f\n",
           name, n, total, total / n, s2s[s2s.size() / 2], gap[gap.size() / 2], gap.front(), gap.back(), host_us, err,
           "", sk[sk.size() / 2], lastst[lastst.size() / 2], esk[esk.size() / 2], xg.front(), xg[xg.size() / 2], xg.back());
}

int main(int argc, char** argv) {
    std::string hsaco = argc > 1 ? argv[1] : "aql_chain.hsaco";
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    const int ncu = prop.multiProcessorCount;

    Buffers b;
    CHECK(hipMalloc(&b.stamps, (size_t)8 * 2 * MAXWG * MAXST)); CHECK(hipMalloc(&b.counters, (size_t)64 * 4 * 80 * MAXST)); CHECK(hipMalloc(&b.err, 256)); CHECK(hipMalloc(&b.misc, 4096));
    CHECK(hipHostMalloc(&b.kernarg, 256 * 1024, hipHostMallocCoherent));
    g_khost = b.kernarg;
    if (argc > 2 && !strcmp(argv[2], "dev")) CHECK(hipMalloc(&g_kdev, 256 * 1024));
    printf("own-queue kernargs in %s memory\n", g_kdev ? "DEVICE" : "host-coherent");
    hipStream_t st; CHECK(hipStreamCreate(&st));

    // ---- HSA: own queue + the code object through the HSA loader
    HCHECK(hsa_init());
    HCHECK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have_gpu) { printf("no GPU agent\n"); return 1; }
    FILE* f = fopen(hsaco.c_str(), "rb"); if (!f) { printf("cannot open %s\n", hsaco.c_str()); return 1; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> image(sz); if (fread(image.data(), 1, sz, f) != (size_t)sz) return 1; fclose(f);
    hsa_code_object_reader_t reader; HCHECK(hsa_code_object_reader_create_from_memory(image.data(), image.size(), &reader));
    hsa_executable_t exe; HCHECK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HCHECK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HCHECK(hsa_executable_freeze(exe, nullptr));
    printf("code object %s loaded through the HSA loader:\n", hsaco.c_str());
    Kernel k_stage = get_kernel(exe, "stage_kernel.kd");
    Kernel k_order = get_kernel(exe, "order_kernel.kd");
    Queue Q; Q.create();
    Queue Q2; Q2.create(); Queue Q3; Q3.create(); Queue Q4; Q4.create();

    unsigned WG = 768;                  // the decode GEMV's grid: 3 x 256
    // ================================================================================================================ (a)
    printf("\n(a) two trivial %u-workgroup kernels back to back (512 threads; body 0 / 3 us); best of 20 trials by B-start - A-end\n", WG);
    for (int body : {0, 300}) {
        auto args_of = [&](int slot) { StageArgs a{b.stamps, b.counters, b.err, slot, -1, 0u, body, 0, 0, (int)WG}; return a; };
        struct Mode { const char* name; int kind; bool barrier; int fence; };
        const Mode modes[] = {
            {"HIP stream (barrier bit, agent fences)", 0, true, 0},
            {"hipExtLaunchKernelGGL + hipExtAnyOrderLaunch", 1, false, 0},
            {"own AQL queue, barrier 1, fences AGENT", 2, true, HSA_FENCE_SCOPE_AGENT},
            {"own AQL queue, barrier 1, fences NONE", 2, true, HSA_FENCE_SCOPE_NONE},
            {"own AQL queue, barrier 0, fences AGENT", 2, false, HSA_FENCE_SCOPE_AGENT},
            {"own AQL queue, barrier 0, fences NONE", 2, false, HSA_FENCE_SCOPE_NONE},
        };
        for (const Mode& m : modes) {
            std::vector<unsigned long long> best; long long best_gap = 1ll << 60;
            for (int trial = 0; trial < 20; ++trial) {
                reset(b);
                if (m.kind == 0) {
                    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stage_kernel, dim3(WG), dim3(512), g_dyn_lds, st, args_of(i));
                    CHECK(hipStreamSynchronize(st));
                } else if (m.kind == 1) {
                    for (int i = 0; i < 2; ++i) hipExtLaunchKernelGGL(stage_kernel, dim3(WG), dim3(512), g_dyn_lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, args_of(i));
                    CHECK(hipStreamSynchronize(st));
                } else {
                    std::vector<hsa_kernel_dispatch_packet_t> pk;
                    for (int i = 0; i < 2; ++i) { StageArgs a = args_of(i); memcpy(b.kernarg + 512 * i, &a, sizeof a); pk.push_back(make_packet(k_stage, WG, 512, b.kernarg + 512 * i)); }
                    Q.submit_and_wait(pk, m.barrier, m.fence);
                }
                auto s = read_stamps(b, 2);
                const long long gap = (long long)(s[4] - s[3]);
                if (gap < best_gap || best.empty()) { best_gap = gap; best = s; }
            }
            char nm[96]; snprintf(nm, sizeof nm, "[body %d us] %s", body / 100, m.name);
            report_pair(nm, best);
        }
    }

    // ================================================================================================================ (b)
    printf("\n(b) in-order launch: A = 4 x resident capacity (%d workgroups of 256 threads, body 5 us), B = chip-filling (%d), barrier 0, fences NONE\n", ncu * 8 * 4, ncu * 8);
    {
        const unsigned a_grid = (unsigned)ncu * 8 * 4, b_grid = (unsigned)ncu * 8;
        int ok = 0, ok_xcd = 0, trials = 10; unsigned worst = ~0u; unsigned errs = 0;
        for (int trial = 0; trial < trials; ++trial) {
            reset(b);
            OrderArgs aa{b.misc + 0, b.misc + 256, b.misc + 512, b.err, a_grid, 0, 500};
            OrderArgs bb = aa; bb.role = 1;
            memcpy(b.kernarg, &aa, sizeof aa); memcpy(b.kernarg + 512, &bb, sizeof bb);
            std::vector<hsa_kernel_dispatch_packet_t> pk{make_packet(k_order, a_grid, 256, b.kernarg), make_packet(k_order, b_grid, 256, b.kernarg + 512)};
            Q.submit_and_wait(pk, false, HSA_FENCE_SCOPE_NONE);
            unsigned m[1024]; CHECK(hipMemcpy(m, b.misc, 4096, hipMemcpyDeviceToHost));
            const unsigned e = read_err(b);
            bool per_xcd_ok = true;
            for (int x = 0; x < 8; ++x) per_xcd_ok &= (m[512 + 64 + 16 * x] == m[64 + 16 * x]);
            if (per_xcd_ok && !e) ++ok_xcd;
            if (m[512] == a_grid && !e) ++ok;
            worst = std::min(worst, m[512]); errs += e;
            if (trial < 3) {
                printf("  trial %d: A started %u finished %u | min over B workgroups of A-started-at-B-entry %u | timeout %u\n    per XCD (A workgroups on it : min over ITS B workgroups of ITS A-started):", trial, m[0], m[256], m[512], e);
                for (int x = 0; x < 8; ++x) printf(" %u:%u", m[64 + 16 * x], m[512 + 64 + 16 * x]);
                printf("\n");
            }
        }
        printf("  => %d / %d trials: every B workgroup entered after ALL %u A workgroups had entered (worst min %u), timeouts %u\n", ok, trials, a_grid, worst, errs);
        printf("  => %d / %d trials: on EVERY XCD, every B workgroup entered after all of that XCD's A workgroups had entered\n", ok_xcd, trials);
        // three packets: A, B (waits for A, chip-filling), A' -- B must not strand A' either (it does not wait for it), sanity of a longer queue
    }

    // ================================================================================================================ (c)
    const int N = 160;
    for (unsigned cfg = 0; cfg < 4; ++cfg) {
    const unsigned dyn = cfg == 1 ? 49152u : 0u;
    WG = cfg == 2 ? 256 : cfg == 3 ? 384 : 768; g_wg = (int)WG;
    g_dyn_lds = dyn;
    printf("\n=== %u workgroups per stage", WG);
    printf("\n=== dynamic LDS %u B per workgroup: %s\n", dyn, dyn ? "3 workgroups per CU, the decode GEMV's residency -- a successor's workgroup enters only when a predecessor's retires" : "4 workgroups per CU fit");
    printf("\n(c) a chain of %d dependent %u-workgroup stages (prologue P, body T; 10 ns ticks), one submission\n", N, WG);
    struct Shape { int pro, body; };
    for (Shape sh : {Shape{0, 0}, Shape{100, 300}, Shape{150, 500}}) {
        printf(" prologue %.1f us + body %.1f us:\n", sh.pro * 0.01, sh.body * 0.01);
        // (i) HIP graph of N launches (barrier bit): what the decode step does today
        {
            reset(b);
            hipGraph_t g; hipGraphExec_t ge;
            CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) { StageArgs a{b.stamps, b.counters, b.err, i, -1, 0u, sh.body + sh.pro, 0, 0, (int)WG}; hipLaunchKernelGGL(stage_kernel, dim3(WG), dim3(512), g_dyn_lds, st, a); }
            CHECK(hipStreamEndCapture(st, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CHECK(hipGraphLaunch(ge, st)); CHECK(hipStreamSynchronize(st));
            reset(b);
            timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
            CHECK(hipGraphLaunch(ge, st)); CHECK(hipStreamSynchronize(st));
            clock_gettime(CLOCK_MONOTONIC, &t1);
            report_chain("hipGraph, queue barrier between stages", read_stamps(b, N), N, (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3, read_err(b));
            CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
        }
        // (ii) own queue, barrier bit 1 (control: same ordering, our submission)
        for (int fence : {(int)HSA_FENCE_SCOPE_AGENT, (int)HSA_FENCE_SCOPE_NONE}) {
            double host = 0;
            for (int rep = 0; rep < 2; ++rep) {
                reset(b);
                std::vector<hsa_kernel_dispatch_packet_t> pk;
                for (int i = 0; i < N; ++i) { StageArgs a{b.stamps, b.counters, b.err, i, -1, 0u, sh.body + sh.pro, 0, 0, (int)WG}; memcpy(b.kernarg + 512 * i, &a, sizeof a); pk.push_back(make_packet(k_stage, WG, 512, b.kernarg + 512 * i)); }
                host = Q.submit_and_wait(pk, true, fence);
            }
            report_chain(fence ? "own queue, barrier 1, fences AGENT" : "own queue, barrier 1, fences NONE", read_stamps(b, N), N, host, read_err(b));
        }
        // (iii) own queue, barrier bit 0, ordered by epoch counters: stage i waits (after its prologue) for counters[i-1] == WG
        for (int fence : {(int)HSA_FENCE_SCOPE_AGENT, (int)HSA_FENCE_SCOPE_NONE}) {
            double host = 0;
            for (int rep = 0; rep < 2; ++rep) {
                reset(b);
                std::vector<hsa_kernel_dispatch_packet_t> pk;
                for (int i = 0; i < N; ++i) { StageArgs a{b.stamps, b.counters, b.err, i, i - 1, 1u, sh.body, sh.pro, 1, (int)WG}; memcpy(b.kernarg + 512 * i, &a, sizeof a); pk.push_back(make_packet(k_stage, WG, 512, b.kernarg + 512 * i)); }
                host = Q.submit_and_wait(pk, false, fence);
            }
            report_chain(fence ? "own queue, barrier 0 + flags, fences AGENT" : "own queue, barrier 0 + flags, fences NONE", read_stamps(b, N), N, host, read_err(b));
        }
        // (iv) barrier 0, NO ordering at all (independent kernels): the packet processor's own rate
        {
            double host = 0;
            for (int rep = 0; rep < 2; ++rep) {
                reset(b);
                std::vector<hsa_kernel_dispatch_packet_t> pk;
                for (int i = 0; i < N; ++i) { StageArgs a{b.stamps, b.counters, b.err, i, -1, 0u, sh.body + sh.pro, 0, 0, (int)WG}; memcpy(b.kernarg + 512 * i, &a, sizeof a); pk.push_back(make_packet(k_stage, WG, 512, b.kernarg + 512 * i)); }
                host = Q.submit_and_wait(pk, false, HSA_FENCE_SCOPE_NONE);
            }
            report_chain("own queue, barrier 0, independent", read_stamps(b, N), N, host, read_err(b));
        }
        // (v) stage i on queue i % nq, no ordering at all: do DIFFERENT queues overlap?
        for (int nq : {2, 4}) {
            double host = 0;
            std::vector<Queue*> qs{&Q, &Q2}; if (nq == 4) { qs.push_back(&Q3); qs.push_back(&Q4); }
            for (int rep = 0; rep < 2; ++rep) {
                reset(b);
                std::vector<hsa_kernel_dispatch_packet_t> pk;
                for (int i = 0; i < N; ++i) { StageArgs a{b.stamps, b.counters, b.err, i, -1, 0u, sh.body + sh.pro, 0, 0, (int)WG}; memcpy(b.kernarg + 512 * i, &a, sizeof a); pk.push_back(make_packet(k_stage, WG, 512, b.kernarg + 512 * i)); }
                host = submit_multi(qs, pk, HSA_FENCE_SCOPE_NONE);
            }
            report_chain(nq == 2 ? "TWO queues round-robin, independent" : "FOUR queues round-robin, independent", read_stamps(b, N), N, host, read_err(b));
        }
        // (vi) two queues, flag-ordered -- only where TWO stages fit on the chip together (else a waiting successor could strand
        //      its predecessor's unlaunched workgroups: different queues give no launch order)
        if (WG <= 384 && dyn == 0) {
            double host = 0;
            for (int rep = 0; rep < 2; ++rep) {
                reset(b);
                std::vector<hsa_kernel_dispatch_packet_t> pk;
                for (int i = 0; i < N; ++i) { StageArgs a{b.stamps, b.counters, b.err, i, i - 1, 1u, sh.body, sh.pro, 1, (int)WG}; memcpy(b.kernarg + 512 * i, &a, sizeof a); pk.push_back(make_packet(k_stage, WG, 512, b.kernarg + 512 * i)); }
                host = submit_multi({&Q, &Q2}, pk, HSA_FENCE_SCOPE_NONE);
            }
            report_chain("TWO queues round-robin + flags", read_stamps(b, N), N, host, read_err(b));
        }
    }
    }
    g_dyn_lds = 0;
    HCHECK(hsa_queue_destroy(Q.q)); HCHECK(hsa_queue_destroy(Q2.q)); HCHECK(hsa_queue_destroy(Q3.q)); HCHECK(hsa_queue_destroy(Q4.q));
    printf("done\n");
    return 0;
}
#endif
