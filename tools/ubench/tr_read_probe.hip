// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does lane L receive?
// LDS holds u16 values = element index of a [16 rows][STRIDE halves] matrix; lane L (group g = L>>4, i = L&15 = 4q+p) supplies the
// address of row (4g + q), columns 4p..4p+3.  Run 1: 32-byte rows (the 16 addresses of a group are contiguous); run 2: 256-byte
// rows (each row's four lanes contiguous, rows far apart) -- are per-lane addresses honoured?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
template <int STRIDE>
__global__ void probe(short* out) {
    __shared__ short sm[16 * STRIDE];
    for (int i = threadIdx.x; i < 16 * STRIDE; i += 64) sm[i] = (short)((i / STRIDE) * 16 + (i % STRIDE) % 16 + ((i % STRIDE) >= 16 ? 1000 : 0));
    __syncthreads();
    const int L = threadIdx.x, g = L >> 4, q = (L & 15) >> 2, p = L & 3;
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + (4 * g + q) * STRIDE + 4 * p));
    for (int e = 0; e < 4; ++e) out[L * 4 + e] = v[e];
}
template <int STRIDE> void run() {
    short* d; (void)hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(probe<STRIDE>, dim3(1), dim3(64), 0, 0, d);
    short h[256]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("row stride %d halves\n", STRIDE);
    for (int L = 0; L < 64; L += 5) {
        printf("  lane %2d:", L);
        for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[L * 4 + e] / 16, h[L * 4 + e] % 16);
        printf("\n");
    }
}
int main() { run<16>(); run<128>(); return 0; }
