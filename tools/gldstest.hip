// Semantics check of LDS-DMA (global_load_lds_dwordx4) with 8-byte-aligned (not 16-byte-aligned) per-lane
// global addresses, wave-uniform LDS base, explicit vmcnt wait + raw barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k(const double *__restrict__ src, double *__restrict__ dst, int shift_doubles) {
    __shared__ __attribute__((aligned(16))) double lds[4 * 128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // wave w copies 64 pieces of 16 B starting at src + shift + w*128 doubles
    const double *g = src + shift_doubles + w * 128 + 2 * lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                     (__attribute__((address_space(3))) void *) (lds + w * 128), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // read back another wave's data to check cross-wave visibility after the barrier
    const int ww = (w + 1) & 3;
    dst[blockIdx.x * 512 + threadIdx.x * 2] = lds[ww * 128 + 2 * lane];
    dst[blockIdx.x * 512 + threadIdx.x * 2 + 1] = lds[ww * 128 + 2 * lane + 1];
}

int main() {
    const int n = 4096;
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = i + 0.25;
    double *s, *d; CK(hipMalloc(&s, n * 8)); CK(hipMalloc(&d, 512 * 8));
    CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice));
    for (int shift : {0, 1, 3}) {
        CK(hipMemset(d, 0, 512 * 8));
        k<<<1, 256>>>(s, d, shift);
        CK(hipDeviceSynchronize());
        std::vector<double> o(512);
        CK(hipMemcpy(o.data(), d, 512 * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int t = 0; t < 256; ++t) {
            const int lane = t & 63, w = t >> 6, ww = (w + 1) & 3;
            for (int c = 0; c < 2; ++c) {
                const double want = h[shift + ww * 128 + 2 * lane + c];
                if (o[2 * t + c] != want) { if (bad < 4) printf("shift %d t %d c %d got %g want %g\n", shift, t, c, o[2 * t + c], want); ++bad; }
            }
        }
        printf("shift %d doubles: %s (%d mismatches)\n", shift, bad ? "FAIL" : "ok", bad);
    }
    return 0;
}
