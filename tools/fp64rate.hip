// fp64 FMA issue rate on gfx950: cycles per wave64 v_fma_f64 per SIMD for 1 / 2 / 4 waves per SIMD, coefficient operand in a VGPR or an SGPR.
//   hipcc -O3 --offload-arch=gfx950 -o tools/fp64rate tools/fp64rate.hip && tools/fp64rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: coefficient in a VGPR, 1: coefficient in an SGPR pair, 2: SGPR with negated use mixed (v_fma_f64 VOP3)
__global__ void __launch_bounds__(1024) k_rate(double *out, const double *coef, int iters, long long *cycles) {
    double acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double c[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) c[i] = coef[i];
    double x = out[threadIdx.x & 63];
    if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) (__double_as_longlong(c[i]) & 0xffffffffll));
            unsigned hi = __builtin_amdgcn_readfirstlane((unsigned) (__double_as_longlong(c[i]) >> 32));
            c[i] = __longlong_as_double(((long long) hi << 32) | lo);
        }
    }
    long long t0 = 0, t1 = 0;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                if (MODE == 2 && (i & 1)) acc[i] = fma(-c[(i + r) % 6], x, acc[i]);
                else acc[i] = fma(c[(i + r) % 6], x, acc[i]);
            }
        asm volatile("" : "+v"(x));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    double *out, *coef; long long *cyc;
    const int nblk = 256;
    hipMalloc(&out, sizeof(double) * nblk * 1024); hipMalloc(&coef, 64); hipMalloc(&cyc, sizeof(long long) * nblk);
    hipMemset(out, 0, sizeof(double) * nblk * 1024);
    double hc[6] = {1e-9, -2e-9, 3e-9, 1.5e-9, -2.5e-9, 0.5e-9};
    hipMemcpy(coef, hc, sizeof(hc), hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int waves : {4, 8, 16}) {          // per block = per CU: 1, 2, 4 waves per SIMD
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k_rate<0><<<nblk, waves * 64>>>(out, coef, iters, cyc);
                else if (mode == 1) k_rate<1><<<nblk, waves * 64>>>(out, coef, iters, cyc);
                else k_rate<2><<<nblk, waves * 64>>>(out, coef, iters, cyc);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            std::vector<long long> h(nblk);
            hipMemcpy(h.data(), cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : h) mean += v; mean /= nblk;
            const double fmas_per_simd = (double) iters * 96 * (waves / 4);
            const double tf = 2.0 * nblk * waves * 64.0 * iters * 96 / (ms * 1e-3) / 1e12;
            printf("mode %d (%s)  %d wave(s) per SIMD: %.0f s_memtime ticks, %.3f ms by events -> %.2f ticks per wave64 FMA per SIMD, %.1f TFLOP/s on %d CUs, %.2f ns per tick\n", mode,
                   mode == 0 ? "VGPR coef" : (mode == 1 ? "SGPR coef" : "SGPR coef, half negated (VOP3)"), waves / 4, mean, ms, mean / fmas_per_simd, tf, nblk, ms * 1e6 / mean);
        }
    return 0;
}
