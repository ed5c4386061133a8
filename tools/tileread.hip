// Read-pattern microbenchmark for the x-marching apply kernel: how fast can 512-thread blocks pull
// (TY+1) rows x 195 doubles per plane while marching through planes?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VARIANT>   // 0: 8-byte loads, tile rows (as the kernel); 1: 16-byte loads; 2: march along y instead of x
__global__ void __launch_bounds__(512) tile_read(const double *__restrict__ u, double *__restrict__ sink, int NX, int NY, int NZ, int ppc) {
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int k0 = blockIdx.y * 63, j0 = blockIdx.z * 7;
    const int p0 = blockIdx.x * ppc;
    int p1 = p0 + ppc; if (p1 > NX) p1 = NX;
    const long long plane = (long long) NY * NZ;
    double acc = 0.0;
    if (VARIANT == 0 || VARIANT == 2) {
        int goff[4];
        for (int s = 0; s < 4; ++s) {
            const int L = s * 512 + tid, r = L / 195, q = L - r * 195;
            int jj = j0 + r; if (jj > NY - 1) jj = NY - 1;
            long long g = 3LL * ((long long) jj * NZ + k0) + q;
            if (g > 3 * plane - 1) g = 3 * plane - 1;
            goff[s] = (int) g;
        }
        for (int i = p0; i < p1; ++i) {
            const double *up = (VARIANT == 0) ? u + 3 * (long long) i * plane : u + 3 * (long long) i * plane;
            for (int s = 0; s < 4; ++s) acc += up[goff[s]];
        }
    } else {
        // 16-byte loads: rows of 98 double2 (1568 B), 9 rows = 882 double2 <= 2 * 512
        int goff[2];
        for (int s = 0; s < 2; ++s) {
            const int L = s * 512 + tid, r = L / 98, q = L - r * 98;
            int jj = j0 + r; if (jj > NY - 1) jj = NY - 1;
            long long g = (3LL * ((long long) jj * NZ + k0)) / 2 + q;
            if (g > 3 * plane / 2 - 1) g = 3 * plane / 2 - 1;
            goff[s] = (int) g;
        }
        for (int i = p0; i < p1; ++i) {
            const double2 *up = (const double2 *) (u + 3 * (long long) i * plane);
            for (int s = 0; s < 2; ++s) { double2 v = up[goff[s]]; acc += v.x + v.y; }
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

int main() {
    const int NX = 513, NY = 513, NZ = 513;
    const long long n = 3LL * NX * NY * NZ;
    double *u, *sink;
    CK(hipMalloc(&u, n * 8 + 64)); CK(hipMalloc(&sink, 64)); CK(hipMemset(u, 0, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int chunks : {8, 16, 64}) {
        const int ppc = (NX + chunks - 1) / chunks;
        dim3 blk(64, 8), grd(chunks, (NZ + 62) / 63, (NY + 6) / 7);
        for (int v = 0; v < 2; ++v) {
            float ms = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (v == 0) tile_read<0><<<grd, blk>>>(u, sink, NX, NY, NZ, ppc); else tile_read<1><<<grd, blk>>>(u, sink, NX, NY, NZ, ppc);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms += t / 3;
            }
            printf("chunks %3d variant %d: %.3f ms  %.2f TB/s (unique bytes)\n", chunks, v, ms, n * 8.0 / ms / 1e9);
        }
    }
    return 0;
}
