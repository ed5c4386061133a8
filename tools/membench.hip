// HBM microbenchmarks to calibrate what the apply kernel's access shapes can reach on this box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void copy8(const double *a, double *b, long long n) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy16(const double2 *a, double2 *b, long long n) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read8(const double *a, double *b, long long n) {
    double s = 0;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) s += a[i];
    if (s == 1.2345) b[0] = s;
}
__global__ void read16(const double2 *a, double *b, long long n) {
    double s = 0;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345) b[0] = s;
}
// AoS xyz per lane: 3 loads / 3 stores of 8 B at 24-B stride (what a node-per-lane kernel does)
__global__ void copy_aos(const double *a, double *b, long long nn) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += (long long) gridDim.x * blockDim.x) {
        double x = a[3 * i], y = a[3 * i + 1], z = a[3 * i + 2];
        b[3 * i] = x; b[3 * i + 1] = y; b[3 * i + 2] = z;
    }
}
__global__ void write8(double *b, long long n) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) b[i] = 1.0;
}
int main() {
    const long long n = 1LL << 29;   // 4 GiB per array
    double *a, *b;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, double bytes, auto fn) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-28s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    for (int grid : {2048, 8192, 32768}) {
        printf("grid %d x 256\n", grid);
        time("copy 8B/lane", 2.0 * n * 8, [&] { copy8<<<grid, 256>>>(a, b, n); });
        time("copy 16B/lane", 2.0 * n * 8, [&] { copy16<<<grid, 256>>>((double2 *) a, (double2 *) b, n / 2); });
        time("read 8B/lane", 1.0 * n * 8, [&] { read8<<<grid, 256>>>(a, b, n); });
        time("read 16B/lane", 1.0 * n * 8, [&] { read16<<<grid, 256>>>((double2 *) a, b, n / 2); });
        time("write 8B/lane", 1.0 * n * 8, [&] { write8<<<grid, 256>>>(b, n); });
        time("copy AoS 3x8B stride 24", 2.0 * (n / 3) * 24, [&] { copy_aos<<<grid, 256>>>(a, b, n / 3); });
    }
    return 0;
}
