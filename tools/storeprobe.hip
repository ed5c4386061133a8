// Store-shape probe for the stiffness apply (VERDICT r01, item 5 iii): what write bandwidth does the apply kernel's store
// geometry allow, and what would a padded / line-aligned layout of the result allow?
//
// The apply writes, per block and node plane, 11 row segments of 63 nodes (1512 B) of a [NX][NY][NZ][3] fp64 field whose rows are
// NZ * 24 B = 12312 B apart (NZ = 513): neither end of a segment falls on a 128-byte line and consecutive rows shift by 24 B
// against the line grid.  This program reproduces exactly that geometry (grid 8 x z-tiles x y-tiles, 12 waves per block, one
// block barrier per plane, fire-and-forget stores) with nothing else in the kernel, and varies
//   pitch  : nodes per row in memory (513 = the reference layout; 528 = rows that are whole lines, 99 x 128 B)
//   width  : nodes a wave writes per row (63 as the apply; 64 = 12 whole lines when the row is aligned)
//   vec    : 0 = three 8-byte stores per lane (as the apply), 1 = one 16-byte + one 8-byte store per lane
//   barrier: 1 = one __syncthreads per plane (as the apply), 0 = none
//   mix    : 1 = every wave also loads the same row segment of a second field (16-byte loads) before it stores
// Output: one line per case with the time per pass over the field and GB/s of bytes written (+ read).
//   hipcc -O3 --offload-arch=gfx950 -o tools/storeprobe tools/storeprobe.hip && tools/storeprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// round 4: `skew` = 1 keeps the REFERENCE layout (pitch 513) and lets the window a wave writes start on a 384-byte
// boundary of the address space instead: a node is 24 B, 16 nodes are three whole lines, and the row pitch 513 = 1 (mod 16), so the
// window of row (i, y) is shifted by s = (i NY + y) mod 16 nodes against the tile grid (a parallelogram tile): 64 nodes = 12 whole lines.
struct Geo { int NX, NY, NZ, pitch, width, rows, planes_per_chunk, skew; };

template <int VEC, int BARRIER, int MIX>
__global__ void __launch_bounds__(768) k_store(Geo g, const double *__restrict__ in, double *__restrict__ out) {
    const int lane = threadIdx.x, w = threadIdx.y;
    const int y = blockIdx.z * g.rows + w;
    const int z0 = blockIdx.y * g.width + lane;
    const bool ok0 = lane < g.width && w < g.rows && y < g.NY;
    const int p0 = blockIdx.x * g.planes_per_chunk;
    int p1 = p0 + g.planes_per_chunk;
    if (p1 > g.NX) p1 = g.NX;
    double acc = 0.0;
    for (int i = p0; i < p1; ++i) {
        const int z = g.skew ? z0 - (int) (((long long) i * g.NY + y) & 15) : z0;
        const bool ok = ok0 && z >= 0 && z < g.NZ;
        const long long n = ((long long) i * g.NY + y) * g.pitch + z;
        if (ok) {
            double v0 = 1.0 + i, v1 = 2.0 + lane, v2 = 3.0 + w;
            if (MIX) {
                const double *q = in + 3 * n;
                v0 += q[0]; v1 += q[1]; v2 += q[2];
            }
            if (VEC) {
                // 24 contiguous bytes per lane: one 16-byte and one 8-byte store (the row base is 8-byte aligned only)
                typedef double d2_t __attribute__((ext_vector_type(2)));
                d2_t a = {v0, v1};
                __builtin_memcpy(out + 3 * n, &a, 16);
                out[3 * n + 2] = v2;
            } else {
                out[3 * n] = v0; out[3 * n + 1] = v1; out[3 * n + 2] = v2;
            }
            acc += v0;
        }
        if (BARRIER) __syncthreads();
    }
    if (acc == -1.0) out[0] = acc;
}

__global__ void k_fill(double *out, long long n) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) out[i] = 1.0;
}

int main() {
    const int NX = 513, NY = 513, NZ = 513, PAD = 528;
    const size_t bytes = (size_t) NX * NY * PAD * 24;
    double *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 0, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, double moved, auto launch) {
        for (int r = 0; r < 3; ++r) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-78s %7.3f ms  %6.2f TB/s\n", name, ms, moved / ms / 1e9);
        fflush(stdout);
    };
    run("linear fill, 8 B per lane (calibration)", (double) NX * NY * NZ * 24, [&] { k_fill<<<8192, 256>>>(out, (long long) NX * NY * NZ * 3); });
    struct Case { const char *name; int pitch, width, vec, barrier, mix, skew; };
    const Case cases[] = {
        {"apply geometry: pitch 513, 63 nodes per wave row, 3 x 8 B, barrier per plane", 513, 63, 0, 1, 0},
        {"same without the barrier", 513, 63, 0, 0, 0},
        {"same with 16 B + 8 B stores", 513, 63, 1, 1, 0},
        {"64 nodes per wave row (pitch 513)", 513, 64, 0, 1, 0},
        {"padded rows: pitch 528 (whole lines), 63 nodes per wave row", 528, 63, 0, 1, 0},
        {"padded rows: pitch 528, 64 nodes per wave row (12 whole lines per segment)", 528, 64, 0, 1, 0},
        {"padded, 64 wide, 16 B + 8 B stores", 528, 64, 1, 1, 0},
        {"apply geometry + load of the same segment (1R + 1W)", 513, 63, 0, 1, 1},
        {"padded 528 / 64 wide + load of the same segment (1R + 1W)", 528, 64, 0, 1, 1},
        {"REFERENCE layout, 64-node windows on 384-byte boundaries (parallelogram tile)", 513, 64, 0, 1, 0, 1},
        {"same with 16 B + 8 B stores", 513, 64, 1, 1, 0, 1},
        {"same + load of the same segment (1R + 1W)", 513, 64, 0, 1, 1, 1},
    };
    for (const Case &c : cases) {
        Geo g{NX, NY, NZ, c.pitch, c.width, 11, (NX + 7) / 8, c.skew};
        const dim3 grd(8, (NZ + (c.skew ? 15 : 0) + c.width - 1) / c.width, (NY + g.rows - 1) / g.rows), blk(64, 12, 1);
        const double moved = (double) NX * NY * NZ * 24 * (c.mix ? 2 : 1);
        auto launch = [&] {
            if (c.vec == 0 && c.barrier == 1 && c.mix == 0) k_store<0, 1, 0><<<grd, blk>>>(g, in, out);
            else if (c.vec == 0 && c.barrier == 0 && c.mix == 0) k_store<0, 0, 0><<<grd, blk>>>(g, in, out);
            else if (c.vec == 1 && c.barrier == 1 && c.mix == 0) k_store<1, 1, 0><<<grd, blk>>>(g, in, out);
            else k_store<0, 1, 1><<<grd, blk>>>(g, in, out);
        };
        run(c.name, moved, launch);
    }
    // occupancy: the apply runs ONE block per CU (147 KB of LDS); the cases above let up to two blocks share a CU.  Repeat the
    // reference geometry with 120 KB of dynamic LDS requested so that only one block fits
    {
        Geo g{NX, NY, NZ, 513, 63, 11, (NX + 7) / 8, 0};
        const dim3 grd(8, (NZ + 62) / 63, (NY + 10) / 11), blk(64, 12, 1);
        hipFuncSetAttribute((const void *) k_store<0, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        run("apply geometry, one block per CU (120 KB LDS requested)", (double) NX * NY * NZ * 24, [&] { k_store<0, 1, 0><<<grd, blk, 120 * 1024>>>(g, in, out); });
        Geo gp{NX, NY, NZ, 528, 64, 11, (NX + 7) / 8, 0};
        const dim3 grdp(8, (NZ + 63) / 64, (NY + 10) / 11);
        run("padded 528 / 64 wide, one block per CU", (double) NX * NY * NZ * 24, [&] { k_store<0, 1, 0><<<grdp, blk, 120 * 1024>>>(gp, in, out); });
        Geo gs{NX, NY, NZ, 513, 64, 11, (NX + 7) / 8, 1};
        const dim3 grds(8, (NZ + 15 + 63) / 64, (NY + 10) / 11);
        run("reference layout, 64-node windows on 384-byte boundaries, one block per CU", (double) NX * NY * NZ * 24, [&] { k_store<0, 1, 0><<<grds, blk, 120 * 1024>>>(gs, in, out); });
    }
    return 0;
}
