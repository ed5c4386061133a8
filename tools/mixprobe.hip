// What HBM lets a LINEAR streaming kernel do with the stiffness apply's byte mix.  The apply (k_apply_dma) moves, per launch at 512^3,
// 4.87 GB of loads and 3.26 GB of stores (PMC, profiles/r02_apply512_pmc.json): 1.5 bytes read per byte written.  This program streams
// the same volumes through the simplest possible kernels (16 bytes per lane, perfectly coalesced, grid-stride, nothing else) for a
// set of read : write ratios, and -- second table -- replays the apply's STORE geometry (11 row segments of 63 nodes per block and
// plane at a 12.3 KB pitch) under different block orders, to see whether the order in which tiles are dealt to the CUs matters.
//   hipcc -O3 --offload-arch=gfx950 -o tools/mixprobe tools/mixprobe.hip && tools/mixprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double d2_t __attribute__((ext_vector_type(2)));

// every thread handles pieces of 16 bytes: NR of them loaded (from NR different arrays), NW stored (to NW different arrays)
template <int NR, int NW>
__global__ void __launch_bounds__(256) k_mix(long long n16, const d2_t *__restrict__ a, const d2_t *__restrict__ b, const d2_t *__restrict__ c,
                                             d2_t *__restrict__ o0, d2_t *__restrict__ o1) {
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long) gridDim.x * 256) {
        d2_t v = {1.0, 2.0};
        if (NR >= 1) v += a[i];
        if (NR >= 2) v += b[i];
        if (NR >= 3) v += c[i];
        if (NW >= 1) o0[i] = v;
        if (NW >= 2) o1[i] = v * 2.0;
        if (NW == 0 && v[0] == -1.2345) o0[0] = v;       // keeps the loads
    }
}

struct Geo { int NX, NY, NZ, rows, planes_per_chunk, nchunk, ntz, nty, order; };
// the apply's store geometry, one block per CU (LDS requested), block -> (chunk, z tile, y tile) by `order`
__global__ void __launch_bounds__(768) k_store(Geo g, double *__restrict__ out) {
    const int lane = threadIdx.x, w = threadIdx.y;
    int bid = blockIdx.x, ch, tz, ty;
    if (g.order == 0) { ch = bid % g.nchunk; bid /= g.nchunk; tz = bid % g.ntz; ty = bid / g.ntz; }          // chunk fastest (as k_apply_dma)
    else if (g.order == 1) { tz = bid % g.ntz; bid /= g.ntz; ch = bid % g.nchunk; ty = bid / g.nchunk; }     // z tiles of a row adjacent
    else if (g.order == 2) { tz = bid % g.ntz; bid /= g.ntz; ty = bid % g.nty; ch = bid / g.nty; }           // a whole chunk before the next
    else { ty = bid % g.nty; bid /= g.nty; tz = bid % g.ntz; ch = bid / g.ntz; }                              // y tiles fastest
    const int z = tz * 63 + lane, y = ty * g.rows + w;
    const bool ok = lane < 63 && z < g.NZ && w < g.rows && y < g.NY;
    const int p0 = ch * g.planes_per_chunk;
    int p1 = p0 + g.planes_per_chunk;
    if (p1 > g.NX) p1 = g.NX;
    for (int i = p0; i < p1; ++i) {
        const long long n = ((long long) i * g.NY + y) * g.NZ + z;
        if (ok) { out[3 * n] = 1.0 + i; out[3 * n + 1] = 2.0 + lane; out[3 * n + 2] = 3.0 + w; }
        __syncthreads();
    }
}

int main() {
    const int NX = 513, NY = 513, NZ = 513;
    const long long ndoubles = (long long) NX * NY * NZ * 3;
    const size_t bytes = (size_t) ndoubles * 8;
    double *buf[5];
    for (int q = 0; q < 5; ++q) { CK(hipMalloc(&buf[q], bytes)); CK(hipMemset(buf[q], 0, bytes)); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, double moved, auto launch) {
        for (int r = 0; r < 3; ++r) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-86s %7.3f ms  %6.2f TB/s moved\n", name, ms, moved / ms / 1e9);
        fflush(stdout);
    };
    const long long n16 = ndoubles / 2;
    const d2_t *A = (const d2_t *) buf[0], *B = (const d2_t *) buf[1], *C = (const d2_t *) buf[2];
    d2_t *O0 = (d2_t *) buf[3], *O1 = (d2_t *) buf[4];
    printf("linear streams of 3.24 GB arrays (16 B per lane, grid-stride):\n");
    run("1 read, 0 writes", 1.0 * bytes, [&] { k_mix<1, 0><<<8192, 256>>>(n16, A, B, C, O0, O1); });
    run("0 reads, 1 write", 1.0 * bytes, [&] { k_mix<0, 1><<<8192, 256>>>(n16, A, B, C, O0, O1); });
    run("1 read, 1 write (copy)", 2.0 * bytes, [&] { k_mix<1, 1><<<8192, 256>>>(n16, A, B, C, O0, O1); });
    run("3 reads, 2 writes (the apply's mix, 1.5 : 1)", 5.0 * bytes, [&] { k_mix<3, 2><<<8192, 256>>>(n16, A, B, C, O0, O1); });
    run("2 reads, 1 write", 3.0 * bytes, [&] { k_mix<2, 1><<<8192, 256>>>(n16, A, B, C, O0, O1); });
    run("3 reads, 1 write", 4.0 * bytes, [&] { k_mix<3, 1><<<8192, 256>>>(n16, A, B, C, O0, O1); });
    printf("the apply's store geometry (11 x 63-node row segments per block and plane, 12.3 KB pitch), one block per CU, by block order:\n");
    hipFuncSetAttribute((const void *) k_store, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    const char *names[4] = {"x-chunk fastest (k_apply_dma's order)", "the nine z tiles of a row adjacent, then chunks", "z tiles, y tiles, then chunks (one x-slab at a time)",
                            "y tiles fastest"};
    for (int nchunk : {8, 16})
        for (int order = 0; order < 4; ++order) {
            Geo g{NX, NY, NZ, 11, (NX + nchunk - 1) / nchunk, nchunk, (NZ + 62) / 63, (NY + 10) / 11, order};
            char nm[160];
            snprintf(nm, sizeof nm, "%d x-chunks, %s", nchunk, names[order]);
            run(nm, 1.0 * bytes, [&] { k_store<<<dim3(g.nchunk * g.ntz * g.nty), dim3(64, 12, 1), 120 * 1024>>>(g, buf[3]); });
        }
    return 0;
}
