// Production matrix-free stiffness apply for the finest level:  out = K(rho) u  with
// Ke = E_e * K0  (reference: TensorProductSimulator::applyK, VoxelFEM/TensorProductSimulator.hh:905-952).
//
// Why not the reference's form: K0*u_e costs 576 fp64 FMA per voxel against 56 B of HBM traffic, which
// is fp64-bound on gfx950 (~48 % of the HBM roofline at best).  A box voxel with an isotropic/orthotropic
// material is symmetric under the three axis reflections, so in the basis of 2x2x2 sum/difference
// (Hadamard) modes per displacement component K0 has only 45 non-zeros:
//        K0 = T^T Dm T,   T = H (x) H (x) H,  H = [[1,1],[-1,1]]
// (21 diagonal entries + 12 symmetric couplings; coefficients built on the host in vfem_sim::update_k0).
// Apply = adds, 45 multiply-adds, adds  (~185 fp64 instructions per voxel) => HBM-bound.
//
// Mapping: lanes run along z (contiguous axis), the 8 waves of a block along y, and the block marches
// along x.  The y/z part of the transform of a node plane ("face modes", 12 values per element column) is
// computed once per plane and kept in registers, so every u value is read from HBM once per tile and each
// plane's transform is shared by the two element layers that touch it.  The transposed transform
// accumulates per-face sums in registers across the two adjacent element layers and scatters to the 4
// nodes of the face through LDS: no global atomics, deterministic order.
//
// Tile: 8 x 64 element columns need 9 x 65 node columns and complete 7 x 63 node columns (one-sided
// overlap), i.e. 86 % of the lanes produce output.  blockIdx.x is the x-chunk: consecutive block ids are
// dealt round-robin to the 8 XCDs, so with a multiple of 8 chunks every XCD owns one slab of x-planes and
// the halo rows/columns shared by neighbouring tiles are served by that XCD's own L2.
#include "vfem_internal.h"

namespace vfem {

constexpr int TY = 8;          // element rows per tile (waves per block)
constexpr int TZ = 64;         // element columns per tile (lanes)
constexpr int ROW_D = 196;     // doubles per staged node row (65 nodes x 3 = 195 used)

struct DmArgs { double v[36]; };

template <int MODE>   // 0: out = K u   1: out = zeroDirichlet(b - K u)   2: out = zeroDirichlet(K u)
__global__ void __launch_bounds__(TY * TZ) k_apply_fast(Dims d, DmArgs dm, const double *__restrict__ E,
                                                        const double *__restrict__ u, const double *__restrict__ b,
                                                        const uint8_t *__restrict__ mask, double *__restrict__ out,
                                                        int planes_per_chunk) {
    __shared__ double su[(TY + 1) * ROW_D];
    __shared__ double sB[3 * TY * TZ];
    __shared__ double sC[3 * TY * TZ];
    __shared__ double sD[3 * TY * TZ];

    const int tz = threadIdx.x, ty = threadIdx.y;
    const int k0 = blockIdx.y * (TZ - 1) - 1;      // node column of lane 0
    const int j0 = blockIdx.z * (TY - 1) - 1;      // node row of wave 0
    const int p0 = blockIdx.x * planes_per_chunk;  // first output plane of this chunk
    int p1 = p0 + planes_per_chunk - 1;
    if (p1 > d.NX - 1) p1 = d.NX - 1;
    if (p0 > d.NX - 1) return;

    const int ej = j0 + ty, ek = k0 + tz;          // element column / node owned by this thread
    const bool elem_ok = ej >= 0 && ej < d.ny && ek >= 0 && ek < d.nz;
    const bool out_ok = ty >= 1 && tz >= 1 && ej < d.NY && ek < d.NZ;   // ej, ek >= 0 follows from ty, tz >= 1
    const long long plane = (long long) d.NY * d.NZ;

    // stage one node plane (9 rows x 195 doubles) into su; rows/columns outside the grid read as zero
    auto stage_plane = [&](int i) {
        auto load_row = [&](int r, int q) {
            const int jj = j0 + r;
            const int kk = k0 + q / 3;
            double v = 0.0;
            if (q < 195 && jj >= 0 && jj < d.NY && kk >= 0 && kk < d.NZ)
                v = u[3 * ((long long) i * plane + (long long) jj * d.NZ + k0) + q];
            if (q < 195) su[r * ROW_D + q] = v;
        };
        load_row(ty, tz);
        load_row(ty, tz + 64);
        load_row(ty, tz + 128);
        if (tz < 3) load_row(ty, tz + 192);
        if (ty < 4) {   // ninth row: split over waves 0..3
            const int q = ty * 64 + tz;
            load_row(TY, q);
        }
    };

    // y/z transform of the face (ty,tz) of the staged plane: f[2*py+pz][c]
    auto face_modes = [&](double f[4][3]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double a  = su[ty * ROW_D + 3 * tz + c];
            const double bb = su[ty * ROW_D + 3 * (tz + 1) + c];
            const double cc = su[(ty + 1) * ROW_D + 3 * tz + c];
            const double dd = su[(ty + 1) * ROW_D + 3 * (tz + 1) + c];
            const double s0 = a + bb, d0 = bb - a, s1 = cc + dd, d1 = dd - cc;
            f[0][c] = s0 + s1;   // py 0, pz 0
            f[1][c] = d0 + d1;   // py 0, pz 1
            f[2][c] = s1 - s0;   // py 1, pz 0
            f[3][c] = d1 - d0;   // py 1, pz 1
        }
    };

    // transposed y/z transform of the face sums + scatter to the 4 nodes of the face; returns own share
    auto scatter_face = [&](const double acc[4][3], double wa[3]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double p = acc[0][c] - acc[2][c], q = acc[1][c] - acc[3][c];
            const double r = acc[0][c] + acc[2][c], t = acc[1][c] + acc[3][c];
            wa[c] = p - q;                                  // node (ty,   tz)
            sB[(c * TY + ty) * TZ + tz] = p + q;            // node (ty,   tz+1)
            sC[(c * TY + ty) * TZ + tz] = r - t;            // node (ty+1, tz)
            sD[(c * TY + ty) * TZ + tz] = r + t;            // node (ty+1, tz+1)
        }
    };

    auto emit_plane = [&](int i, const double wa[3]) {
        if (!out_ok) return;
        const long long n = (long long) i * plane + (long long) ej * d.NZ + ek;
        double w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            w[c] = wa[c] + sB[(c * TY + ty) * TZ + tz - 1] + sC[(c * TY + ty - 1) * TZ + tz] +
                   sD[(c * TY + ty - 1) * TZ + tz - 1];
        if (MODE == 0) {
            out[3 * n] = w[0]; out[3 * n + 1] = w[1]; out[3 * n + 2] = w[2];
        } else {
            const uint8_t m = mask ? mask[n] : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double v = (MODE == 1) ? b[3 * n + c] - w[c] : w[c];
                out[3 * n + c] = ((m >> c) & 1) ? 0.0 : v;
            }
        }
    };

    double fold[4][3], carry[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) carry[q][c] = 0.0;

    const int i_start = p0 > 0 ? p0 - 1 : 0;
    const int i_end = p1 + 1 < d.NX - 1 ? p1 + 1 : d.NX - 1;
    stage_plane(i_start);
    __syncthreads();
    face_modes(fold);
    __syncthreads();

    for (int i = i_start + 1; i <= i_end; ++i) {
        stage_plane(i);
        const double Ee = elem_ok ? E[((long long) (i - 1) * d.ny + ej) * d.nz + ek] : 0.0;
        __syncthreads();
        double fnew[4][3];
        face_modes(fnew);
        // x stage: modes m[4*px + 2*py + pz][c] of element layer i-1
        double m[8][3];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                m[q][c] = fold[q][c] + fnew[q][c];
                m[4 + q][c] = fnew[q][c] - fold[q][c];
                fold[q][c] = fnew[q][c];
            }
        // q = Dm m  (diagonal + 12 symmetric couplings; same enumeration as vfem_sim::update_k0)
        double qv[8][3];
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int c = 0; c < 3; ++c) qv[p][c] = dm.v[3 * p + c] * m[p][c];
        {
            int idx = 24;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bb = a + 1; bb < 3; ++bb) {
                    const int t = 3 - a - bb;
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                        for (int type = 0; type < 2; ++type) {
                            const int ba = 1 << (2 - a), bbit = 1 << (2 - bb), bt = 1 << (2 - t);
                            const int pa = (type == 0 ? ba : bbit) | (pt ? bt : 0);
                            const int pb = (type == 0 ? bbit : ba) | (pt ? bt : 0);
                            const double v = dm.v[idx++];
                            qv[pa][a] = fma(v, m[pb][bb], qv[pa][a]);
                            qv[pb][bb] = fma(v, m[pa][a], qv[pb][bb]);
                        }
                }
        }
        // scale by the element modulus, transposed x stage, accumulate the face sums of plane i-1
        double acc[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double lo = Ee * qv[q][c], hi = Ee * qv[4 + q][c];
                acc[q][c] = carry[q][c] + (lo - hi);
                carry[q][c] = lo + hi;
            }
        double wa[3];
        scatter_face(acc, wa);
        __syncthreads();
        if (i - 1 >= p0) emit_plane(i - 1, wa);
    }
    if (p1 == d.NX - 1) {   // last plane of the grid: only the element layer below contributes
        double wa[3];
        __syncthreads();
        scatter_face(carry, wa);
        __syncthreads();
        emit_plane(d.NX - 1, wa);
    }
}

void launch_apply_fast(const Dims &d, const double *Dm_host, const double *E, const double *u, const double *b,
                       const uint8_t *mask, int mode, double *out, hipStream_t s) {
    DmArgs dm;
    for (int q = 0; q < 36; ++q) dm.v[q] = Dm_host[q];
    int nchunks = d.NX >= 64 ? 8 : (d.NX >= 16 ? 4 : 1);
    if (d.NX >= 1024) nchunks = 16;
    const int ppc = (d.NX + nchunks - 1) / nchunks;
    dim3 blk(TZ, TY, 1), grd((d.NX + ppc - 1) / ppc, (d.NZ + TZ - 2) / (TZ - 1), (d.NY + TY - 2) / (TY - 1));
    if (mode == 0)      k_apply_fast<0><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc);
    else if (mode == 1) k_apply_fast<1><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc);
    else                k_apply_fast<2><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
