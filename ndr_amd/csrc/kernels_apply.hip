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
// Tile: TY x 64 element columns need (TY+1) x 65 node columns and complete (TY-1) x 63 node columns
// (one-sided overlap), i.e. 82 % of the lanes produce output at TY = 6.  blockIdx.x is the x-chunk: consecutive block ids are
// dealt round-robin to the 8 XCDs, so with a multiple of 8 chunks every XCD owns one slab of x-planes and
// the halo rows/columns shared by neighbouring tiles are served by that XCD's own L2.
#include "vfem_internal.h"

namespace vfem {

constexpr int TY = 8;          // element rows per tile = waves per block (2 per SIMD, one block per CU)
constexpr int TZ = 64;         // element columns per tile (lanes)
constexpr int ROW_D = 196;     // doubles per staged node row (65 nodes x 3 = 195 used)
constexpr int NLD = 4;         // staged doubles per thread and plane: (TY+1)*195 = 1755 <= 4 * 512
constexpr int SU_SIZE = (TY + 1) * ROW_D;               // element 195 of every row is the dump slot of unused staging slots

struct DmArgs { double v[36]; };

template <int MODE, int WPS, int PD>   // MODE 0: out = K u   1: out = zeroDirichlet(b - K u)   2: out = zeroDirichlet(K u); PD = planes in flight
__global__ void __launch_bounds__(TY * TZ, WPS == 1 ? 2 : WPS) k_apply_fast(Dims d, DmArgs dm, const double *__restrict__ E,
                                                        const double *__restrict__ u, const double *__restrict__ b,
                                                        const uint8_t *__restrict__ mask, double *__restrict__ out,
                                                        int planes_per_chunk, int store_mode) {
    __shared__ double su2[2 * SU_SIZE];
    __shared__ double so[TY * 192];               // per-wave output row staging (store_mode 2)          // staged node planes, double-buffered
    __shared__ double sS[2 * 9 * TY * TZ];        // face -> node scatter slots (B, C, D shares), double-buffered

    const int tz = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * TZ + tz;
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
    const int plane3 = 3 * d.NY * d.NZ;

    // Staging slots: the (TY+1) x 195 doubles of a node plane are dealt to the TY*64 threads, 4 per thread.
    // Offsets do not depend on the plane, so they are computed once.  Loads are unconditional: rows/columns
    // outside the grid read some in-bounds (finite) value instead, which only ever meets elements outside
    // the grid, and those carry modulus 0.
    int goff[NLD];            // offset (in doubles) from the start of a plane, clamped into the plane
    int loff[NLD];            // LDS offset (unused slots land in the dump area)
#pragma unroll
    for (int s4 = 0; s4 < NLD; ++s4) {
        const int L = s4 * (TY * TZ) + tid;
        const int r = L / 195, q = L - r * 195;
        const bool slot = r <= TY;
        int jj = j0 + r;
        jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
        int g = 3 * (jj * d.NZ + k0) + q;
        g = g < 0 ? 0 : (g > plane3 - 1 ? plane3 - 1 : g);
        goff[s4] = g;
        loff[s4] = slot ? r * ROW_D + q : (tid % (TY + 1)) * ROW_D + 195;
    }
    const int ejc = ej < 0 ? 0 : (ej > d.ny - 1 ? d.ny - 1 : ej);
    const int ekc = ek < 0 ? 0 : (ek > d.nz - 1 ? d.nz - 1 : ek);
    const int eoff = ejc * d.nz + ekc;
    const long long elayer = (long long) d.ny * d.nz;

    auto issue_loads = [&](int i, double v[NLD], double &Ev) {     // plane i of u, element layer i of E
        const double *up = u + 3 * (long long) i * plane;
#pragma unroll
        for (int s4 = 0; s4 < NLD; ++s4) v[s4] = up[goff[s4]];
        const int il = i < d.nx ? i : d.nx - 1;
        Ev = E[il * elayer + eoff];
    };
    auto stage_store = [&](const double v[NLD], int buf) {
        double *su = su2 + buf * SU_SIZE;
#pragma unroll
        for (int s4 = 0; s4 < NLD; ++s4) su[loff[s4]] = v[s4];
    };

    // y/z transform of the face (ty,tz) of the staged plane: f[2*py+pz][c]
    auto face_modes = [&](double f[4][3], int buf) {
        const double *su = su2 + buf * SU_SIZE;
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
    auto scatter_face = [&](const double acc[4][3], double wa[3], int buf) {
        double *sB = sS + buf * (9 * TY * TZ), *sC = sB + 3 * TY * TZ, *sD = sC + 3 * TY * TZ;
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

    auto emit_plane = [&](int i, const double wa[3], int buf) {
        if (MODE == 0 && store_mode == 2) {
            // transpose through a per-wave LDS row so that every store instruction writes 512 contiguous bytes
            const double *sB = sS + buf * (9 * TY * TZ), *sC = sB + 3 * TY * TZ, *sD = sC + 3 * TY * TZ;
            double *row = so + ty * 192;
            if (ty >= 1 && tz >= 1) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    row[3 * (tz - 1) + c] = wa[c] + sB[(c * TY + ty) * TZ + tz - 1] + sC[(c * TY + ty - 1) * TZ + tz] +
                                            sD[(c * TY + ty - 1) * TZ + tz - 1];
            }
            __builtin_amdgcn_wave_barrier();
            if (ty >= 1 && ej < d.NY) {
                const int kfirst = k0 + 1;                       // node column of row[0]
                int nvalid = d.NZ - kfirst; if (nvalid > TZ - 1) nvalid = TZ - 1;
                double *dst = out + 3 * ((long long) i * plane + (long long) ej * d.NZ + kfirst);
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const int q = tz + 64 * s3;
                    if (q < 3 * nvalid) dst[q] = row[q];
                }
            }
            __builtin_amdgcn_wave_barrier();
            return;
        }
        if (!out_ok) return;
        const double *sB = sS + buf * (9 * TY * TZ), *sC = sB + 3 * TY * TZ, *sD = sC + 3 * TY * TZ;
        const long long n = (long long) i * plane + (long long) ej * d.NZ + ek;
        double w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            w[c] = wa[c] + sB[(c * TY + ty) * TZ + tz - 1] + sC[(c * TY + ty - 1) * TZ + tz] +
                   sD[(c * TY + ty - 1) * TZ + tz - 1];
        if (MODE == 0) {
            if (store_mode == 0) { out[3 * n] = w[0]; out[3 * n + 1] = w[1]; out[3 * n + 2] = w[2]; }
            else if (store_mode == 1) { if (p0 < -5) out[3 * n] = w[0] + w[1] + w[2]; }
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

    // element layer i-1 between the staged plane i (in su2[buf]) and the previous one (in fold); leaves the
    // face sums of plane i-1 in the scatter slots sS[buf] and returns this thread's own share in wa
    auto process = [&](double Ee, int buf, double wa[3]) {
        if (!elem_ok) Ee = 0.0;
        if (WPS == 1) {   // diagnostic build: memory/LDS skeleton only (no transform arithmetic)
            const double *su = su2 + buf * SU_SIZE;
            double acc[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[q][c] = Ee + su[ty * ROW_D + 3 * tz + c] + su[(ty + 1) * ROW_D + 3 * (tz + 1) + c];
            scatter_face(acc, wa, buf);
            return;
        }
        double fnew[4][3];
        face_modes(fnew, buf);
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
            for (int c = 0; c < 3; ++c) qv[p][c] = (p == 0) ? 0.0 : dm.v[3 * p + c] * m[p][c];   // translations are null modes
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
                acc[q][c] = fma(Ee, qv[q][c] - qv[4 + q][c], carry[q][c]);
                carry[q][c] = Ee * (qv[q][c] + qv[4 + q][c]);
            }
        scatter_face(acc, wa, buf);
    };

    // Software pipeline, one barrier per plane.  While plane i is processed out of su2[buf], plane i+1 is
    // written to the other LDS buffer and the loads of planes i+2 .. i+1+PD are in flight in registers
    // (the kernel is latency-bound otherwise: ~18 KB per plane and CU, one block per CU).
    double R[PD][NLD], ER[PD];
    issue_loads(i_start, R[0], ER[0]);               // plane i_start, element layer i_start
    stage_store(R[0], 0);
    double Eim1 = 0.0, Ei = ER[0];
#pragma unroll
    for (int k = 0; k < PD; ++k)
        if (i_start + 1 + k <= i_end) issue_loads(i_start + 1 + k, R[k], ER[k]);   // R[k]: plane i_start+1+k (+ m PD)
    __syncthreads();
    face_modes(fold, 0);
    if (i_start + 1 <= i_end) {
        stage_store(R[0], 1);
        Eim1 = Ei; Ei = ER[0];
        if (i_start + 1 + PD <= i_end) issue_loads(i_start + 1 + PD, R[0], ER[0]);
    }
    __syncthreads();

    int buf = 1;     // LDS buffer holding plane i
    for (int i = i_start + 1; i <= i_end; i += PD) {
#pragma unroll
        for (int kk = 0; kk < PD; ++kk) {
            const int ii = i + kk;
            if (ii > i_end) break;
            constexpr int dummy = 0; (void) dummy;
            const int k = (kk + 1) % PD;             // register set holding plane ii+1
            double Enext = 0.0;
            if (ii + 1 <= i_end) {
                stage_store(R[k], buf ^ 1);
                Enext = ER[k];
                if (ii + 1 + PD <= i_end) issue_loads(ii + 1 + PD, R[k], ER[k]);
            }
            double wa[3];
            process(Eim1, buf, wa);
            __syncthreads();
            if (ii - 1 >= p0) emit_plane(ii - 1, wa, buf);
            buf ^= 1;
            Eim1 = Ei; Ei = Enext;
        }
    }
    if (p1 == d.NX - 1) {   // last plane of the grid: only the element layer below contributes
        double wa[3];
        scatter_face(carry, wa, buf);
        __syncthreads();
        emit_plane(d.NX - 1, wa, buf);
    }
}

#ifdef VFEM_ABLATION
int g_ablate_apply = 0, g_ablate_store = 0, g_ablate_mlp = 0;
#endif

void launch_apply_fast(const Dims &d, const double *Dm_host, const double *E, const double *u, const double *b,
                       const uint8_t *mask, int mode, double *out, hipStream_t s, int pd) {
    DmArgs dm;
    for (int q = 0; q < 36; ++q) dm.v[q] = Dm_host[q];
    int nchunks = d.NX >= 64 ? 8 : (d.NX >= 16 ? 4 : 1);
    if (d.NX >= 1024) nchunks = 16;
    const int ppc = (d.NX + nchunks - 1) / nchunks;
    dim3 blk(TZ, TY, 1), grd((d.NX + ppc - 1) / ppc, (d.NZ + TZ - 2) / (TZ - 1), (d.NY + TY - 2) / (TY - 1));
    const int st = ablate_store();
#ifdef VFEM_ABLATION
    if (ablate_apply()) {      // memory skeleton (wrong results, timing only)
        if (pd == 2)      k_apply_fast<0, 1, 2><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc, st);
        else if (pd == 3) k_apply_fast<0, 1, 3><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc, st);
        else              k_apply_fast<0, 1, 4><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc, st);
        VFEM_HIP(hipGetLastError());
        return;
    }
#endif
    if (mode == 0) {
        if (pd == 2)      k_apply_fast<0, 2, 2><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc, st);
        else if (pd == 3) k_apply_fast<0, 2, 3><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc, st);
        else              k_apply_fast<0, 2, 4><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc, st);
    }
    else if (mode == 1) k_apply_fast<1, 2, 2><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc, st);
    else                k_apply_fast<2, 2, 2><<<grd, blk, 0, s>>>(d, dm, E, u, b, mask, out, ppc, st);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
