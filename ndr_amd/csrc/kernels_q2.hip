// Degree-2 (27-node hexahedron) matrix-free stiffness apply and sensitivity: TensorProductSimulator<2,2,2>
// (reference: VoxelFEM/TensorProductSimulator.hh:905-952 applyK, :730-751 complianceGradient, instantiated with
// Degrees = 2,2,2; the reference leaves it unbound, VoxelFEM.cc:226-229).
//
// Node grid (2nx+1) x (2ny+1) x (2nz+1), element (i,j,k) owns nodes (2i+a, 2j+b, 2k+c), local index 9a+3b+c.
// Gather form, one launch per node class (parity of the node coordinate per axis): an even coordinate lies on an
// element boundary (two incident elements along that axis), an odd one is a mid node (one element).  Inside a class
// the local index of the node in each incident element is the same for every lane, so the K0 rows are wave-uniform
// (scalar loads).  First correct version: dense 81x81 reference matrix, fp64-FMA-bound (6561 FMA per voxel).
#include "vfem_internal.h"

namespace vfem {

struct DimsQ2 { int nx, ny, nz, NX, NY, NZ; };   // elements / nodes per dim (N = 2n + 1)

__global__ void __launch_bounds__(256) k_apply_q2(DimsQ2 d, const double *__restrict__ K0, const double *__restrict__ E,
                                                  const double *__restrict__ u, double *__restrict__ out, int px, int py, int pz) {
    // node (i, j, k) = (2 a + px, 2 b + py, 2 c + pz)
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y, a = blockIdx.z;
    const int i = 2 * a + px, j = 2 * b + py, k = 2 * c + pz;
    if (i >= d.NX || j >= d.NY || k >= d.NZ) return;
    double S0 = 0.0, S1 = 0.0, S2 = 0.0;
    const int nex = px ? 1 : 2, ney = py ? 1 : 2, nez = pz ? 1 : 2;
    for (int sx = 0; sx < nex; ++sx) {
        // mid node: element a, local 1; boundary node: elements a-1 (local 2) and a (local 0)
        const int ex = px ? a : a - 1 + sx, lx = px ? 1 : (sx ? 0 : 2);
        if (ex < 0 || ex >= d.nx) continue;
        for (int sy = 0; sy < ney; ++sy) {
            const int ey = py ? b : b - 1 + sy, ly = py ? 1 : (sy ? 0 : 2);
            if (ey < 0 || ey >= d.ny) continue;
            for (int sz = 0; sz < nez; ++sz) {
                const int ez = pz ? c : c - 1 + sz, lz = pz ? 1 : (sz ? 0 : 2);
                if (ez < 0 || ez >= d.nz) continue;
                const int ln = 9 * lx + 3 * ly + lz;
                const double Ee = E[((long long) ex * d.ny + ey) * d.nz + ez];
                const double *r0 = K0 + (3 * ln) * 81, *r1 = r0 + 81, *r2 = r1 + 81;
                double t0 = 0.0, t1 = 0.0, t2 = 0.0;
                for (int ma = 0; ma < 3; ++ma)
                    for (int mb = 0; mb < 3; ++mb) {
                        const long long rowbase = ((long long) (2 * ex + ma) * d.NY + (2 * ey + mb)) * d.NZ + 2 * ez;
#pragma unroll
                        for (int mc = 0; mc < 3; ++mc) {
                            const int m = 9 * ma + 3 * mb + mc;
                            const double *um = u + 3 * (rowbase + mc);
                            const double u0 = um[0], u1 = um[1], u2 = um[2];
                            t0 = fma(r0[3 * m], u0, fma(r0[3 * m + 1], u1, fma(r0[3 * m + 2], u2, t0)));
                            t1 = fma(r1[3 * m], u0, fma(r1[3 * m + 1], u1, fma(r1[3 * m + 2], u2, t1)));
                            t2 = fma(r2[3 * m], u0, fma(r2[3 * m + 1], u1, fma(r2[3 * m + 2], u2, t2)));
                        }
                    }
                S0 = fma(Ee, t0, S0); S1 = fma(Ee, t1, S1); S2 = fma(Ee, t2, S2);
            }
        }
    }
    const long long n = ((long long) i * d.NY + j) * d.NZ + k;
    out[3 * n] = S0; out[3 * n + 1] = S1; out[3 * n + 2] = S2;
}

void launch_apply_q2(int nx, int ny, int nz, const double *K0, const double *E, const double *u, double *out, hipStream_t s) {
    DimsQ2 d{nx, ny, nz, 2 * nx + 1, 2 * ny + 1, 2 * nz + 1};
    for (int cls = 0; cls < 8; ++cls) {
        const int px = (cls >> 2) & 1, py = (cls >> 1) & 1, pz = cls & 1;
        const int cx = (d.NX - 1 - px) / 2 + 1, cy = (d.NY - 1 - py) / 2 + 1, cz = (d.NZ - 1 - pz) / 2 + 1;
        dim3 blk(64, 4, 1), grd((cz + 63) / 64, (cy + 3) / 4, cx);
        k_apply_q2<<<grd, blk, 0, s>>>(d, K0, E, u, out, px, py, pz);
    }
    VFEM_HIP(hipGetLastError());
}

// g_e = -1/2 gamma rho^(gamma-1) (E0 - Emin) u_e^T K0 u_e, one wave per element (81 dofs over 64 lanes)
__global__ void __launch_bounds__(256) k_gradient_q2(DimsQ2 d, const double *__restrict__ K0, const double *__restrict__ rho,
                                                     double E0, double Emin, double gamma, const double *__restrict__ u,
                                                     double *__restrict__ g) {
    __shared__ double ue[4][81];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long) blockIdx.x * 4 + w;
    const long long ne = (long long) d.nx * d.ny * d.nz;
    if (e >= ne) return;
    const int ez = (int) (e % d.nz), ey = (int) ((e / d.nz) % d.ny), ex = (int) (e / ((long long) d.nz * d.ny));
    for (int q = lane; q < 81; q += 64) {
        const int m = q / 3, cc = q % 3, ma = m / 9, mb = (m / 3) % 3, mc = m % 3;
        ue[w][q] = u[3 * (((long long) (2 * ex + ma) * d.NY + (2 * ey + mb)) * d.NZ + (2 * ez + mc)) + cc];
    }
    __builtin_amdgcn_wave_barrier();
    double acc = 0.0;
    for (int r = lane; r < 81; r += 64) {
        double t = 0.0;
        for (int cidx = 0; cidx < 81; ++cidx) t = fma(K0[r * 81 + cidx], ue[w][cidx], t);
        acc = fma(ue[w][r], t, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) {
        const double r0 = rho[e];
        g[e] = -0.5 * gamma * pow(r0, gamma - 1.0) * (E0 - Emin) * acc;
    }
}

void launch_gradient_q2(int nx, int ny, int nz, const double *K0, const double *rho, double E0, double Emin, double gamma,
                        const double *u, double *g, hipStream_t s) {
    DimsQ2 d{nx, ny, nz, 2 * nx + 1, 2 * ny + 1, 2 * nz + 1};
    const long long ne = (long long) nx * ny * nz;
    k_gradient_q2<<<dim3((unsigned) ((ne + 3) / 4)), dim3(256), 0, s>>>(d, K0, rho, E0, Emin, gamma, u, g);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
