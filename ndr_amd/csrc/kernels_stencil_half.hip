// Level 1 (degree 1) with its Galerkin operator stored as HALF a 27-point block stencil.
//
// The level-1 operator is never materialised by default: every visit of a node re-forms sum_f E_f cK0[f] for its eight
// elements (5184 multiply-adds, kernels_mg.hip: k_gs_color_mf1_sym), because the whole stencil is 1944 B per node -- as a full
// stencil the sweep moves 33 GB at 512^3 and is slower than the arithmetic (6.5 against 5.2 ms, profiles/r03_level1_stored_vs_virtual.txt).
// The operator is symmetric, A[n,m] = A[m,n]^T: a node stores its diagonal block and the blocks towards its 13 lexicographically
// LATER neighbours only (14 x 72 B = 1008 B), and reads the blocks towards the 13 earlier ones out of THEIR records, transposed.
// Same colour-major, tile-major order as the full stencils (cm_index): the nodes of a colour are numbered row-major and cut into
// tiles of 64; a tile holds its 126 entries as 126 runs of 64 doubles.  For the 64 nodes of a wave an earlier neighbour at a fixed
// offset is one colour-local shift away, so those reads are runs of consecutive doubles too (broken only at row ends).
// MG.hh:604-669 (Galerkin matrices), TPS.hh:649-720 (block matrix), MG.hh:242-251 (the sweep reads a stored column).
#include "vfem_internal.h"
#include "device_utils.h"

namespace vfem {

namespace sh {
constexpr int EPN = 126;         // entries per node: 14 blocks x 9

__device__ __forceinline__ long long node(const Dims &d, int i, int j, int k) { return ((long long) i * d.NY + j) * d.NZ + k; }
// first entry of node (i, j, k); entry e lives 64 e doubles further
__device__ __forceinline__ long long base(const Dims &d, int i, int j, int k) {
    const int ci = i & 1, cj = j & 1, ck = k & 1;
    const long long nx[2] = {(d.NX + 1) >> 1, d.NX >> 1}, ny[2] = {(d.NY + 1) >> 1, d.NY >> 1}, nz[2] = {(d.NZ + 1) >> 1, d.NZ >> 1};
    long long before = 0;
    const int c = ci * 4 + cj * 2 + ck;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < c) before += (nx[(q >> 2) & 1] * ny[(q >> 1) & 1] * nz[q & 1] + 63) / 64 * 64;
    const long long q = ((long long) (i >> 1) * ny[cj] + (j >> 1)) * nz[ck] + (k >> 1);
    return EPN * before + (q >> 6) * (EPN * 64) + (q & 63);
}
}  // namespace sh

long long stencil_half_storage_doubles(const Dims &d) {
    const long long nx[2] = {(d.NX + 1) >> 1, d.NX >> 1}, ny[2] = {(d.NY + 1) >> 1, d.NY >> 1}, nz[2] = {(d.NZ + 1) >> 1, d.NZ >> 1};
    long long total = 0;
    for (int q = 0; q < 8; ++q) total += (nx[(q >> 2) & 1] * ny[(q >> 1) & 1] * nz[q & 1] + 63) / 64 * 64;
    return sh::EPN * total;
}

// blocks A[n, n + delta] for delta = 0 and the 13 later neighbours (nb = 13 .. 26 of the 27-point numbering) from the virtual
// level-1 operator: Ke_c(e) = sum_f Efine[child f of e] cK0[f]  (same sums, same order as k_stencil_build<1>)
__global__ void __launch_bounds__(256) k_stencil_half_build_mf1(Dims d, const double *__restrict__ K, const double *__restrict__ E,
                                                                double *__restrict__ St) {
    const long long gid = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= d.nn * 14) return;
    const long long n = gid % d.nn;
    const int s = (int) (gid / d.nn), nb = 13 + s;
    const int k = (int) (n % d.NZ), j = (int) ((n / d.NZ) % d.NY), i = (int) (n / ((long long) d.NZ * d.NY));
    const int off[3] = {nb / 9 - 1, (nb / 3) % 3 - 1, nb % 3 - 1};
    const int pos[3] = {i, j, k};
    const int nel[3] = {d.nx, d.ny, d.nz};
    double A[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) A[q] = 0.0;
    const int m3[3] = {i + off[0], j + off[1], k + off[2]};
    const bool inside = m3[0] >= 0 && m3[0] < d.NX && m3[1] >= 0 && m3[1] < d.NY && m3[2] >= 0 && m3[2] < d.NZ;
    if (inside) {
        const long long nyf = 2LL * d.ny, nzf = 2LL * d.nz;
        for (int sel = 0; sel < 8; ++sel) {
            int e3[3], ln = 0, lm = 0;
            bool ok = true;
            for (int dd = 0; dd < 3; ++dd) {
                const int bit = (sel >> (2 - dd)) & 1;
                int e;
                if (off[dd] == 0) e = pos[dd] - 1 + bit;
                else { if (bit) { ok = false; break; } e = (off[dd] > 0) ? pos[dd] : pos[dd] - 1; }
                if (e < 0 || e >= nel[dd]) { ok = false; break; }
                e3[dd] = e;
                ln = 2 * ln + (pos[dd] - e);
                lm = 2 * lm + (m3[dd] - e);
            }
            if (!ok) continue;
            for (int f = 0; f < 8; ++f) {
                const int fx = (f >> 2) & 1, fy = (f >> 1) & 1, fz = f & 1;
                const double Ef = E[((2LL * e3[0] + fx) * nyf + (2LL * e3[1] + fy)) * nzf + (2LL * e3[2] + fz)];
                const double *Kf = K + f * 576;
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) A[3 * r + c] = fma(Ef, Kf[(3 * ln + r) * 24 + 3 * lm + c], A[3 * r + c]);
            }
        }
    }
    const long long b0 = sh::base(d, i, j, k);
#pragma unroll
    for (int q = 0; q < 9; ++q) St[b0 + ((long long) s * 9 + q) * 64] = A[q];
}
void launch_stencil_half_from_mf1(const Dims &d, const double *cK0, const double *Efine, double *Sh, hipStream_t s) {
    const long long total = d.nn * 14;
    k_stencil_half_build_mf1<<<dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s>>>(d, cK0, Efine, Sh);
    VFEM_HIP(hipGetLastError());
}

// S += A[n, m] u_m for the neighbours of one x-plane (W = 0: i - 1, 1: i, 2: i + 1) of node (i, j, k); the centre plane also
// yields the diagonal block and the node's own value
template <int W>
__device__ __forceinline__ void half_plane(const Dims &d, const double *__restrict__ St, const double *__restrict__ u, int i, int j, int k,
                                           long long b0, double S[3], double M[9], double uself[3]) {
    static_for<9>([&](auto tc) {
        constexpr int nb = 9 * W + decltype(tc)::value, di = W - 1, dj = (nb / 3) % 3 - 1, dk = nb % 3 - 1;
        const int ii = i + di, jj = j + dj, kk = k + dk;
        const bool inside = ii >= 0 && ii < d.NX && jj >= 0 && jj < d.NY && kk >= 0 && kk < d.NZ;
        const int ic = ii < 0 ? 0 : (ii > d.NX - 1 ? d.NX - 1 : ii), jc = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
        const int kc = kk < 0 ? 0 : (kk > d.NZ - 1 ? d.NZ - 1 : kk);
        const long long m = sh::node(d, ic, jc, kc);
        const double u0 = u[3 * m], u1 = u[3 * m + 1], u2 = u[3 * m + 2];
        if constexpr (nb >= 13) {
            // later neighbour (or the node itself): the block is in this node's record; blocks towards nodes outside the grid are
            // stored as zeros
            const double *a = St + b0 + (long long) (nb - 13) * 9 * 64;
            double A[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) A[q] = a[(long long) q * 64];
#pragma unroll
            for (int r = 0; r < 3; ++r) S[r] += A[3 * r] * u0 + A[3 * r + 1] * u1 + A[3 * r + 2] * u2;
            if (nb == 13) {
#pragma unroll
                for (int q = 0; q < 9; ++q) M[q] = A[q];
                uself[0] = u0; uself[1] = u1; uself[2] = u2;
            }
        } else {
            // earlier neighbour m: A[n, m] = A[m, n]^T, and A[m, n] is m's block towards ITS later neighbour 26 - nb
            const double *a = St + sh::base(d, ic, jc, kc) + (long long) (13 - nb) * 9 * 64;
            double A[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) { const double v = a[(long long) q * 64]; A[q] = inside ? v : 0.0; }
#pragma unroll
            for (int r = 0; r < 3; ++r) S[r] += A[r] * u0 + A[3 + r] * u1 + A[6 + r] * u2;
        }
    });
}

// colour sweep, the 27 neighbour blocks of a node shared by three waves (one x-plane each), as k_gs_color_stencil_split
__global__ void __launch_bounds__(192) k_gs_color_stencil_half(Dims d, const double *__restrict__ St, double *__restrict__ u,
                                                               const double *__restrict__ b, const uint8_t *__restrict__ mask,
                                                               int cx, int cy, int cz, int forward) {
    __shared__ double part[2][3][64];
    const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int cntx = (d.NX - 1 - cx) / 2 + 1, cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
    const long long q0 = (long long) blockIdx.x * 64 + lane, total = (long long) cntx * cnty * cntz;
    const bool live = q0 < total;
    const long long q = live ? q0 : total - 1;
    const int iq = (int) (q / ((long long) cnty * cntz)), rem = (int) (q - (long long) iq * cnty * cntz), jq = rem / cntz;
    const int i = 2 * iq + cx, j = 2 * jq + cy, k = 2 * (rem - jq * cntz) + cz;
    const long long n = sh::node(d, i, j, k);
    const long long b0 = sh::base(d, i, j, k);
    double S[3] = {0.0, 0.0, 0.0}, M[9], uself[3] = {0.0, 0.0, 0.0};
    if (w == 0) half_plane<0>(d, St, u, i, j, k, b0, S, M, uself);
    else if (w == 1) half_plane<1>(d, St, u, i, j, k, b0, S, M, uself);
    else half_plane<2>(d, St, u, i, j, k, b0, S, M, uself);
    if (w != 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) part[w >> 1][c][lane] = S[c];
    }
    __syncthreads();
    if (w != 1 || !live) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) S[c] = (part[0][c][lane] + S[c]) + part[1][c][lane];       // x-planes in ascending order
    double bms[3], ud[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bms[c] = b[3 * n + c] - S[c];
    gs_solve(bms, M, mask[n], forward != 0, ud);
#pragma unroll
    for (int c = 0; c < 3; ++c) u[3 * n + c] = uself[c] + ud[c];
}
void launch_gs_sweep_stencil_half(const Dims &d, const double *Sh, double *u, const double *b, const uint8_t *mask,
                                  int forward, int xparity, int first, int count, hipStream_t s) {
    for (int ci = first; ci < first + count; ++ci) {
        const int lni = forward ? ci : 7 - ci;
        const int cx = ((lni >> 2) & 1) ^ (xparity & 1), cy = (lni >> 1) & 1, cz = lni & 1;
        if (cx > d.NX - 1) continue;
        const int cntx = (d.NX - 1 - cx) / 2 + 1, cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
        const long long cnt = (long long) cntx * cnty * cntz;
        k_gs_color_stencil_half<<<dim3((unsigned) ((cnt + 63) / 64)), dim3(64, 3, 1), 0, s>>>(d, Sh, u, b, mask, cx, cy, cz, forward);
    }
    VFEM_HIP(hipGetLastError());
}

// out = A u (RES = false) or zeroDirichlet(b - A u); three waves per 64 nodes of a colour as the sweep (all eight colours in one
// launch: blockIdx.y)
template <bool RES>
__global__ void __launch_bounds__(192) k_apply_stencil_half(Dims d, const double *__restrict__ St, const double *__restrict__ u,
                                                            const double *__restrict__ b, const uint8_t *__restrict__ mask,
                                                            double *__restrict__ out) {
    __shared__ double part[2][3][64];
    const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int col = blockIdx.y, cx = (col >> 2) & 1, cy = (col >> 1) & 1, cz = col & 1;
    if (cx > d.NX - 1 || cy > d.NY - 1 || cz > d.NZ - 1) return;
    const int cntx = (d.NX - 1 - cx) / 2 + 1, cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
    const long long q0 = (long long) blockIdx.x * 64 + lane, total = (long long) cntx * cnty * cntz;
    if ((long long) blockIdx.x * 64 >= total) return;                      // (block-uniform)
    const bool live = q0 < total;
    const long long q = live ? q0 : total - 1;
    const int iq = (int) (q / ((long long) cnty * cntz)), rem = (int) (q - (long long) iq * cnty * cntz), jq = rem / cntz;
    const int i = 2 * iq + cx, j = 2 * jq + cy, k = 2 * (rem - jq * cntz) + cz;
    const long long n = sh::node(d, i, j, k);
    const long long b0 = sh::base(d, i, j, k);
    double S[3] = {0.0, 0.0, 0.0}, M[9], uself[3];
    if (w == 0) half_plane<0>(d, St, u, i, j, k, b0, S, M, uself);
    else if (w == 1) half_plane<1>(d, St, u, i, j, k, b0, S, M, uself);
    else half_plane<2>(d, St, u, i, j, k, b0, S, M, uself);
    if (w != 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) part[w >> 1][c][lane] = S[c];
    }
    __syncthreads();
    if (w != 1 || !live) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) S[c] = (part[0][c][lane] + S[c]) + part[1][c][lane];
    if (RES) {
        const uint8_t m = mask ? mask[n] : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * n + c] = ((m >> c) & 1) ? 0.0 : b[3 * n + c] - S[c];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * n + c] = S[c];
    }
}
void launch_apply_stencil_half(const Dims &d, const double *Sh, const double *u, const double *b, const uint8_t *mask, int res, double *out,
                               hipStream_t s) {
    const long long most = (long long) ((d.NX + 1) / 2) * ((d.NY + 1) / 2) * ((d.NZ + 1) / 2);      // the largest colour
    const dim3 grd((unsigned) ((most + 63) / 64), 8, 1), blk(64, 3, 1);
    if (res) k_apply_stencil_half<true><<<grd, blk, 0, s>>>(d, Sh, u, b, mask, out);
    else     k_apply_stencil_half<false><<<grd, blk, 0, s>>>(d, Sh, u, b, mask, out);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
