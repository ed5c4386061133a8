// Dimension- and degree-generic voxel FEM path: TensorProductSimulator<p,..,p> / MultigridSolver<p,..,p> for
// N in {2,3}, p in {1,2} (reference templates: VoxelFEM/TensorProductSimulator.hh, VoxelFEM/MultigridSolver.hh).
// It serves every instantiation other than the tuned <1,1,1> path: the 2-D simulators of the reference's own
// CPU-runnable configuration (<1,1>, plane stress) and the degree-2 elements (<2,2>, <2,2,2>).
//
// One kernel family, "one wave per node": the wave gathers the node's incident elements (1..2^N), each lane owns
// dofs of the element vector, and the N rows of the element matrix that belong to the node are read as contiguous
// runs (K is symmetric, so rows are columns).  Level 0 uses E_e * K0, coarser levels the stored Galerkin element
// matrices Ke_e (MG.hh:604-669).  The same gather yields S = sum_e (K_e u_e)[node rows] and the diagonal block M, so
// the operator apply, the residual and the multicoloured block Gauss-Seidel (MG.hh:193-340) share it.  Deterministic:
// fixed lane ownership and a fixed xor-shuffle reduction tree.
#include "vfem_internal.h"
#include "q2_modes.h"


#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

namespace vfem {

struct GDims {
    int N, p;
    int ne[3], nn[3];
    int npe, ke;                     // nodes per element (p+1)^N, element matrix size N * npe
    long long nnodes, nelems;
};

static GDims make_gdims(int N, int p, const long long *ne) {
    GDims d{};
    d.N = N; d.p = p; d.npe = 1; d.nnodes = 1; d.nelems = 1;
    for (int a = 0; a < 3; ++a) {
        d.ne[a] = a < N ? (int) ne[a] : 1;
        d.nn[a] = a < N ? p * d.ne[a] + 1 : 1;
        if (a < N) { d.npe *= p + 1; d.nnodes *= d.nn[a]; d.nelems *= d.ne[a]; }
    }
    d.ke = N * d.npe;
    return d;
}

struct GWeights { double w[5][3]; };     // w[t][a]: coarse Lagrange basis a at fine offset t/(2p), t = 0..2p

// ------------------------------------------------------------------------------------------
// node gather
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ long long g_node_flat(const GDims &d, const int idx[3]) {
    long long n = idx[0];
    for (int a = 1; a < d.N; ++a) n = n * d.nn[a] + idx[a];
    return n;
}
template <int N>
__device__ __forceinline__ long long g_node_flat_n(const GDims &d, const int idx[3]) {
    long long n = idx[0];
#pragma unroll
    for (int a = 1; a < N; ++a) n = n * d.nn[a] + idx[a];
    return n;
}

// S[r] = sum over incident elements e, element dofs q of scale_e * K_e[N*ln + r][q] * u_e[q]   (complete in every lane)
// M[r][c] = sum_e scale_e * K_e[N*ln + r][N*ln + c]
template <int N, int p>       // dimension and degree at compile time: the index arithmetic below folds to shifts and constants
__device__ __forceinline__ void g_gather(const GDims &d, const double *__restrict__ K, long long kstride,
                                         const double *__restrict__ scale, const double *__restrict__ u, const int idx[3],
                                         int lane, double S[3], double M[9]) {
    constexpr int q1 = p + 1, ke = N * (N == 3 ? q1 * q1 * q1 : q1 * q1);
    int cnt[3] = {1, 1, 1}, el[3][2], lo[3][2];
    for (int a = 0; a < N; ++a) {
        const int r = idx[a] % p, e = idx[a] / p;
        if (r != 0) { cnt[a] = 1; el[a][0] = e; lo[a][0] = r; }
        else {
            int c = 0;
            if (e - 1 >= 0) { el[a][c] = e - 1; lo[a][c] = p; ++c; }
            if (e < d.ne[a]) { el[a][c] = e; lo[a][c] = 0; ++c; }
            cnt[a] = c;
        }
    }
    double sp[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < 9; ++q) M[q] = 0.0;
    // dof -> (local node offsets, component) of this lane's (up to two) element dofs
    int qn[2], qc[2], qoff[2][3];
    for (int t = 0; t < 2; ++t) {
        const int q = lane + 64 * t;
        qn[t] = q < ke ? q / N : -1; qc[t] = q % N;
        int m = qn[t] < 0 ? 0 : qn[t];
        for (int a = N - 1; a >= 0; --a) { qoff[t][a] = m % q1; m /= q1; }
    }
    for (int i0 = 0; i0 < cnt[0]; ++i0)
        for (int i1 = 0; i1 < cnt[1]; ++i1)
            for (int i2 = 0; i2 < (N == 3 ? cnt[2] : 1); ++i2) {
                const int sel[3] = {i0, i1, i2};
                long long e = 0; int ln = 0; int ebase[3] = {0, 0, 0};
                for (int a = 0; a < N; ++a) {
                    e = e * d.ne[a] + el[a][sel[a]];
                    ln = ln * q1 + lo[a][sel[a]];
                    ebase[a] = p * el[a][sel[a]];
                }
                const double sc = scale ? scale[e] : 1.0;
                const double *Kp = K + e * kstride + (long long) (N * ln) * ke;
                double kv[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};       // this lane's entries of the node's N rows
                for (int t = 0; t < 2; ++t) {
                    if (qn[t] < 0) continue;
                    int g[3];
                    for (int a = 0; a < N; ++a) g[a] = ebase[a] + qoff[t][a];
                    const double uv = sc * u[N * g_node_flat_n<N>(d, g) + qc[t]];
                    const int q = lane + 64 * t;
                    for (int r = 0; r < N; ++r) { kv[t][r] = Kp[r * ke + q]; sp[r] = fma(kv[t][r], uv, sp[r]); }
                }
                // diagonal block: entries N*ln + c of the same rows, already held by lane (N*ln + c) mod 64 -- the element and the
                // local node are the same in every lane of the wave, so the source lane is uniform (v_readlane, no further loads)
                for (int c = 0; c < N; ++c) {
                    const int qd = __builtin_amdgcn_readfirstlane(N * ln + c);
                    for (int r = 0; r < N; ++r) {
                        const double src = qd < 64 ? kv[0][r] : kv[1][r];
                        const unsigned long long bits = __double_as_longlong(src);
                        const unsigned lo = __builtin_amdgcn_readlane((unsigned) (bits & 0xffffffffull), qd & 63);
                        const unsigned hi = __builtin_amdgcn_readlane((unsigned) (bits >> 32), qd & 63);
                        const double kd = __longlong_as_double(((unsigned long long) hi << 32) | lo);
                        M[3 * r + c] = fma(sc, kd, M[3 * r + c]);
                    }
                }
            }
    for (int r = 0; r < 3; ++r) {
        double v = r < N ? sp[r] : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        S[r] = v;
    }
}

// The same gather for the nodes of one Gauss-Seidel colour of a 3-D grid: all of them have the same local index per axis, so
// the number of incident elements per axis (C = 1: mid node, 2: node on an element boundary) is a template parameter, the
// element loops unroll completely and the loads of all incident elements can be in flight together (rolled, every element was
// one more dependent round trip for the wave).  Elements that do not exist (grid faces) are read at the clamped neighbour
// with scale 0.
template <int p, int C0, int C1, int C2>
__device__ __forceinline__ void g_gather_fixed3(const GDims &d, const double *__restrict__ K, long long kstride,
                                                const double *__restrict__ scale, const double *__restrict__ u, const int idx[3],
                                                int lane, double S[3], double M[9]) {
    constexpr int N = 3, q1 = p + 1, ke = N * q1 * q1 * q1;
    constexpr int CN[3] = {C0, C1, C2};
    int el[3][2], lo[3][2];
    bool okc[3][2];
#pragma unroll
    for (int a = 0; a < N; ++a) {
        const int r = idx[a] % p, e = idx[a] / p;
        if (CN[a] == 1) { el[a][0] = e; lo[a][0] = r; okc[a][0] = true; el[a][1] = e; lo[a][1] = r; okc[a][1] = false; }
        else {
            okc[a][0] = e - 1 >= 0;      el[a][0] = okc[a][0] ? e - 1 : 0;            lo[a][0] = p;
            okc[a][1] = e < d.ne[a];     el[a][1] = okc[a][1] ? e : d.ne[a] - 1;      lo[a][1] = 0;
        }
    }
    double sp[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 9; ++q) M[q] = 0.0;
    int qn[2], qc[2], qoff[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = lane + 64 * t;
        qn[t] = q < ke ? q / N : -1; qc[t] = q % N;
        int m = qn[t] < 0 ? 0 : qn[t];
#pragma unroll
        for (int a = N - 1; a >= 0; --a) { qoff[t][a] = m % q1; m /= q1; }
    }
#pragma unroll
    for (int i0 = 0; i0 < C0; ++i0)
#pragma unroll
        for (int i1 = 0; i1 < C1; ++i1)
#pragma unroll
            for (int i2 = 0; i2 < C2; ++i2) {
                const int sel[3] = {i0, i1, i2};
                long long e = 0; int ln = 0; int ebase[3];
                bool valid = true;
#pragma unroll
                for (int a = 0; a < N; ++a) {
                    e = e * d.ne[a] + el[a][sel[a]];
                    ln = ln * q1 + lo[a][sel[a]];
                    ebase[a] = p * el[a][sel[a]];
                    valid = valid && okc[a][sel[a]];
                }
                const double sv = scale ? scale[e] : 1.0;
                const double sc = valid ? sv : 0.0;
                const double *Kp = K + e * kstride + (long long) (N * ln) * ke;
                double kv[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (qn[t] < 0) continue;
                    int g[3];
#pragma unroll
                    for (int a = 0; a < N; ++a) g[a] = ebase[a] + qoff[t][a];
                    const double uv = sc * u[N * g_node_flat_n<N>(d, g) + qc[t]];
                    const int q = lane + 64 * t;
#pragma unroll
                    for (int r = 0; r < N; ++r) { kv[t][r] = Kp[r * ke + q]; sp[r] = fma(kv[t][r], uv, sp[r]); }
                }
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    const int qd = __builtin_amdgcn_readfirstlane(N * ln + c);
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        const double src = qd < 64 ? kv[0][r] : kv[1][r];
                        const unsigned long long bits = __double_as_longlong(src);
                        const unsigned lo32 = __builtin_amdgcn_readlane((unsigned) (bits & 0xffffffffull), qd & 63);
                        const unsigned hi32 = __builtin_amdgcn_readlane((unsigned) (bits >> 32), qd & 63);
                        const double kd = __longlong_as_double(((unsigned long long) hi32 << 32) | lo32);
                        M[3 * r + c] = fma(sc, kd, M[3 * r + c]);
                    }
                }
            }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double v = sp[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        S[r] = v;
    }
}

// mode 0: out = K u;  1: out = zeroDirichlet(b - K u);  2: out = zeroDirichlet(K u)
template <int N, int p>
__global__ void __launch_bounds__(256) kg_apply(GDims d, const double *__restrict__ K, long long kstride,
                                                const double *__restrict__ scale, const double *__restrict__ u,
                                                const double *__restrict__ b, const uint8_t *__restrict__ mask, int mode,
                                                double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= d.nnodes) return;
    int idx[3] = {0, 0, 0};
    { long long m = n; for (int a = N - 1; a >= 0; --a) { idx[a] = (int) (m % d.nn[a]); m /= d.nn[a]; } }
    double S[3], M[9];
    g_gather<N, p>(d, K, kstride, scale, u, idx, lane, S, M);
    if (lane < N) {
        double v = S[lane];
        if (mode == 1) v = b[N * n + lane] - v;
        if (mode != 0 && mask && ((mask[n] >> lane) & 1)) v = 0.0;
        out[N * n + lane] = v;
    }
}

struct GColor { int start[3], inc[3], cnt[3]; long long total; };

// component-sequential solve of m_smoothNode (MG.hh:254-264) for the node the wave gathered
__device__ __forceinline__ void g_relax(const GDims &d, int N, long long n, const double S[3], const double M[9], double *__restrict__ u,
                                        const double *__restrict__ b, const uint8_t *__restrict__ mask, int forward) {
    const uint8_t dc = mask ? mask[n] : 0;
    double bms[3], diff[3] = {0.0, 0.0, 0.0};
    for (int r = 0; r < N; ++r) bms[r] = b[N * n + r] - S[r];
    for (int s = 0; s < N; ++s) {
        const int i = forward ? s : N - 1 - s;
        double acc = 0.0;
        for (int c = 0; c < N; ++c) acc += M[3 * i + c] * diff[c];
        const double fac = (double) (((dc >> i) & 1) == 0) / M[3 * i + i];
        diff[i] = (bms[i] - acc) * fac;
    }
    for (int r = 0; r < N; ++r) u[N * n + r] += diff[r];
}

template <int p, int C0, int C1, int C2>
__global__ void __launch_bounds__(256) kg_gs_color_fixed3(GDims d, GColor col, const double *__restrict__ K, long long kstride,
                                                          const double *__restrict__ scale, double *__restrict__ u,
                                                          const double *__restrict__ b, const uint8_t *__restrict__ mask, int forward) {
    const int lane = threadIdx.x & 63;
    const long long w = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= col.total) return;
    int idx[3] = {0, 0, 0};
    { long long m = w; for (int a = 2; a >= 0; --a) { idx[a] = col.start[a] + (int) (m % col.cnt[a]) * col.inc[a]; m /= col.cnt[a]; } }
    double S[3], M[9];
    g_gather_fixed3<p, C0, C1, C2>(d, K, kstride, scale, u, idx, lane, S, M);
    if (lane == 0) g_relax(d, 3, g_node_flat_n<3>(d, idx), S, M, u, b, mask, forward);
}

// one colour of smoothingMulticoloredGS (MG.hh:285-340) with m_smoothNode's component-sequential solve (MG.hh:254-264)
template <int N, int p>
__global__ void __launch_bounds__(256) kg_gs_color(GDims d, GColor col, const double *__restrict__ K, long long kstride,
                                                   const double *__restrict__ scale, double *__restrict__ u,
                                                   const double *__restrict__ b, const uint8_t *__restrict__ mask, int forward) {
    const int lane = threadIdx.x & 63;
    const long long w = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= col.total) return;
    int idx[3] = {0, 0, 0};
    { long long m = w; for (int a = N - 1; a >= 0; --a) { idx[a] = col.start[a] + (int) (m % col.cnt[a]) * col.inc[a]; m /= col.cnt[a]; } }
    double S[3], M[9];
    g_gather<N, p>(d, K, kstride, scale, u, idx, lane, S, M);
    if (lane == 0) {
        const long long n = g_node_flat_n<N>(d, idx);
        const uint8_t dc = mask ? mask[n] : 0;
        double bms[3], diff[3] = {0.0, 0.0, 0.0};
        for (int r = 0; r < N; ++r) bms[r] = b[N * n + r] - S[r];
        for (int s = 0; s < N; ++s) {
            const int i = forward ? s : N - 1 - s;
            double acc = 0.0;
            for (int c = 0; c < N; ++c) acc += M[3 * i + c] * diff[c];
            const double fac = (double) (((dc >> i) & 1) == 0) / M[3 * i + i];
            diff[i] = (bms[i] - acc) * fac;
        }
        for (int r = 0; r < N; ++r) u[N * n + r] += diff[r];
    }
}

__device__ __forceinline__ long long g_child(const GDims &f, const GDims &c, long long ec, int fi);

// The same triple product for 27-node elements with the interpolation applied axis by axis: phi_f = Px (x) Py (x) Pz with 3 x 3
// factors P_a[l][c] = w[l + 2 bit_a(f)][c], so a 27-vector is contracted in three passes of 27 x 3 multiply-adds instead of
// 27 x 27.  Thread t < 243 owns, in the first half, row i = t / 3 and component b = t % 3 of the child's matrix
// (T[i][(m,b)] = sum_q A[i][(q,b)] phi[q][m]) and, in the second half, component a = t / 81 and column j = t % 81
// (out[(n,a)][j] += sum_q phi[q][n] T[(q,a)][j]); its 27 results of the second half stay in registers across the eight children.
__device__ __forceinline__ void q2_contract27(double (&v)[27], const double (&P)[3][3][3]) {
    // P[axis][l][c]; v index = 9 qx + 3 qy + qz  ->  9 mx + 3 my + mz
    double t[27];
#pragma unroll
    for (int g = 0; g < 9; ++g)            // z: groups of 3 consecutive entries
#pragma unroll
        for (int c = 0; c < 3; ++c) t[3 * g + c] = v[3 * g] * P[2][0][c] + v[3 * g + 1] * P[2][1][c] + v[3 * g + 2] * P[2][2][c];
#pragma unroll
    for (int x = 0; x < 3; ++x)            // y: stride 3
#pragma unroll
        for (int z = 0; z < 3; ++z)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[9 * x + 3 * c + z] = t[9 * x + z] * P[1][0][c] + t[9 * x + 3 + z] * P[1][1][c] + t[9 * x + 6 + z] * P[1][2][c];
#pragma unroll
    for (int r = 0; r < 9; ++r)            // x: stride 9
#pragma unroll
        for (int c = 0; c < 3; ++c) t[9 * c + r] = v[r] * P[0][0][c] + v[9 + r] * P[0][1][c] + v[18 + r] * P[0][2][c];
#pragma unroll
    for (int q = 0; q < 27; ++q) v[q] = t[q];
}

__global__ void __launch_bounds__(256) kg_coarsen_next_q2(GDims f, GDims c, GWeights W, const double *__restrict__ Kef,
                                                          double *__restrict__ Kec) {
    extern __shared__ double g_sm[];
    constexpr int ke = 81, kk = ke * ke;
    double *A = g_sm, *T = g_sm + kk;
    const long long ec = blockIdx.x;
    const int t = threadIdx.x;
    const bool act = t < 243;
    const int i1 = t / 3, b1 = t % 3;          // first half
    const int a2 = t / 81, j2 = t % 81;        // second half
    double acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = 0.0;
    for (int fi = 0; fi < 8; ++fi) {
        const double *Kf = Kef + g_child(f, c, ec, fi) * kk;
        double P[3][3][3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const int bit = (fi >> ax) & 1;
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) P[ax][l][cc] = bit ? W.w[l + 2][cc] : W.w[l][cc];
        }
        __syncthreads();                        // the previous child's T is no longer read
        for (int q = t; q < kk; q += 256) A[q] = Kf[q];
        __syncthreads();
        if (act) {
            double v[27];
#pragma unroll
            for (int q = 0; q < 27; ++q) v[q] = A[i1 * ke + 3 * q + b1];
            q2_contract27(v, P);
#pragma unroll
            for (int m = 0; m < 27; ++m) T[i1 * ke + 3 * m + b1] = v[m];
        }
        __syncthreads();
        if (act) {
            double v[27];
#pragma unroll
            for (int q = 0; q < 27; ++q) v[q] = T[(3 * q + a2) * ke + j2];
            q2_contract27(v, P);
#pragma unroll
            for (int n = 0; n < 27; ++n) acc[n] += v[n];
        }
    }
    if (act) {
#pragma unroll
        for (int n = 0; n < 27; ++n) Kec[ec * kk + (3 * n + a2) * ke + j2] = acc[n];
    }
}

// ------------------------------------------------------------------------------------------
// grid transfers (MG.hh:116-161): thread per node of the level written
// ------------------------------------------------------------------------------------------
// `xs` (slab hierarchies): local fine plane = 2 * (local coarse plane) + xs along axis 0 -- the two local grids of a rank need
// not start at nested positions; xs <= 0 and every fine plane lies inside the coarse local grid.
__global__ void __launch_bounds__(256) kg_prolong(GDims f, GDims c, GWeights W, const double *__restrict__ cv,
                                                  double *__restrict__ fv, int accumulate, int xs) {
    const long long n = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= f.nnodes) return;
    const int N = f.N, p = f.p;
    int e[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    { long long m = n; for (int a = N - 1; a >= 0; --a) { int i = (int) (m % f.nn[a]); m /= f.nn[a];
        if (a == 0) i -= xs;
        int ee = i / (2 * p); if (ee > c.ne[a] - 1) ee = c.ne[a] - 1; e[a] = ee; t[a] = i - 2 * p * ee; } }
    double acc[3] = {0.0, 0.0, 0.0};
    const int q1 = p + 1;
    const int n2 = N == 3 ? q1 : 1;
    for (int a0 = 0; a0 < q1; ++a0) {
        const double w0 = W.w[t[0]][a0];
        if (w0 == 0.0) continue;
        for (int a1 = 0; a1 < q1; ++a1) {
            const double w1 = w0 * W.w[t[1]][a1];
            if (w1 == 0.0) continue;
            for (int a2 = 0; a2 < n2; ++a2) {
                const double w2 = N == 3 ? w1 * W.w[t[2]][a2] : w1;
                if (w2 == 0.0) continue;
                int g[3] = {p * e[0] + a0, p * e[1] + a1, p * e[2] + a2};
                const long long cn = g_node_flat(c, g);
                for (int r = 0; r < N; ++r) acc[r] = fma(w2, cv[N * cn + r], acc[r]);
            }
        }
    }
    for (int r = 0; r < N; ++r) fv[N * n + r] = accumulate ? fv[N * n + r] + acc[r] : acc[r];
}

__device__ __forceinline__ double g_restrict_weight(const GWeights &W, int p, int nce, int I, int i_f) {
    int e = i_f / (2 * p); if (e > nce - 1) e = nce - 1;
    const int t = i_f - 2 * p * e, a = I - p * e;
    return (a < 0 || a > p) ? 0.0 : W.w[t][a];
}

// (coarse nodes whose fine support leaves the local fine grid -- ghost planes of a slab -- receive partial sums; the driver
// never reads them)
__global__ void __launch_bounds__(256) kg_restrict(GDims f, GDims c, GWeights W, const double *__restrict__ fv,
                                                   double *__restrict__ cv, int xs) {
    const long long n = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= c.nnodes) return;
    const int N = f.N, p = f.p, R = 2 * p - 1;
    int I[3] = {0, 0, 0};
    { long long m = n; for (int a = N - 1; a >= 0; --a) { I[a] = (int) (m % c.nn[a]); m /= c.nn[a]; } }
    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {                 // axis 0 in the coarse grid's doubled coordinates (fine local = that + xs)
        const int sh = a == 0 ? xs : 0;
        lo[a] = a < N ? max(-sh, 2 * I[a] - R) : 0;
        hi[a] = a < N ? min(f.nn[a] - 1 - sh, 2 * I[a] + R) : 0;
    }
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i0 = lo[0]; i0 <= hi[0]; ++i0) {
        const double w0 = g_restrict_weight(W, p, c.ne[0], I[0], i0);
        if (w0 == 0.0) continue;
        for (int i1 = lo[1]; i1 <= hi[1]; ++i1) {
            const double w1 = w0 * g_restrict_weight(W, p, c.ne[1], I[1], i1);
            if (w1 == 0.0) continue;
            for (int i2 = lo[2]; i2 <= hi[2]; ++i2) {
                const double w2 = N == 3 ? w1 * g_restrict_weight(W, p, c.ne[2], I[2], i2) : w1;
                if (w2 == 0.0) continue;
                const int g[3] = {i0 + xs, i1, i2};
                const long long fn = g_node_flat(f, g);
                for (int r = 0; r < N; ++r) acc[r] = fma(w2, fv[N * fn + r], acc[r]);
            }
        }
    }
    for (int r = 0; r < N; ++r) cv[N * n + r] = acc[r];
}

// ------------------------------------------------------------------------------------------
// The same transfers axis by axis for 3-D grids: the interpolation weights are products of one-dimensional weights, so the
// restriction of a level is three passes (z, y, x) over shrinking intermediates and the prolongation three passes (x, y, z) over
// growing ones -- 7 / 3 reads per output value instead of up to 343 / 27, all but the z pass with unit-stride lanes.  (The
// single-pass kernels above took 7.2 and 6.8 ms between the two finest degree-2 levels at 256^3.)
// in: [n0][n1][n2][3] with the transferred axis of extent nin; out: the same with extent nout.  stride = nodes between
// neighbours along the axis, `inner` = nodes below it (product of the faster extents).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kg_restrict_axis(GWeights W, int p, int nce, int nin, int nout, long long inner, long long total_out,
                                                        const double *__restrict__ in, double *__restrict__ out, int xs) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;          // one (node, component) of the output
    if (t >= total_out) return;
    const long long in3 = inner * 3;
    const long long lo = t % in3, I = (t / in3) % nout, hi = t / (in3 * nout);
    const int R = 2 * p - 1;
    const int a = max(-xs, 2 * (int) I - R), b = min(nin - 1 - xs, 2 * (int) I + R);
    double acc = 0.0;
    for (int i = a; i <= b; ++i) {
        const double w = g_restrict_weight(W, p, nce, (int) I, i);
        if (w != 0.0) acc = fma(w, in[(hi * nin + (i + xs)) * in3 + lo], acc);
    }
    out[t] = acc;
}
__global__ void __launch_bounds__(256) kg_prolong_axis(GWeights W, int p, int nce, int nin, int nout, long long inner, long long total_out,
                                                       const double *__restrict__ in, double *__restrict__ out, int accumulate, int xs) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_out) return;
    const long long in3 = inner * 3;
    const long long lo = t % in3, i = (t / in3) % nout, hi = t / (in3 * nout);
    const int ii = (int) i - xs;
    int e = ii / (2 * p); if (e > nce - 1) e = nce - 1;
    const int tt = ii - 2 * p * e;
    double acc = 0.0;
    for (int a = 0; a <= p; ++a) {
        const double w = W.w[tt][a];
        if (w != 0.0) acc = fma(w, in[(hi * nin + (p * e + a)) * in3 + lo], acc);
    }
    out[t] = accumulate ? out[t] + acc : acc;
}

// ------------------------------------------------------------------------------------------
// Galerkin element matrices (buildPESCoarse, MG.hh:604-669)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ long long g_child(const GDims &f, const GDims &c, long long ec, int fi) {
    int idx[3] = {0, 0, 0};
    { long long m = ec; for (int a = c.N - 1; a >= 0; --a) { idx[a] = (int) (m % c.ne[a]); m /= c.ne[a]; } }
    long long e = 0;
    for (int a = 0; a < c.N; ++a) e = e * f.ne[a] + 2 * idx[a] + ((fi >> a) & 1);
    return e;
}

// Level 2 of the degree-2 hierarchy straight from the fine moduli when level 1 is virtual: Ke2 = sum_g I_g^T (sum_f E_{g,f} cK0[f]) I_g
//   = sum_{g,f} E_{g,f} c2K0[g][f],   c2K0[g][f] = I_g^T cK0[f] I_g  (64 constant 81 x 81 matrices, built on the host).
// A lane owns one level-2 element and keeps its 64 fine moduli in registers; the 64 constants of an entry q are the same for
// every lane and arrive by scalar loads (table laid out [q][64]); 64 multiply-adds per entry, the entry stored per lane.
// Replaces forming level-1 matrices in a scratch buffer and coarsening them (206 of 3200 ms of a 256^3 solve).
__global__ void __launch_bounds__(256) kg_coarsen_level2_q2(GDims f0, GDims c2, const double *__restrict__ tab /* [6561][64] */,
                                                            const double *__restrict__ Ef, double *__restrict__ Kec) {
    const long long e2 = (long long) blockIdx.x * 256 + threadIdx.x;
    const bool live = e2 < c2.nelems;
    const long long ee = live ? e2 : c2.nelems - 1;
    const int z2 = (int) (ee % c2.ne[2]), y2 = (int) ((ee / c2.ne[2]) % c2.ne[1]), x2 = (int) (ee / ((long long) c2.ne[2] * c2.ne[1]));
    double E[64];                                            // index 8 g + f, child bits as g_child: bit a = upper half along axis a (bit 0: x)
#pragma unroll
    for (int q = 0; q < 64; ++q) {
        const int g = q >> 3, f = q & 7;
        const int fx = 4 * x2 + 2 * (g & 1) + (f & 1), fy = 4 * y2 + 2 * ((g >> 1) & 1) + ((f >> 1) & 1), fz = 4 * z2 + 2 * ((g >> 2) & 1) + ((f >> 2) & 1);
        E[q] = Ef[((long long) fx * f0.ne[1] + fy) * f0.ne[2] + fz];
    }
    double *dst = Kec + (size_t) ee * 6561;
    for (int q = 0; q < 6561; ++q) {
        const double *c = tab + (size_t) q * 64;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int t = 0; t < 64; t += 2) { acc0 = fma(E[t], c[t], acc0); acc1 = fma(E[t + 1], c[t + 1], acc1); }
        if (live) dst[q] = acc0 + acc1;
    }
}

// level 1: Ke_c = sum_f E_f * cK0[f]
__global__ void __launch_bounds__(256) kg_coarsen_first(GDims f, GDims c, const double *__restrict__ cK0,
                                                        const double *__restrict__ Ef, double *__restrict__ Kec) {
    const long long ec = blockIdx.x;
    const int kk = c.ke * c.ke, nch = 1 << c.N;
    double E[8];
    for (int fi = 0; fi < nch; ++fi) E[fi] = Ef[g_child(f, c, ec, fi)];
    for (int q = threadIdx.x; q < kk; q += blockDim.x) {
        double v = 0.0;
        for (int fi = 0; fi < nch; ++fi) v = fma(E[fi], cK0[(long long) fi * kk + q], v);
        Kec[ec * kk + q] = v;
    }
}

// deeper levels: Ke_c = sum_f I_f^T Ke_f I_f with I_f = phi_f (x) Id_N; one block per coarse element
template <int N, int p>      // dimension and degree at compile time (entry indices fold to constants); weights of the child in LDS
__global__ void __launch_bounds__(256) kg_coarsen_next(GDims f, GDims c, const double *__restrict__ phi,
                                                       const double *__restrict__ Kef, double *__restrict__ Kec) {
    extern __shared__ double g_sm[];
    constexpr int q1 = p + 1, npe = N == 3 ? q1 * q1 * q1 : q1 * q1, ke = N * npe, kk = ke * ke, nch = 1 << N;
    double *A = g_sm, *T = g_sm + kk, *ph = g_sm + 2 * kk;            // ph[fine_n * npe + coarse_n]
    const long long ec = blockIdx.x;
    for (int q = threadIdx.x; q < kk; q += blockDim.x) Kec[ec * kk + q] = 0.0;
    for (int fi = 0; fi < nch; ++fi) {
        const double *Kf = Kef + g_child(f, c, ec, fi) * kk;
        __syncthreads();
        for (int q = threadIdx.x; q < kk; q += blockDim.x) A[q] = Kf[q];
        for (int q = threadIdx.x; q < npe * npe; q += blockDim.x) ph[q] = phi[(long long) fi * npe * npe + q];
        __syncthreads();
        // T[i][(m,b)] = sum_qn A[i][(qn,b)] ph[qn][m]
        for (int q = threadIdx.x; q < kk; q += blockDim.x) {
            const int i = q / ke, j = q % ke, m = j / N, bcomp = j % N;
            double v = 0.0;
            for (int qn = 0; qn < npe; ++qn) v = fma(A[i * ke + N * qn + bcomp], ph[qn * npe + m], v);
            T[q] = v;
        }
        __syncthreads();
        // out[(n,a)][j] += sum_pn ph[pn][n] T[(pn,a)][j]
        for (int q = threadIdx.x; q < kk; q += blockDim.x) {
            const int i = q / ke, j = q % ke, n = i / N, a = i % N;
            double v = 0.0;
            for (int pn = 0; pn < npe; ++pn) v = fma(ph[pn * npe + n], T[(N * pn + a) * ke + j], v);
            Kec[ec * kk + q] += v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// small vector kernels with N components per node
// ------------------------------------------------------------------------------------------
__global__ void kg_dirichlet(long long nn, int N, const uint8_t *__restrict__ mask, const double *__restrict__ vals,
                             double *__restrict__ u) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn * N) return;
    if ((mask[i / N] >> (i % N)) & 1) u[i] = vals ? vals[i] : 0.0;
}

__global__ void kg_dense_finish(long long n, int N, const uint8_t *__restrict__ mask, double *__restrict__ A) {
    const long long gid = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * n) return;
    const long long r = gid / n, c = gid % n;
    const bool fr = (mask[r / N] >> (r % N)) & 1, fc = (mask[c / N] >> (c % N)) & 1;
    if (fr || fc) { A[gid] = 0.0; return; }
    if (c < r) A[gid] = A[c * n + r];      // rocSOLVER "lower" (column-major) = row-major upper triangle
}

__global__ void __launch_bounds__(256) kg_gradient(GDims d, const double *__restrict__ K0, const double *__restrict__ rho,
                                                   double E0, double Emin, double gamma, const double *__restrict__ u,
                                                   double *__restrict__ g) {
    __shared__ double ue[4][81];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long) blockIdx.x * 4 + w;
    if (e >= d.nelems) return;
    const int N = d.N, q1 = d.p + 1, ke = d.ke;
    int eb[3] = {0, 0, 0};
    { long long m = e; for (int a = N - 1; a >= 0; --a) { eb[a] = d.p * (int) (m % d.ne[a]); m /= d.ne[a]; } }
    for (int q = lane; q < ke; q += 64) {
        int m = q / N, gg[3] = {0, 0, 0};
        for (int a = N - 1; a >= 0; --a) { gg[a] = eb[a] + m % q1; m /= q1; }
        ue[w][q] = u[N * g_node_flat(d, gg) + q % N];
    }
    __builtin_amdgcn_wave_barrier();
    double acc = 0.0;
    for (int r = lane; r < ke; r += 64) {
        double t = 0.0;
        for (int c = 0; c < ke; ++c) t = fma(K0[r * ke + c], ue[w][c], t);
        acc = fma(ue[w][r], t, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) g[e] = -0.5 * gamma * pow(rho[e], gamma - 1.0) * (E0 - Emin) * acc;
}

}  // namespace vfem

using namespace vfem;


// ------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------
struct vfem_gsim {
    GDims d;
    double h[3] = {1, 1, 1};
    double lambda = 0.0, mu = 0.5;                 // ETensor(1, 0), TPS.hh:1379
    double E0 = 1.0, Emin = 1e-9, gamma = 3.0;     // TPS.hh:1392-1394
    std::vector<double> K0;
    DevBuf<double> dK0, rho, E, dvals;
    DevBuf<double> red;                            // scratch of the compliance reduction
    DevBuf<double> q2tab;                          // degree-2 hexahedra: packed mode-space blocks (q2_modes.h)
    DevBuf<double> q2gstab;                        // ... and K0 regrouped for the finest-level sweep ordered by neighbour node
    int transfer_axis = 1;                         // vfem_gsim_set_option(17, v): 3-D grid transfers axis by axis (1) or in one pass (0)
    int q2_gs_impl = 2;                            // vfem_gsim_set_option(16, v): finest-level sweep 0 by element, 1 by neighbour node, 2 the same with the neighbour rows staged through LDS
    bool q2_fast = false;
    int q2_l1_virtual = 2;                         // vfem_gsim_set_option(14, v): level 1 of a degree-2 hierarchy 0 stored, 1 virtual, 2 by size
    int q2_impl = 0;                               // vfem_gsim_set_option(6, v): 0 marching kernel (mode space), 1 dense gather kernel (cross-check), 2 pencil kernel
    DevBuf<uint8_t> dmask;
    std::vector<uint8_t> hmask;
    // slab decomposition: element layers (axis 0) stored below / above the node grid; they only feed the Galerkin
    // element matrices of the ghost elements of coarser levels.  rho / E hold ex_lo + ne[0] + ex_hi layers.
    long long ex_lo = 0, ex_hi = 0;
    long long layer() const { return (long long) d.ne[1] * d.ne[2]; }
    long long stored_elems() const { return d.nelems + (ex_lo + ex_hi) * layer(); }
    const double *E_local() const { return E.p + ex_lo * layer(); }
    const double *rho_local() const { return rho.p + ex_lo * layer(); }
    void update_k0();
    void update_E(hipStream_t s) { launch_simp(stored_elems(), rho.p, E0, Emin, gamma, E.p, s); }
};

struct GLevel {
    GDims d;
    std::vector<uint8_t> hmask;
    DevBuf<uint8_t> mask;
    DevBuf<double> Ke, x, b, r;
    // slab hierarchies: element layers of Ke stored below / above the node grid (they feed the ghost elements of the next
    // level), and the plane shift of the transfers to the next FINER level (fine local plane = 2 * local plane + xs)
    long long pad_lo = 0, pad_hi = 0;
    int xs = 0;
    long long layer() const { return (long long) d.ne[1] * d.ne[2]; }
    long long stored_elems() const { return d.nelems + (pad_lo + pad_hi) * layer(); }
    const double *Ke_local() const { return Ke.p + (size_t) pad_lo * layer() * d.ke * d.ke; }
};

struct vfem_gmg {
    vfem_gsim *fine = nullptr;
    int L = 0;
    bool symmetric_gs = true, operators_valid = false;
    bool slab = false;                 // local hierarchy of one rank: levels 0..L hold fields and transfers, L only serves the transfers
    int first_active = 0;              // replicated coarse hierarchy: levels below hold no fields or operators
    int external_ke_level = -1;        // element matrices of this level were imported (vfem_gmg_import_level_ke)
    bool l1_virtual = false;           // degree 2: level 1 applies sum_f E_f cK0[f] on the fly, lv[1].Ke is not stored
    std::vector<GLevel> lv;
    GWeights W;
    DevBuf<double> cK0, phi, Ainv, pr, pd, pAd, ps, scal, scratch;
    DevBuf<double> tr1, tr2;           // intermediates of the axis-by-axis transfers
    DevBuf<double> c2tab;              // degree-2 hexahedra: c2K0[g][f] = I_g^T cK0[f] I_g as [entry][64] (level 2 straight from the moduli)
    DevBuf<double> l1tab;              // degree-2 hexahedra: cK0 regrouped for k_q2_level1, [ln][m][f][r][c]
    vfem::DenseWork dense;             // workspace of the coarsest-level inverse (dense_spd.hip)
};

static double lagrange1d(int p, int a, double x) {            // LagrangePolynomial.hh:9,42-56
    double v = 1.0;
    for (int j = 0; j <= p; ++j) if (j != a) v *= (x - (double) j / p) / ((double) a / p - (double) j / p);
    return v;
}
static double dlagrange1d(int p, int a, double x) {
    double s = 0.0;
    for (int k = 0; k <= p; ++k) {
        if (k == a) continue;
        double t = 1.0 / ((double) a / p - (double) k / p);
        for (int j = 0; j <= p; ++j) if (j != a && j != k) t *= (x - (double) j / p) / ((double) a / p - (double) j / p);
        s += t;
    }
    return s;
}

// Element_T::Stiffness (TPS.hh:127-140) by (p+1)-point Gauss quadrature per axis (TensorProductQuadrature<2p,...>)
void vfem_gsim::update_k0() {
    static const double g2[2] = {0.21132486540518711775, 0.78867513459481288225}, w2[2] = {0.5, 0.5};
    static const double g3[3] = {0.11270166537925831148, 0.5, 0.88729833462074168852}, w3[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
    const int N = d.N, p = d.p, q1 = p + 1, npe = d.npe, ke = d.ke;
    const double *gp = p == 1 ? g2 : g3, *gw = p == 1 ? w2 : w3;
    K0.assign((size_t) ke * ke, 0.0);
    double vol = 1.0;
    for (int a = 0; a < N; ++a) vol *= h[a];
    int nq = 1;
    for (int a = 0; a < N; ++a) nq *= q1;
    std::vector<double> G((size_t) npe * 3);
    for (int qi = 0; qi < nq; ++qi) {
        int qa[3] = {0, 0, 0};
        { int m = qi; for (int a = N - 1; a >= 0; --a) { qa[a] = m % q1; m /= q1; } }
        double w = vol;
        for (int a = 0; a < N; ++a) w *= gw[qa[a]];
        for (int n = 0; n < npe; ++n) {
            int l[3] = {0, 0, 0};
            { int m = n; for (int a = N - 1; a >= 0; --a) { l[a] = m % q1; m /= q1; } }
            for (int dd = 0; dd < N; ++dd) {
                double v = 1.0;
                for (int e = 0; e < N; ++e) v *= e == dd ? dlagrange1d(p, l[e], gp[qa[e]]) : lagrange1d(p, l[e], gp[qa[e]]);
                G[3 * n + dd] = v / h[dd];
            }
        }
        for (int n = 0; n < npe; ++n)
            for (int m = 0; m < npe; ++m) {
                double dot = 0.0;
                for (int dd = 0; dd < N; ++dd) dot += G[3 * n + dd] * G[3 * m + dd];
                for (int a = 0; a < N; ++a)
                    for (int b = 0; b < N; ++b)
                        K0[(size_t) (N * n + a) * ke + N * m + b] +=
                            w * (lambda * G[3 * n + a] * G[3 * m + b] + mu * G[3 * n + b] * G[3 * m + a] + (a == b ? mu * dot : 0.0));
            }
    }
    dK0.alloc(K0.size());
    VFEM_HIP(hipMemcpy(dK0.p, K0.data(), K0.size() * sizeof(double), hipMemcpyHostToDevice));
    q2_fast = false;
    if (N == 3 && p == 2) {
        std::vector<double> gst;
        build_q2_gs_table(K0.data(), gst);
        q2gstab.alloc(gst.size());
        VFEM_HIP(hipMemcpy(q2gstab.p, gst.data(), gst.size() * sizeof(double), hipMemcpyHostToDevice));
        // mode-space matrix Kt = T^-T K0 T^-1 (q2_modes.h); T^-1 per axis: u0 = (s - a)/2, u1 = m, u2 = (s + a)/2
        static const double Ti[3][3] = {{0.5, 0.0, -0.5}, {0.0, 1.0, 0.0}, {0.5, 0.0, 0.5}};      // [node][mode]
        std::vector<double> T3(27 * 27), A((size_t) 81 * 81), Kt((size_t) 81 * 81);
        for (int n = 0; n < 27; ++n)
            for (int m = 0; m < 27; ++m)
                T3[n * 27 + m] = Ti[n / 9][m / 9] * Ti[(n / 3) % 3][(m / 3) % 3] * Ti[n % 3][m % 3];
        for (int i = 0; i < 81; ++i)                       // A = K0 T^-1
            for (int m = 0; m < 27; ++m)
                for (int c = 0; c < 3; ++c) {
                    double v = 0.0;
                    for (int n = 0; n < 27; ++n) v += K0[(size_t) i * 81 + 3 * n + c] * T3[n * 27 + m];
                    A[(size_t) i * 81 + 3 * m + c] = v;
                }
        double scale = 0.0;
        for (int m = 0; m < 27; ++m)                       // Kt = T^-T A
            for (int c = 0; c < 3; ++c)
                for (int j = 0; j < 81; ++j) {
                    double v = 0.0;
                    for (int n = 0; n < 27; ++n) v += T3[n * 27 + m] * A[(size_t) (3 * n + c) * 81 + j];
                    Kt[(size_t) (3 * m + c) * 81 + j] = v;
                    scale = std::max(scale, std::fabs(v));
                }
        std::vector<int> cls(81, -1);
        for (int P = 0; P < 8; ++P)
            for (int j = 0; j < Q2C.n[P]; ++j) cls[Q2C.idx[P][j]] = P;
        double off = 0.0;
        for (int i = 0; i < 81; ++i)
            for (int j = 0; j < 81; ++j)
                if (cls[i] != cls[j]) off = std::max(off, std::fabs(Kt[(size_t) i * 81 + j]));
        if (off <= 1e-12 * scale) {
            std::vector<double> tab(Q2_TABLE_DOUBLES, 0.0);
            for (int P = 0; P < 8; ++P)
                for (int i = 0; i < Q2C.n[P]; ++i)
                    for (int j = 0; j < Q2C.n[P]; ++j)
                        tab[(size_t) (Q2C.rowbase[P] + i) * 12 + j] = Kt[(size_t) Q2C.idx[P][i] * 81 + Q2C.idx[P][j]];
            q2tab.alloc(tab.size());
            VFEM_HIP(hipMemcpy(q2tab.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
            q2_fast = true;
        }
    }
}

static inline hipStream_t GS(void *s) { return (hipStream_t) s; }
#define G_TRY try {
#define G_CATCH } catch (const std::exception &e) { vfem::set_error(e.what()); return 1; } return 0;

static void g_apply(const GDims &d, const double *K, long long kstride, const double *scale, const double *u,
                    const double *b, const uint8_t *mask, int mode, double *out, hipStream_t s) {
    const dim3 grd((unsigned) ((d.nnodes + 3) / 4)), blk(256);
    if (d.N == 3 && d.p == 2) kg_apply<3, 2><<<grd, blk, 0, s>>>(d, K, kstride, scale, u, b, mask, mode, out);
    else if (d.N == 3) kg_apply<3, 1><<<grd, blk, 0, s>>>(d, K, kstride, scale, u, b, mask, mode, out);
    else if (d.p == 2) kg_apply<2, 2><<<grd, blk, 0, s>>>(d, K, kstride, scale, u, b, mask, mode, out);
    else kg_apply<2, 1><<<grd, blk, 0, s>>>(d, K, kstride, scale, u, b, mask, mode, out);
    VFEM_HIP(hipGetLastError());
}

// level 0 normally applies E_e * K0; a hierarchy whose level-0 element matrices were imported (the replicated coarse part of a
// slab decomposition, created on the grid of its first level) reads them like any coarser level
static void level_op(const vfem_gmg *mg, int l, const double *&K, long long &kstride, const double *&scale) {
    if (l == 0 && mg->external_ke_level != 0) { K = mg->fine->dK0.p; kstride = 0; scale = mg->fine->E_local(); }
    else { K = mg->lv[l].Ke_local(); kstride = (long long) mg->lv[l].d.ke * mg->lv[l].d.ke; scale = nullptr; }
}

static void gmg_apply(vfem_gmg *mg, int l, const double *u, const double *b, int mode, double *out, hipStream_t s) {
    const vfem_gsim *sim = mg->fine;
    if (l == 0 && mg->external_ke_level != 0 && sim->d.N == 3 && sim->d.p == 2 && sim->q2_fast && sim->q2_impl != 1) {       // finest degree-2 level: mode-space kernels
        if (sim->q2_impl == 0) launch_apply_q2_march(sim->d.ne[0], sim->d.ne[1], sim->d.ne[2], sim->q2tab.p, sim->E_local(), u, out, s);
        else launch_apply_q2_pencil(sim->d.ne[0], sim->d.ne[1], sim->d.ne[2], sim->q2tab.p, sim->E_local(), u, out, s);
        if (mode != 0) launch_q2_residual_fix(sim->d.nnodes, b, mg->lv[0].mask.p, mode, out, s);
        return;
    }
    if (l == 1 && mg->l1_virtual) {
        const GDims &d = mg->lv[1].d;
        launch_apply_q2_level1(d.ne[0], d.ne[1], d.ne[2], mg->l1tab.p, sim->E.p, (int) (2 * mg->lv[1].pad_lo), u, b, mg->lv[1].mask.p, mode, out, s);
        return;
    }
    const double *K, *scale; long long ks;
    level_op(mg, l, K, ks, scale);
    g_apply(mg->lv[l].d, K, ks, scale, u, b, mg->lv[l].mask.p, mode, out, s);
}

// colours [first, first + count) of the sweep in visiting order (forward: local node index ascending, MG.hh:293-295); the
// colours of one x index are consecutive, which is what a slab driver exchanges halos between
static void gmg_smooth(vfem_gmg *mg, int l, double *u, const double *b, int forward, hipStream_t s, int first = 0, int count = -1) {
    const GDims &d = mg->lv[l].d;
    if (count < 0) count = 27;
    if (l == 0 && mg->external_ke_level != 0 && d.N == 3 && d.p == 2 && mg->fine->q2_impl != 1) {  // finest degree-2 level: thread per node
        if (mg->fine->q2_gs_impl >= 1)
            launch_gs_sweep_q2_level0_nodes(d.ne[0], d.ne[1], d.ne[2], mg->fine->q2gstab.p, mg->fine->E_local(), u, b, mg->lv[0].mask.p, forward, s, first, count,
                                            mg->fine->q2_gs_impl == 2);
        else
            launch_gs_sweep_q2_level0(d.ne[0], d.ne[1], d.ne[2], mg->fine->dK0.p, mg->fine->E_local(), u, b, mg->lv[0].mask.p, forward, s, first, count);
        return;
    }
    if (l == 1 && mg->l1_virtual) {
        launch_gs_sweep_q2_level1(d.ne[0], d.ne[1], d.ne[2], mg->l1tab.p, mg->fine->E.p, (int) (2 * mg->lv[1].pad_lo), u, b, mg->lv[1].mask.p, forward, s,
                                  first, count);
        return;
    }
    const double *K, *scale; long long ks;
    level_op(mg, l, K, ks, scale);
    int ncol = 1;
    for (int a = 0; a < d.N; ++a) ncol *= d.p + 1;
    for (int i = first; i < std::min(ncol, first + count); ++i) {
        const int lni = forward ? i : ncol - 1 - i;                   // MG.hh:293-295
        GColor col{};
        col.total = 1;
        int m = lni;
        for (int a = d.N - 1; a >= 0; --a) {
            const int la = m % (d.p + 1); m /= d.p + 1;
            const bool boundary = la == 0 || la == d.p;
            col.start[a] = la;
            col.inc[a] = (1 + (boundary ? 1 : 0)) * d.p;              // MG.hh:301-305
            col.cnt[a] = la > d.nn[a] - 1 ? 0 : (d.nn[a] - 1 - la) / col.inc[a] + 1;
            col.total *= col.cnt[a];
        }
        for (int a = d.N; a < 3; ++a) { col.start[a] = 0; col.inc[a] = 1; col.cnt[a] = 1; }
        if (col.total == 0) continue;
        const dim3 grd((unsigned) ((col.total + 3) / 4)), blk(256);
        const uint8_t *mk = mg->lv[l].mask.p;
        if (d.N == 3 && d.p == 2) {
            // incident elements per axis: 1 for a mid node (local index 1), 2 for a node on an element boundary
#define VFEM_GSF(A, B, C) kg_gs_color_fixed3<2, A, B, C><<<grd, blk, 0, s>>>(d, col, K, ks, scale, u, b, mk, forward)
            const int key = (col.start[0] == 1 ? 0 : 4) + (col.start[1] == 1 ? 0 : 2) + (col.start[2] == 1 ? 0 : 1);
            switch (key) {
                case 0: VFEM_GSF(1, 1, 1); break;
                case 1: VFEM_GSF(1, 1, 2); break;
                case 2: VFEM_GSF(1, 2, 1); break;
                case 3: VFEM_GSF(1, 2, 2); break;
                case 4: VFEM_GSF(2, 1, 1); break;
                case 5: VFEM_GSF(2, 1, 2); break;
                case 6: VFEM_GSF(2, 2, 1); break;
                default: VFEM_GSF(2, 2, 2);
            }
#undef VFEM_GSF
        }
        else if (d.N == 3) kg_gs_color<3, 1><<<grd, blk, 0, s>>>(d, col, K, ks, scale, u, b, mk, forward);
        else if (d.p == 2) kg_gs_color<2, 2><<<grd, blk, 0, s>>>(d, col, K, ks, scale, u, b, mk, forward);
        else kg_gs_color<2, 1><<<grd, blk, 0, s>>>(d, col, K, ks, scale, u, b, mk, forward);
    }
    VFEM_HIP(hipGetLastError());
}

static void gmg_restrict(vfem_gmg *mg, int l, const double *fine, double *coarse, hipStream_t s) {
    const GDims &f = mg->lv[l].d, &c = mg->lv[l + 1].d;
    const int xs = mg->lv[l + 1].xs;
    if (f.N == 3 && mg->fine->transfer_axis && f.nnodes > 100000) {   // axis by axis (z, y, x); small levels keep the single pass
        const long long n1 = (long long) f.nn[0] * f.nn[1] * c.nn[2], n2 = (long long) f.nn[0] * c.nn[1] * c.nn[2];
        mg->tr1.reserve((size_t) n1 * 3); mg->tr2.reserve((size_t) n2 * 3);
        auto grid = [](long long n) { return dim3((unsigned) ((n + 255) / 256)); };
        kg_restrict_axis<<<grid(n1 * 3), dim3(256), 0, s>>>(mg->W, f.p, c.ne[2], f.nn[2], c.nn[2], 1, n1 * 3, fine, mg->tr1.p, 0);
        kg_restrict_axis<<<grid(n2 * 3), dim3(256), 0, s>>>(mg->W, f.p, c.ne[1], f.nn[1], c.nn[1], c.nn[2], n2 * 3, mg->tr1.p, mg->tr2.p, 0);
        kg_restrict_axis<<<grid(c.nnodes * 3), dim3(256), 0, s>>>(mg->W, f.p, c.ne[0], f.nn[0], c.nn[0], (long long) c.nn[1] * c.nn[2], c.nnodes * 3,
                                                                  mg->tr2.p, coarse, xs);
    } else
        kg_restrict<<<dim3((unsigned) ((c.nnodes + 255) / 256)), dim3(256), 0, s>>>(f, c, mg->W, fine, coarse, xs);
    VFEM_HIP(hipGetLastError());
}
static void gmg_prolong(vfem_gmg *mg, int l, const double *coarse, double *fine, int accumulate, hipStream_t s) {
    const GDims &f = mg->lv[l].d, &c = mg->lv[l + 1].d;
    const int xs = mg->lv[l + 1].xs;
    if (f.N == 3 && mg->fine->transfer_axis && f.nnodes > 100000) {   // axis by axis (x, y, z)
        const long long n1 = (long long) f.nn[0] * c.nn[1] * c.nn[2], n2 = (long long) f.nn[0] * f.nn[1] * c.nn[2];
        mg->tr2.reserve((size_t) n1 * 3); mg->tr1.reserve((size_t) n2 * 3);
        auto grid = [](long long n) { return dim3((unsigned) ((n + 255) / 256)); };
        kg_prolong_axis<<<grid(n1 * 3), dim3(256), 0, s>>>(mg->W, f.p, c.ne[0], c.nn[0], f.nn[0], (long long) c.nn[1] * c.nn[2], n1 * 3, coarse, mg->tr2.p, 0, xs);
        kg_prolong_axis<<<grid(n2 * 3), dim3(256), 0, s>>>(mg->W, f.p, c.ne[1], c.nn[1], f.nn[1], c.nn[2], n2 * 3, mg->tr2.p, mg->tr1.p, 0, 0);
        kg_prolong_axis<<<grid(f.nnodes * 3), dim3(256), 0, s>>>(mg->W, f.p, c.ne[2], c.nn[2], f.nn[2], 1, f.nnodes * 3, mg->tr1.p, fine, accumulate, 0);
    } else
        kg_prolong<<<dim3((unsigned) ((f.nnodes + 255) / 256)), dim3(256), 0, s>>>(f, c, mg->W, coarse, fine, accumulate, xs);
    VFEM_HIP(hipGetLastError());
}
static void g_dirichlet(const GDims &d, const uint8_t *mask, const double *vals, double *u, hipStream_t s) {
    kg_dirichlet<<<dim3((unsigned) ((d.nnodes * d.N + 255) / 256)), dim3(256), 0, s>>>(d.nnodes, d.N, mask, vals, u);
    VFEM_HIP(hipGetLastError());
}
static void gmg_coarsest(vfem_gmg *mg, const double *b, double *x, hipStream_t s) {
    if (!mg->Ainv.p) throw Error("coarsest grid too large for the dense coarsest-level solve; use more coarsening levels");
    launch_gemv_sym((long long) mg->lv[mg->L].d.N * mg->lv[mg->L].d.nnodes, mg->Ainv.p, b, x, s);
}

// Galerkin element matrices of the coarse elements `c` from their 2^N children in `f` (buildPESCoarse, MG.hh:604-669);
// first: the children are E-scaled copies of K0 (src = moduli), otherwise src = the children's element matrices
static void g_coarsen(vfem_gmg *mg, const GDims &f, const GDims &c, bool first, const double *src, double *out, hipStream_t s) {
    const int N = c.N;
    const size_t kk = (size_t) c.ke * c.ke;
    if (c.nelems == 0) return;
    const dim3 grd((unsigned) c.nelems), blk(256);
    if (first) kg_coarsen_first<<<grd, blk, 0, s>>>(f, c, mg->cK0.p, src, out);
    else if (N == 3 && c.p == 2) {
        static bool attr2 = false;
        if (!attr2) { VFEM_HIP(hipFuncSetAttribute((const void *) kg_coarsen_next_q2, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 81 * 81 * 8)); attr2 = true; }
        kg_coarsen_next_q2<<<grd, blk, 2 * kk * sizeof(double), s>>>(f, c, mg->W, src, out);
    } else {
        const size_t lds = (2 * kk + (size_t) c.npe * c.npe) * sizeof(double);
        if (N == 3) kg_coarsen_next<3, 1><<<grd, blk, lds, s>>>(f, c, mg->phi.p, src, out);
        else if (c.p == 2) kg_coarsen_next<2, 2><<<grd, blk, lds, s>>>(f, c, mg->phi.p, src, out);
        else kg_coarsen_next<2, 1><<<grd, blk, lds, s>>>(f, c, mg->phi.p, src, out);
    }
    VFEM_HIP(hipGetLastError());
}
// dims of the stored element array of a level (slab hierarchies keep extra layers along axis 0)
static GDims stored_dims(const vfem_gmg *mg, int l) {
    const GLevel &lv = mg->lv[l];
    const long long lo = l == 0 ? mg->fine->ex_lo : lv.pad_lo, hi = l == 0 ? mg->fine->ex_hi : lv.pad_hi;
    const long long ne[3] = {lv.d.ne[0] + lo + hi, lv.d.ne[1], lv.d.ne[2]};
    return make_gdims(lv.d.N, lv.d.p, ne);
}

// Element matrices of `count` x-layers of level 2, starting at stored layer `first2`, when level 1 is virtual: straight from
// the fine moduli (kg_coarsen_level2_q2)
static void coarsen_through_virtual_level1(vfem_gmg *mg, long long first2, long long count, double *out, hipStream_t s) {
    const GDims d0 = stored_dims(mg, 0), d2 = stored_dims(mg, 2);
    if (count <= 0) return;
    const long long ne2[3] = {count, d2.ne[1], d2.ne[2]}, ne0[3] = {4 * count, d0.ne[1], d0.ne[2]};
    const GDims c2 = make_gdims(3, 2, ne2), c0 = make_gdims(3, 2, ne0);
    const long long layer0 = (long long) d0.ne[1] * d0.ne[2];
    kg_coarsen_level2_q2<<<dim3((unsigned) ((c2.nelems + 255) / 256)), dim3(256), 0, s>>>(c0, c2, mg->c2tab.p, mg->fine->E.p + (size_t) (4 * first2) * layer0, out);
    VFEM_HIP(hipGetLastError());
}

static void gmg_update(vfem_gmg *mg, hipStream_t s) {
    vfem_gsim *sim = mg->fine;
    const int N = sim->d.N;
    if (mg->first_active > 0 && mg->external_ke_level != mg->first_active)
        throw Error("the element matrices of the first active level have not been imported (vfem_gmg_import_level_ke)");
    const int l_first = mg->first_active > 0 ? mg->first_active + 1 : 1;
    const int l_last = mg->slab ? mg->L - 1 : mg->L;             // a slab hierarchy's last level only serves the transfers
    // level 1 stays virtual when it is an operator level that is not the coarsest one and level 0 applies E K0
    // (measured, tools/q2_level1_probe.py: from 64^3 fine elements on the on-the-fly form is the faster one -- CG-MG iterations/s
    // 43.3 against 42.1 at 64^3, 11.0 against 9.3 at 128^3, 2.19 against 1.54 at 256^3, where the stored matrices are 110 GB;
    // below, a launch has too few waves to hide its scalar-load latency)
    mg->l1_virtual = N == 3 && sim->d.p == 2 && mg->first_active == 0 && mg->external_ke_level != 0 && mg->L >= 2 &&
                     (sim->q2_l1_virtual == 1 ||
                      (sim->q2_l1_virtual == 2 && (double) stored_dims(mg, 1).nelems * 81 * 81 * sizeof(double) > 1.5e9));
    if (mg->l1_virtual) mg->lv[1].Ke.release();
    for (int l = l_first; l <= l_last; ++l) {
        GLevel &lv = mg->lv[l];
        const size_t kk = (size_t) lv.d.ke * lv.d.ke;
        const GDims cd = stored_dims(mg, l), fd = stored_dims(mg, l - 1);
        if (fd.ne[0] != 2 * cd.ne[0]) throw Error("stored element layers of consecutive levels must halve exactly");
        if (mg->l1_virtual && l == 1) continue;
        lv.Ke.alloc((size_t) cd.nelems * kk);
        if (mg->l1_virtual && l == 2) { coarsen_through_virtual_level1(mg, 0, cd.ne[0], lv.Ke.p, s); continue; }
        const bool from_moduli = l == 1 && mg->external_ke_level != 0;
        g_coarsen(mg, fd, cd, from_moduli, from_moduli ? sim->E.p : mg->lv[l - 1].Ke.p, lv.Ke.p, s);
    }
    if (mg->slab) { mg->operators_valid = true; return; }
    // coarsest level: assemble the dense matrix on the host (a few elements), invert with rocSOLVER
    GLevel &cl = mg->lv[mg->L];
    const GDims &d = cl.d;
    const long long n = (long long) N * d.nnodes;
    if (n > 40000) {
        // a grid that cannot be coarsened (odd element counts) and is too large for the dense factorisation can still be
        // solved by the unpreconditioned CG (mgSmoothingIterations = 0, MG.hh:476-479); any cycle on it throws
        if (mg->L > 0) throw Error("coarsest grid too large for the dense coarsest-level solve (" + std::to_string(n) + " dofs); use more coarsening levels");
        mg->Ainv.release();
        mg->operators_valid = true;
        return;
    }
    const size_t kk = (size_t) d.ke * d.ke;
    std::vector<double> Ke((size_t) d.nelems * kk);
    if (mg->L == 0 && mg->external_ke_level != 0) {
        std::vector<double> E((size_t) d.nelems);
        VFEM_HIP(hipMemcpyAsync(E.data(), sim->E_local(), E.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        VFEM_HIP(hipStreamSynchronize(s));
        for (long long e = 0; e < d.nelems; ++e)
            for (size_t q = 0; q < kk; ++q) Ke[e * kk + q] = E[e] * sim->K0[q];
    } else {
        VFEM_HIP(hipMemcpyAsync(Ke.data(), cl.Ke_local(), Ke.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        VFEM_HIP(hipStreamSynchronize(s));
    }
    std::vector<double> A((size_t) n * n, 0.0);
    const int q1 = d.p + 1;
    std::vector<long long> dofs(d.ke);
    for (long long e = 0; e < d.nelems; ++e) {
        int eb[3] = {0, 0, 0};
        { long long m = e; for (int a = N - 1; a >= 0; --a) { eb[a] = d.p * (int) (m % d.ne[a]); m /= d.ne[a]; } }
        for (int ln = 0; ln < d.npe; ++ln) {
            int m = ln, g[3] = {0, 0, 0};
            for (int a = N - 1; a >= 0; --a) { g[a] = eb[a] + m % q1; m /= q1; }
            long long node = g[0];
            for (int a = 1; a < N; ++a) node = node * d.nn[a] + g[a];
            for (int c = 0; c < N; ++c) dofs[N * ln + c] = N * node + c;
        }
        for (int i = 0; i < d.ke; ++i)
            for (int j = 0; j < d.ke; ++j) A[(size_t) dofs[i] * n + dofs[j]] += Ke[e * kk + (size_t) i * d.ke + j];
    }
    for (long long r = 0; r < n; ++r)
        if ((cl.hmask[r / N] >> (r % N)) & 1) {
            for (long long c = 0; c < n; ++c) { A[(size_t) r * n + c] = 0.0; A[(size_t) c * n + r] = 0.0; }
            A[(size_t) r * n + r] = 1.0;
        }
    mg->Ainv.alloc((size_t) n * n);
    VFEM_HIP(hipStreamSynchronize(s));
    VFEM_HIP(hipMemcpy(mg->Ainv.p, A.data(), A.size() * sizeof(double), hipMemcpyHostToDevice));       // (synchronous: the source is pageable host memory)
    dense_spd_inverse(n, mg->Ainv.p, mg->dense, s);      // own kernels, fixed summation order (dense_spd.hip)
    kg_dense_finish<<<dim3((unsigned) ((n * n + 255) / 256)), dim3(256), 0, s>>>(n, N, cl.mask.p, mg->Ainv.p);
    VFEM_HIP(hipGetLastError());
    VFEM_HIP(hipStreamSynchronize(s));
    mg->operators_valid = true;
}

// MG.hh:516-553
static void gmg_vcycle(vfem_gmg *mg, int l, int nsmooth, bool residual_system, hipStream_t s) {
    GLevel &L = mg->lv[l];
    if (l == mg->L) { gmg_coarsest(mg, L.b.p, L.x.p, s); return; }
    GLevel &C = mg->lv[l + 1];
    g_dirichlet(L.d, L.mask.p, (l == 0 && !residual_system) ? mg->fine->dvals.p : nullptr, L.x.p, s);
    for (int i = 0; i < nsmooth; ++i) gmg_smooth(mg, l, L.x.p, L.b.p, 1, s);
    gmg_apply(mg, l, L.x.p, L.b.p, 1, L.r.p, s);
    gmg_restrict(mg, l, L.r.p, C.b.p, s);
    C.x.zero(s);
    gmg_vcycle(mg, l + 1, nsmooth, true, s);
    gmg_prolong(mg, l, C.x.p, L.x.p, 1, s);
    for (int i = 0; i < nsmooth; ++i) gmg_smooth(mg, l, L.x.p, L.b.p, mg->symmetric_gs ? 0 : 1, s);
}
// MG.hh:486-508
static void gmg_fmg(vfem_gmg *mg, int l, int nsmooth, bool residual_system, hipStream_t s) {
    GLevel &L = mg->lv[l];
    if (l == mg->L) { gmg_coarsest(mg, L.b.p, L.x.p, s); return; }
    GLevel &C = mg->lv[l + 1];
    gmg_restrict(mg, l, L.b.p, C.b.p, s);
    gmg_fmg(mg, l + 1, nsmooth, residual_system, s);
    gmg_prolong(mg, l, C.x.p, L.x.p, 0, s);
    gmg_vcycle(mg, l, nsmooth, residual_system, s);
}
static void gmg_cycles(vfem_gmg *mg, int num_steps, int nsmooth, bool zero_dirichlet, bool fmg, hipStream_t s) {
    if (fmg) {
        gmg_fmg(mg, 0, nsmooth, zero_dirichlet, s);
        for (int i = 1; i < num_steps; ++i) gmg_vcycle(mg, 0, nsmooth, zero_dirichlet, s);
    } else
        for (int i = 0; i < num_steps; ++i) gmg_vcycle(mg, 0, nsmooth, zero_dirichlet, s);
}

static void coarsen_mask(const GDims &f, const std::vector<uint8_t> &fm, const GDims &c, std::vector<uint8_t> &cm) {
    // MG.hh:57-84 in integer arithmetic
    const int N = f.N, p = f.p;
    cm.assign((size_t) c.nnodes, 0);
    for (long long nf = 0; nf < f.nnodes; ++nf) {
        const uint8_t m = fm[nf];
        if (!m) continue;
        int g[3] = {0, 0, 0};
        { long long q = nf; for (int a = N - 1; a >= 0; --a) { g[a] = (int) (q % f.nn[a]); q /= f.nn[a]; } }
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        bool any = false;
        for (int a = 0; a < N; ++a) {
            const int e = std::min(g[a] / (2 * p), c.ne[a] - 1), t = g[a] - 2 * p * e;
            if (t == 0) { lo[a] = hi[a] = p * e; any = true; }
            else if (t == 2 * p) { lo[a] = hi[a] = p * e + p; any = true; }
            else { lo[a] = p * e; hi[a] = p * e + p; }
        }
        if (!any) throw Error("Dirichlet constraints on internal nodes are not supported");
        for (int a0 = lo[0]; a0 <= hi[0]; ++a0)
            for (int a1 = lo[1]; a1 <= hi[1]; ++a1)
                for (int a2 = lo[2]; a2 <= hi[2]; ++a2) {
                    const int gg[3] = {a0, a1, a2};
                    long long node = gg[0];
                    for (int a = 1; a < N; ++a) node = node * c.nn[a] + gg[a];
                    cm[node] |= m;
                }
    }
}

extern "C" {

int vfem_gsim_create_padded(vfem_gsim **out, int dim, int degree, const double *bbmin, const double *bbmax, const int64_t *ne,
                            int64_t ex_lo, int64_t ex_hi) {
    G_TRY
    if (dim != 2 && dim != 3) throw Error("dimension must be 2 or 3");
    if (degree != 1 && degree != 2) throw Error("No template instantiation matching degreesPerDimension!");
    long long n[3] = {1, 1, 1};
    for (int a = 0; a < dim; ++a) {
        if (ne[a] < 1 || ne[a] > 4096) throw Error("elements per dimension must be in [1, 4096]");
        n[a] = ne[a];
    }
    std::unique_ptr<vfem_gsim> sim(new vfem_gsim);
    sim->d = make_gdims(dim, degree, n);
    for (int a = 0; a < dim; ++a) {
        sim->h[a] = (bbmax[a] - bbmin[a]) / (double) ne[a];
        if (!(sim->h[a] > 0)) throw Error("empty domain bounding box");
    }
    sim->update_k0();
    if (ex_lo < 0 || ex_hi < 0 || ((ex_lo || ex_hi) && dim != 3)) throw Error("element padding needs a 3-D grid and non-negative layer counts");
    sim->ex_lo = ex_lo; sim->ex_hi = ex_hi;
    sim->rho.alloc((size_t) sim->stored_elems()); sim->rho.zero(nullptr);
    sim->E.alloc((size_t) sim->stored_elems());
    sim->update_E(nullptr);
    sim->hmask.assign((size_t) sim->d.nnodes, 0);
    sim->dmask.alloc((size_t) sim->d.nnodes); sim->dmask.zero(nullptr);
    sim->dvals.alloc((size_t) sim->d.nnodes * dim); sim->dvals.zero(nullptr);
    VFEM_HIP(hipDeviceSynchronize());
    *out = sim.release();
    G_CATCH
}
int vfem_gsim_create(vfem_gsim **out, int dim, int degree, const double *bbmin, const double *bbmax, const int64_t *ne) {
    return vfem_gsim_create_padded(out, dim, degree, bbmin, bbmax, ne, 0, 0);
}
int vfem_gsim_destroy(vfem_gsim *sim) { G_TRY delete sim; G_CATCH }
int64_t vfem_gsim_num_stored_elements(const vfem_gsim *sim) { return sim->stored_elems(); }
int64_t vfem_gsim_num_nodes(const vfem_gsim *sim) { return sim->d.nnodes; }
int64_t vfem_gsim_num_elements(const vfem_gsim *sim) { return sim->d.nelems; }
int vfem_gsim_ke_size(const vfem_gsim *sim) { return sim->d.ke; }
int vfem_gsim_set_isotropic(vfem_gsim *sim, double young, double poisson) {
    G_TRY
    // ElasticityTensor::setIsotropic (ElasticityTensor.hh:100-133): 3-D Lame parameters; 2-D = plane stress
    sim->lambda = sim->d.N == 2 ? poisson * young / (1.0 - poisson * poisson)
                                : poisson * young / ((1.0 + poisson) * (1.0 - 2.0 * poisson));
    sim->mu = young / (2.0 + 2.0 * poisson);
    sim->update_k0();
    G_CATCH
}
int vfem_gsim_set_simp(vfem_gsim *sim, double E0, double Emin, double gamma) {
    G_TRY
    sim->E0 = E0; sim->Emin = Emin; sim->gamma = gamma;
    sim->update_E(nullptr);
    VFEM_HIP(hipDeviceSynchronize());
    G_CATCH
}
int vfem_gsim_k0(const vfem_gsim *sim, double *K0_host) {
    G_TRY std::memcpy(K0_host, sim->K0.data(), sim->K0.size() * sizeof(double)); G_CATCH
}
int vfem_gsim_set_dirichlet(vfem_gsim *sim, const uint8_t *mask_host, const double *values_host) {
    G_TRY
    sim->hmask.assign(mask_host, mask_host + sim->d.nnodes);
    VFEM_HIP(hipMemcpy(sim->dmask.p, mask_host, (size_t) sim->d.nnodes, hipMemcpyHostToDevice));
    VFEM_HIP(hipMemcpy(sim->dvals.p, values_host, (size_t) sim->d.nnodes * sim->d.N * sizeof(double), hipMemcpyHostToDevice));
    G_CATCH
}
int vfem_gsim_set_densities(vfem_gsim *sim, const double *rho, void *stream) {
    G_TRY
    // padded simulators take all stored layers (ex_lo + ne[0] + ex_hi), x slowest
    VFEM_HIP(hipMemcpyAsync(sim->rho.p, rho, (size_t) sim->stored_elems() * sizeof(double), hipMemcpyDeviceToDevice, GS(stream)));
    sim->update_E(GS(stream));
    G_CATCH
}
int vfem_gsim_get_densities(const vfem_gsim *sim, double *rho, void *stream) {
    G_TRY
    VFEM_HIP(hipMemcpyAsync(rho, sim->rho_local(), (size_t) sim->d.nelems * sizeof(double), hipMemcpyDeviceToDevice, GS(stream)));
    G_CATCH
}
int vfem_gsim_set_option(vfem_gsim *sim, int key, int value) {
    G_TRY
    if (key == 6 && value >= 0 && value <= 2) sim->q2_impl = value;
    else if (key == 14 && value >= 0 && value <= 2) sim->q2_l1_virtual = value;
    else if (key == 16 && value >= 0 && value <= 2) sim->q2_gs_impl = value;
    else if (key == 17 && (value == 0 || value == 1)) sim->transfer_axis = value;
    else throw Error("unknown option or value out of range");
    G_CATCH
}
int vfem_gsim_apply_k(const vfem_gsim *sim, const double *u, double *out, void *stream) {
    G_TRY
    if (sim->d.N == 3 && sim->d.p == 2 && sim->q2_fast && sim->q2_impl == 0)
        launch_apply_q2_march(sim->d.ne[0], sim->d.ne[1], sim->d.ne[2], sim->q2tab.p, sim->E_local(), u, out, GS(stream));
    else if (sim->d.N == 3 && sim->d.p == 2 && sim->q2_fast && sim->q2_impl == 2)
        launch_apply_q2_pencil(sim->d.ne[0], sim->d.ne[1], sim->d.ne[2], sim->q2tab.p, sim->E_local(), u, out, GS(stream));
    else if (sim->d.N == 3 && sim->d.p == 2)
        launch_apply_q2(sim->d.ne[0], sim->d.ne[1], sim->d.ne[2], sim->dK0.p, sim->E_local(), u, out, GS(stream));
    else
        g_apply(sim->d, sim->dK0.p, 0, sim->E_local(), u, nullptr, nullptr, 0, out, GS(stream));
    G_CATCH
}
int vfem_gsim_compliance_gradient(const vfem_gsim *sim, const double *u, double *g, void *stream) {
    G_TRY
    kg_gradient<<<dim3((unsigned) ((sim->d.nelems + 3) / 4)), dim3(256), 0, GS(stream)>>>(sim->d, sim->dK0.p, sim->rho_local(), sim->E0, sim->Emin,
                                                                                       sim->gamma, u, g);
    VFEM_HIP(hipGetLastError());
    G_CATCH
}

int vfem_gsim_compliance(const vfem_gsim *sim, const double *f, const double *u, double *value_host, void *stream) {
    G_TRY
    // stream-ordered scratch (hipMallocAsync does not synchronise the device): evaluations on different streams share nothing
    double *tmp = nullptr;
    VFEM_HIP(hipMallocAsync((void **) &tmp, (4096 + 1) * sizeof(double), GS(stream)));
    launch_dot((long long) sim->d.N * sim->d.nnodes, f, u, tmp, tmp + 4096, GS(stream));
    double v = 0.0;
    VFEM_HIP(hipMemcpyAsync(&v, tmp + 4096, sizeof(double), hipMemcpyDeviceToHost, GS(stream)));
    VFEM_HIP(hipFreeAsync(tmp, GS(stream)));
    VFEM_HIP(hipStreamSynchronize(GS(stream)));
    *value_host = 0.5 * v;                                            // TopologyOptimizationObjective.hh:39-41
    G_CATCH
}

// interpolation weights, compressed interpolation operators phi[fi](fine_n, coarse_n) (MG.hh:557-583) and
// cK0[fi] = I_fi^T K0 I_fi (MG.hh:644-648)
static void gmg_setup_transfer_tables(vfem_gmg *mg) {
    const vfem_gsim *fine = mg->fine;
    const int N = fine->d.N, p = fine->d.p;
    for (int t = 0; t < 5; ++t) for (int a = 0; a < 3; ++a) mg->W.w[t][a] = 0.0;
    for (int t = 0; t <= 2 * p; ++t) for (int a = 0; a <= p; ++a) mg->W.w[t][a] = lagrange1d(p, a, (double) t / (2.0 * p));
    const int npe = fine->d.npe, ke = fine->d.ke, q1 = p + 1, nch = 1 << N;
    std::vector<double> phi((size_t) nch * npe * npe), cK0((size_t) nch * ke * ke, 0.0), T((size_t) ke * ke);
    for (int fi = 0; fi < nch; ++fi) {
        double *ph = phi.data() + (size_t) fi * npe * npe;
        for (int fn = 0; fn < npe; ++fn)
            for (int cn = 0; cn < npe; ++cn) {
                int mf = fn, mc = cn;
                double w = 1.0;
                for (int a = N - 1; a >= 0; --a) {
                    const int lf = mf % q1, lc = mc % q1; mf /= q1; mc /= q1;
                    w *= mg->W.w[lf + p * ((fi >> a) & 1)][lc];
                }
                ph[fn * npe + cn] = w;
            }
        const std::vector<double> &K0 = fine->K0;
        for (int i = 0; i < ke; ++i)
            for (int j = 0; j < ke; ++j) {
                const int m = j / N, b = j % N;
                double v = 0.0;
                for (int qn = 0; qn < npe; ++qn) v += K0[(size_t) i * ke + N * qn + b] * ph[qn * npe + m];
                T[(size_t) i * ke + j] = v;
            }
        double *cK = cK0.data() + (size_t) fi * ke * ke;
        for (int i = 0; i < ke; ++i)
            for (int j = 0; j < ke; ++j) {
                const int n = i / N, a = i % N;
                double v = 0.0;
                for (int pn = 0; pn < npe; ++pn) v += ph[pn * npe + n] * T[(size_t) (N * pn + a) * ke + j];
                cK[(size_t) i * ke + j] = v;
            }
    }
    if (N == 3 && p == 2) {
        std::vector<double> tab((size_t) 27 * 27 * 72);
        for (int ln = 0; ln < 27; ++ln)
            for (int m = 0; m < 27; ++m)
                for (int f = 0; f < 8; ++f)
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 3; ++c)
                            tab[((size_t) (ln * 27 + m) * 8 + f) * 9 + 3 * r + c] = cK0[(size_t) f * 6561 + (size_t) (3 * ln + r) * 81 + 3 * m + c];
        mg->l1tab.alloc(tab.size());
        VFEM_HIP(hipMemcpy(mg->l1tab.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (N == 3 && p == 2) {
        // c2K0[g][f] = I_g^T cK0[f] I_g, stored [entry][8 g + f] for kg_coarsen_level2_q2
        std::vector<double> t2((size_t) 6561 * 64), T2((size_t) ke * ke);
        for (int g = 0; g < 8; ++g) {
            const double *ph = phi.data() + (size_t) g * npe * npe;
            for (int f = 0; f < 8; ++f) {
                const double *A = cK0.data() + (size_t) f * ke * ke;
                for (int i = 0; i < ke; ++i)
                    for (int j = 0; j < ke; ++j) {
                        const int m = j / N, b = j % N;
                        double v = 0.0;
                        for (int qn = 0; qn < npe; ++qn) v += A[(size_t) i * ke + N * qn + b] * ph[qn * npe + m];
                        T2[(size_t) i * ke + j] = v;
                    }
                for (int i = 0; i < ke; ++i)
                    for (int j = 0; j < ke; ++j) {
                        const int n = i / N, a = i % N;
                        double v = 0.0;
                        for (int pn = 0; pn < npe; ++pn) v += ph[pn * npe + n] * T2[(size_t) (N * pn + a) * ke + j];
                        t2[((size_t) i * ke + j) * 64 + 8 * g + f] = v;
                    }
            }
        }
        mg->c2tab.alloc(t2.size());
        VFEM_HIP(hipMemcpy(mg->c2tab.p, t2.data(), t2.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    mg->phi.alloc(phi.size()); mg->cK0.alloc(cK0.size());
    VFEM_HIP(hipMemcpy(mg->phi.p, phi.data(), phi.size() * sizeof(double), hipMemcpyHostToDevice));
    VFEM_HIP(hipMemcpy(mg->cK0.p, cK0.data(), cK0.size() * sizeof(double), hipMemcpyHostToDevice));
}
static void gmg_alloc_level_fields(GLevel &lv) {
    lv.mask.alloc((size_t) lv.d.nnodes);
    VFEM_HIP(hipMemcpy(lv.mask.p, lv.hmask.data(), lv.hmask.size(), hipMemcpyHostToDevice));
    const size_t n = (size_t) lv.d.nnodes * lv.d.N;
    lv.x.alloc(n); lv.b.alloc(n); lv.r.alloc(n);
    lv.x.zero(nullptr); lv.b.zero(nullptr); lv.r.zero(nullptr);
}

// first_active > 0: the replicated coarse part of a slab-decomposed hierarchy -- levels below hold no fields and no operators;
// the element matrices of level first_active are imported (vfem_gmg_import_level_ke), cycles start there
// (vfem_gmg_cycle_from_level); `fine` supplies the grid, the material and the Dirichlet mask only.
static int gmg_create_common(vfem_gmg **out, vfem_gsim *fine, int L, int first_active) {
    G_TRY
    if (L < 0 || L > 16) throw Error("invalid number of coarsening levels");
    if (first_active < 0 || first_active > L) throw Error("first active level out of range");
    if (fine->ex_lo || fine->ex_hi) throw Error("simulators with element padding need vfem_gmg_create_slab");
    std::unique_ptr<vfem_gmg> mg(new vfem_gmg);
    mg->fine = fine; mg->L = L; mg->first_active = first_active;
    mg->lv.resize(L + 1);
    const int N = fine->d.N, p = fine->d.p;
    long long ne[3] = {fine->d.ne[0], fine->d.ne[1], fine->d.ne[2]};
    for (int l = 0; l <= L; ++l) {
        GLevel &lv = mg->lv[l];
        if (l > 0) {
            for (int a = 0; a < N; ++a) {
                if (ne[a] % 2) throw Error("Grid size currently must be divisible by 2^numCoarseningLevels (nonuniform coarsening not yet implemented)");
                ne[a] /= 2;
            }
        }
        lv.d = make_gdims(N, p, ne);
        if (l == 0) lv.hmask = fine->hmask; else coarsen_mask(mg->lv[l - 1].d, mg->lv[l - 1].hmask, lv.d, lv.hmask);
        if (l >= first_active) gmg_alloc_level_fields(lv);
    }
    gmg_setup_transfer_tables(mg.get());
    if (first_active == 0) {
        const size_t n0 = (size_t) fine->d.nnodes * N;
        mg->pr.alloc(n0); mg->pd.alloc(n0); mg->pAd.alloc(n0); mg->ps.alloc(n0);
    }
    mg->scal.alloc(8); mg->scal.zero(nullptr); mg->scratch.alloc(4096);
    VFEM_HIP(hipDeviceSynchronize());
    *out = mg.release();
    G_CATCH
}
int vfem_gmg_create(vfem_gmg **out, vfem_gsim *fine, int L) { return gmg_create_common(out, fine, L, 0); }
int vfem_gmg_create_partial(vfem_gmg **out, vfem_gsim *fine, int L, int first_active_level) {
    return gmg_create_common(out, fine, L, first_active_level);
}
// The local hierarchy of one rank of an x-slab decomposition (3-D).  Level l holds levels_host[l].nx element layers (owned +
// ghost) of a grid whose other two extents halve from level to level; levels_host[l].elem_extra_lo / _hi further element layers
// are kept in the element arrays only (level 0: the simulator's padding), so that the stored layers halve exactly and the
// ghost elements of every level get complete Galerkin matrices; levels_host[l].xshift (l >= 1) places local plane 0 of level l
// at local plane xshift of level l-1 (in level l-1's planes: fine plane = 2 * coarse plane + xshift).  xparity must be 0: the
// driver aligns the slabs so that every local grid starts at an even global element (the colours need no offset then).
// masks_host[l]: the level's Dirichlet masks (slices of the global coarsened masks).  The last level only serves the transfers.
int vfem_gmg_create_slab(vfem_gmg **out, vfem_gsim *fine, int n_levels, const vfem_slab_level *lv_in, const uint8_t *const *masks_host) {
    G_TRY
    if (n_levels < 1) throw Error("need at least one level");
    if (fine->d.N != 3) throw Error("slab hierarchies are three-dimensional");
    std::unique_ptr<vfem_gmg> mg(new vfem_gmg);
    mg->fine = fine; mg->L = n_levels - 1; mg->slab = true;
    mg->lv.resize((size_t) n_levels);
    long long ny = fine->d.ne[1], nz = fine->d.ne[2];
    for (int l = 0; l < n_levels; ++l) {
        GLevel &lv = mg->lv[l];
        if (l > 0) {
            if (ny % 2 || nz % 2) throw Error("Grid size currently must be divisible by 2^numCoarseningLevels (nonuniform coarsening not yet implemented)");
            ny /= 2; nz /= 2;
        }
        if (lv_in[l].xparity != 0) throw Error("slab levels of the generic path must start at an even global element");
        const long long ne[3] = {lv_in[l].nx, ny, nz};
        lv.d = make_gdims(3, fine->d.p, ne);
        lv.pad_lo = lv_in[l].elem_extra_lo; lv.pad_hi = lv_in[l].elem_extra_hi;
        lv.xs = l > 0 ? (int) lv_in[l].xshift : 0;
        if (lv.xs > 0) throw Error("xshift must be <= 0");
        lv.hmask.assign(masks_host[l], masks_host[l] + lv.d.nnodes);
        gmg_alloc_level_fields(lv);
    }
    if (fine->d.ne[0] != mg->lv[0].d.ne[0] || fine->ex_lo != mg->lv[0].pad_lo || fine->ex_hi != mg->lv[0].pad_hi)
        throw Error("level 0 of the slab hierarchy does not match the simulator");
    gmg_setup_transfer_tables(mg.get());
    mg->scal.alloc(8); mg->scal.zero(nullptr); mg->scratch.alloc(4096);
    VFEM_HIP(hipDeviceSynchronize());
    *out = mg.release();
    G_CATCH
}
int vfem_gmg_destroy(vfem_gmg *mg) {
    G_TRY
    delete mg;
    G_CATCH
}
int vfem_gmg_num_levels(const vfem_gmg *mg) { return mg->L + 1; }
int vfem_gmg_level_dims(const vfem_gmg *mg, int level, int64_t ne[3]) {
    G_TRY
    if (level < 0 || level > mg->L) throw Error("level out of range");
    for (int a = 0; a < 3; ++a) ne[a] = mg->lv[level].d.ne[a];
    G_CATCH
}
int64_t vfem_gmg_level_num_nodes(const vfem_gmg *mg, int level) { return (level < 0 || level > mg->L) ? -1 : mg->lv[level].d.nnodes; }
int vfem_gmg_level_dirichlet_mask(const vfem_gmg *mg, int level, uint8_t *mask_host) {
    G_TRY
    if (level < 0 || level > mg->L) throw Error("level out of range");
    std::memcpy(mask_host, mg->lv[level].hmask.data(), mg->lv[level].hmask.size());
    G_CATCH
}
int vfem_gmg_set_symmetric_gauss_seidel(vfem_gmg *mg, int symmetric) { mg->symmetric_gs = symmetric != 0; return 0; }
int vfem_gmg_update_operators(vfem_gmg *mg, void *stream) { G_TRY gmg_update(mg, GS(stream)); G_CATCH }
static void g_check_level(const vfem_gmg *mg, int level, bool need_ops) {
    if (level < 0 || level > mg->L) throw Error("level out of range");
    if (need_ops && (level > 0 || mg->external_ke_level == 0) && !mg->operators_valid) throw Error("coarse operators not built: call updateElementStiffnessMatrices first");
}
int vfem_gmg_apply_k(vfem_gmg *mg, int level, const double *u, double *out, void *stream) {
    G_TRY g_check_level(mg, level, true); gmg_apply(mg, level, u, nullptr, 0, out, GS(stream)); G_CATCH
}
int vfem_gmg_residual(vfem_gmg *mg, int level, const double *u, const double *b, double *r, void *stream) {
    G_TRY g_check_level(mg, level, true); gmg_apply(mg, level, u, b, 1, r, GS(stream)); G_CATCH
}
int vfem_gmg_smooth(vfem_gmg *mg, int level, double *u, const double *b, int forward, void *stream) {
    G_TRY g_check_level(mg, level, true); gmg_smooth(mg, level, u, b, forward, GS(stream)); G_CATCH
}
int vfem_gmg_smooth_colors(vfem_gmg *mg, int level, double *u, const double *b, int forward, int first, int count, void *stream) {
    G_TRY
    g_check_level(mg, level, true);
    int ncol = 1;
    for (int a = 0; a < mg->fine->d.N; ++a) ncol *= mg->fine->d.p + 1;
    if (first < 0 || count < 0 || first + count > ncol) throw Error("colour range out of bounds");
    gmg_smooth(mg, level, u, b, forward, GS(stream), first, count);
    G_CATCH
}
// one cycle of the replicated coarse hierarchy on the residual system of `level` (x = 0 initial guess for the full cycle)
int vfem_gmg_cycle_from_level(vfem_gmg *mg, int level, double *x, const double *b, int nsmooth, int fmg, void *stream) {
    G_TRY
    g_check_level(mg, level, false);
    if (level < mg->first_active) throw Error("level below the first active level of this hierarchy");
    if (mg->slab) throw Error("slab hierarchies are cycled by the distributed driver");
    if (!mg->operators_valid) throw Error("coarse operators not built: call vfem_gmg_update_operators first");
    hipStream_t s = GS(stream);
    GLevel &L = mg->lv[level];
    const size_t bytes = (size_t) L.d.nnodes * L.d.N * sizeof(double);
    VFEM_HIP(hipMemcpyAsync(L.b.p, b, bytes, hipMemcpyDeviceToDevice, s));
    if (fmg) gmg_fmg(mg, level, nsmooth, true, s);
    else {
        VFEM_HIP(hipMemcpyAsync(L.x.p, x, bytes, hipMemcpyDeviceToDevice, s));
        gmg_vcycle(mg, level, nsmooth, true, s);
    }
    VFEM_HIP(hipMemcpyAsync(x, L.x.p, bytes, hipMemcpyDeviceToDevice, s));
    G_CATCH
}
// Galerkin element matrices of `count_x` element layers of `level` computed from their children, which start at stored layer
// `child_first_layer` of level - 1 (level 1: of the simulator's moduli): what a rank contributes to the first replicated level
int vfem_gmg_export_level_ke(vfem_gmg *mg, int level, int64_t child_first_layer, int64_t count_x, double *ke_out, void *stream) {
    G_TRY
    g_check_level(mg, level, false);
    if (level < 1) throw Error("level 0 has no stored element matrices");
    if (level > 1 && !mg->operators_valid) throw Error("coarse operators not built: call vfem_gmg_update_operators first");
    const GDims fs = stored_dims(mg, level - 1);
    if (child_first_layer < 0 || count_x < 0 || child_first_layer + 2 * count_x > fs.ne[0]) throw Error("child layers outside the stored element array");
    const GLevel &lv = mg->lv[level];
    const long long cne[3] = {count_x, lv.d.ne[1], lv.d.ne[2]}, fne[3] = {2 * count_x, fs.ne[1], fs.ne[2]};
    const GDims c = make_gdims(3, lv.d.p, cne), f = make_gdims(3, lv.d.p, fne);
    if (level == 2 && mg->l1_virtual) {
        if (child_first_layer % 2) throw Error("child layers must start at an even stored layer");
        coarsen_through_virtual_level1(mg, child_first_layer / 2, count_x, ke_out, GS(stream));
        return 0;
    }
    const size_t child_stride = level == 1 ? 1 : (size_t) lv.d.ke * lv.d.ke;
    const double *src = (level == 1 ? mg->fine->E.p : mg->lv[level - 1].Ke.p) + (size_t) child_first_layer * fs.ne[1] * fs.ne[2] * child_stride;
    g_coarsen(mg, f, c, level == 1, src, ke_out, GS(stream));
    G_CATCH
}
int vfem_gmg_import_level_ke(vfem_gmg *mg, int level, const double *ke, void *stream) {
    G_TRY
    g_check_level(mg, level, false);
    if (mg->slab) throw Error("element matrices are imported into replicated hierarchies only");
    if (level != mg->first_active) throw Error("element matrices are imported into the first active level (level 0 of a hierarchy created on the coarse grid itself)");
    GLevel &lv = mg->lv[level];
    const size_t n = (size_t) lv.d.nelems * lv.d.ke * lv.d.ke;
    lv.Ke.alloc(n);
    VFEM_HIP(hipMemcpyAsync(lv.Ke.p, ke, n * sizeof(double), hipMemcpyDeviceToDevice, GS(stream)));
    mg->external_ke_level = level;
    mg->operators_valid = false;
    G_CATCH
}
int vfem_gmg_zero_dirichlet(vfem_gmg *mg, int level, double *u, void *stream) {
    G_TRY g_check_level(mg, level, false); g_dirichlet(mg->lv[level].d, mg->lv[level].mask.p, nullptr, u, GS(stream)); G_CATCH
}
int vfem_gmg_restrict(vfem_gmg *mg, int fine_level, const double *fine, double *coarse, void *stream) {
    G_TRY g_check_level(mg, fine_level + 1, false); gmg_restrict(mg, fine_level, fine, coarse, GS(stream)); G_CATCH
}
int vfem_gmg_interpolate(vfem_gmg *mg, int fine_level, const double *coarse, double *fine, int accumulate, void *stream) {
    G_TRY g_check_level(mg, fine_level + 1, false); gmg_prolong(mg, fine_level, coarse, fine, accumulate, GS(stream)); G_CATCH
}
int vfem_gmg_solve(vfem_gmg *mg, double *x, const double *f, int num_steps, int nsmooth, int stiffness_updated,
                   int zero_dirichlet, int fmg, void *stream) {
    G_TRY
    hipStream_t s = GS(stream);
    if (mg->slab || mg->first_active > 0) throw Error("slab / partial hierarchies are cycled by the distributed driver");
    if (!stiffness_updated) gmg_update(mg, s);                         // MG.hh:455
    else if (!mg->operators_valid) throw Error("coarse operators not built");
    if (num_steps == 0) return 0;
    const size_t bytes = (size_t) mg->fine->d.nnodes * mg->fine->d.N * sizeof(double);
    VFEM_HIP(hipMemcpyAsync(mg->lv[0].x.p, x, bytes, hipMemcpyDeviceToDevice, s));
    VFEM_HIP(hipMemcpyAsync(mg->lv[0].b.p, f, bytes, hipMemcpyDeviceToDevice, s));
    gmg_cycles(mg, num_steps, nsmooth, zero_dirichlet != 0, fmg != 0, s);
    VFEM_HIP(hipMemcpyAsync(x, mg->lv[0].x.p, bytes, hipMemcpyDeviceToDevice, s));
    G_CATCH
}
int vfem_gmg_pcg(vfem_gmg *mg, double *x, const double *b, int max_iter, double tol, int mg_iterations, int mg_smoothing,
                 int fmg, vfem_residual_cb residual_cb, void *cb_user, int *iters_out, double *relres_out, void *stream) {
    G_TRY
    hipStream_t s = GS(stream);
    if (mg->slab || mg->first_active > 0) throw Error("slab / partial hierarchies are solved by the distributed driver");
    vfem_gsim *sim = mg->fine;
    const long long nn = sim->d.nnodes, n3 = sim->d.N * nn;
    const size_t bytes = (size_t) n3 * sizeof(double);
    double *r = mg->pr.p, *d = mg->pd.p, *Ad = mg->pAd.p, *sv = mg->ps.p, *sc = mg->scal.p;
    const uint8_t *mask = mg->lv[0].mask.p;
    g_dirichlet(sim->d, mask, sim->dvals.p, x, s);                       // MG.hh:687-688
    gmg_update(mg, s);                                                   // MG.hh:690-691
    double host_sc[4];
    launch_dot(n3, b, b, mg->scratch.p, sc + 4, s);
    gmg_apply(mg, 0, x, b, 1, r, s);                                     // MG.hh:696
    launch_dot(n3, r, r, mg->scratch.p, sc + 3, s);
    VFEM_HIP(hipMemcpyAsync(host_sc, sc + 3, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    VFEM_HIP(hipStreamSynchronize(s));
    double rr = host_sc[0];
    const double bb = host_sc[1];
    int it = 0;
    while (it < max_iter && rr > tol * tol * bb) {                       // MG.hh:711 (counter started at 0)
        ++it;
        if (mg_smoothing == 0) {
            VFEM_HIP(hipMemcpyAsync(sv, r, bytes, hipMemcpyDeviceToDevice, s));
        } else {
            mg->lv[0].x.zero(s);
            VFEM_HIP(hipMemcpyAsync(mg->lv[0].b.p, r, bytes, hipMemcpyDeviceToDevice, s));
            gmg_cycles(mg, mg_iterations, mg_smoothing, true, fmg != 0, s);
            VFEM_HIP(hipMemcpyAsync(sv, mg->lv[0].x.p, bytes, hipMemcpyDeviceToDevice, s));
        }
        g_dirichlet(sim->d, mask, nullptr, sv, s);
        launch_shift_scalar(sc, s);
        launch_dot(n3, r, sv, mg->scratch.p, sc + 0, s);
        launch_pcg_direction(n3, sv, d, sc, it == 1, s);
        gmg_apply(mg, 0, d, nullptr, 2, Ad, s);                          // zeroDirichlet(K d)
        launch_dot(n3, d, Ad, mg->scratch.p, sc + 2, s);
        launch_pcg_step(n3, x, r, d, Ad, sc, s);
        launch_dot(n3, r, r, mg->scratch.p, sc + 3, s);
        VFEM_HIP(hipMemcpyAsync(host_sc, sc + 3, sizeof(double), hipMemcpyDeviceToHost, s));
        VFEM_HIP(hipStreamSynchronize(s));
        rr = host_sc[0];
        if (!(rr == rr)) throw Error("PCG produced NaN residual");
        if (residual_cb) residual_cb(cb_user, it, std::sqrt(rr));
    }
    if (iters_out) *iters_out = it;
    if (relres_out) *relres_out = bb > 0 ? std::sqrt(rr / bb) : 0.0;
    G_CATCH
}

}  // extern "C"
