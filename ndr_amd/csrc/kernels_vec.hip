// Elementwise / reduction kernels: SIMP moduli, Dirichlet masking, CG vector updates with
// wavefront (64-lane) reductions, compliance sensitivity.
#include "vfem_internal.h"

#include <cstdlib>
#include <string>

namespace vfem {

static inline unsigned grid_for(long long n, int block, int cap = 4096) {
    long long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned) g;
}

// E_e = Emin + rho^gamma (E0 - Emin)  (TPS.hh:725-727); computed once per density update instead of
// per element per apply.
__global__ void __launch_bounds__(256) k_simp(long long n, const double *__restrict__ rho, double E0, double Emin,
                                              double gamma, double *__restrict__ E) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const double r = rho[i];
        const double p = (gamma == 3.0) ? r * r * r : pow(r, gamma);
        E[i] = Emin + p * (E0 - Emin);
    }
}
void launch_simp(long long n, const double *rho, double E0, double Emin, double gamma, double *E, hipStream_t s) {
    k_simp<<<grid_for(n, 256), 256, 0, s>>>(n, rho, E0, Emin, gamma, E);
    VFEM_HIP(hipGetLastError());
}

__global__ void __launch_bounds__(256) k_fill(long long n, double v, double *__restrict__ x) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) x[i] = v;
}
void launch_fill(long long n, double v, double *x, hipStream_t s) {
    k_fill<<<grid_for(n, 256), 256, 0, s>>>(n, v, x);
    VFEM_HIP(hipGetLastError());
}

// zeroOutDirichletComponents (MG.hh:364-378)
__global__ void __launch_bounds__(256) k_zero_dirichlet(long long nn, const uint8_t *__restrict__ mask, double *__restrict__ u) {
    for (long long n = (long long) blockIdx.x * blockDim.x + threadIdx.x; n < nn; n += (long long) gridDim.x * blockDim.x) {
        const uint8_t m = mask[n];
        if (m) {
            if (m & 1) u[3 * n] = 0.0;
            if (m & 2) u[3 * n + 1] = 0.0;
            if (m & 4) u[3 * n + 2] = 0.0;
        }
    }
}
void launch_zero_dirichlet(long long nn, const uint8_t *mask, double *u, hipStream_t s) {
    k_zero_dirichlet<<<grid_for(nn, 256), 256, 0, s>>>(nn, mask, u);
    VFEM_HIP(hipGetLastError());
}

// enforceDirichletConditions (MG.hh:386-398)
__global__ void __launch_bounds__(256) k_enforce_dirichlet(long long nn, const uint8_t *__restrict__ mask,
                                                           const double *__restrict__ vals, double *__restrict__ u, int zero) {
    for (long long n = (long long) blockIdx.x * blockDim.x + threadIdx.x; n < nn; n += (long long) gridDim.x * blockDim.x) {
        const uint8_t m = mask[n];
        if (m) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if ((m >> c) & 1) u[3 * n + c] = (zero || !vals) ? 0.0 : vals[3 * n + c];
        }
    }
}
void launch_enforce_dirichlet(long long nn, const uint8_t *mask, const double *vals, double *u, int zero, hipStream_t s) {
    k_enforce_dirichlet<<<grid_for(nn, 256), 256, 0, s>>>(nn, mask, vals, u, zero);
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// dot product: 1024 blocks of per-wave shuffle reductions -> partials -> one final block.
// Deterministic (fixed partition and order), unlike an atomic finish.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

__device__ __forceinline__ double block_sum_256(double v, double *sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) t = sh[0] + sh[1] + sh[2] + sh[3];
    return t;   // valid in thread 0
}

constexpr int DOT_BLOCKS = 1024;

// The element loops of the PCG vector kernels take two consecutive doubles per thread and step -- one 16-byte access where the
// arrays are 16-byte aligned (VEC; otherwise two 8-byte accesses to the same elements, so the partition and the order of a sum do
// not depend on the alignment); an odd last element goes to the first thread of the grid after its pairs.
template <bool VEC>
__device__ __forceinline__ void load2(const double *p, long long i, double &x0, double &x1) {
    if (VEC) { const double2 v = *reinterpret_cast<const double2 *>(p + i); x0 = v.x; x1 = v.y; }
    else { x0 = p[i]; x1 = p[i + 1]; }
}
template <bool VEC>
__device__ __forceinline__ void store2(double *p, long long i, double x0, double x1) {
    if (VEC) *reinterpret_cast<double2 *>(p + i) = make_double2(x0, x1);
    else { p[i] = x0; p[i + 1] = x1; }
}
static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <bool VEC>
__global__ void __launch_bounds__(256) k_dot_partial(long long n, const double *__restrict__ a, const double *__restrict__ b,
                                                     double *__restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    const long long t0 = (long long) blockIdx.x * blockDim.x + threadIdx.x, nt = (long long) gridDim.x * blockDim.x;
    for (long long i = 2 * t0; i + 1 < n; i += 2 * nt) {
        double a0, a1, b0, b1;
        load2<VEC>(a, i, a0, a1); load2<VEC>(b, i, b0, b1);
        acc = fma(a0, b0, acc);
        acc = fma(a1, b1, acc);
    }
    if ((n & 1) && t0 == 0) acc = fma(a[n - 1], b[n - 1], acc);
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void __launch_bounds__(256) k_dot_final(int nparts, const double *__restrict__ partial, double *__restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) *out = t;
}

// zeroDirichlet(bm) and a . bm in one pass over both (bm is written only where a component is constrained); same partition and
// order of the sum as k_dot_partial, so the value equals launch_zero_dirichlet + launch_dot bit for bit
template <bool VEC>
__global__ void __launch_bounds__(256) k_dot_masked_partial(long long n, const double *__restrict__ a, double *__restrict__ bm,
                                                            const uint8_t *__restrict__ mask, double *__restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    const long long t0 = (long long) blockIdx.x * blockDim.x + threadIdx.x, nt = (long long) gridDim.x * blockDim.x;
    auto masked = [&](long long i, double v) {
        const long long node = i / 3;
        if ((mask[node] >> (int) (i - 3 * node)) & 1) { v = 0.0; bm[i] = 0.0; }
        return v;
    };
    for (long long i = 2 * t0; i + 1 < n; i += 2 * nt) {
        double a0, a1, v0, v1;
        load2<VEC>(a, i, a0, a1); load2<VEC>(bm, i, v0, v1);
        acc = fma(a0, masked(i, v0), acc);
        acc = fma(a1, masked(i + 1, v1), acc);
    }
    if ((n & 1) && t0 == 0) acc = fma(a[n - 1], masked(n - 1, bm[n - 1]), acc);
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
void launch_dot_zero_dirichlet(long long n, const double *a, double *bm, const uint8_t *mask, double *scratch, double *out, hipStream_t s) {
    if (aligned16(a) && aligned16(bm)) k_dot_masked_partial<true><<<DOT_BLOCKS, 256, 0, s>>>(n, a, bm, mask, scratch);
    else k_dot_masked_partial<false><<<DOT_BLOCKS, 256, 0, s>>>(n, a, bm, mask, scratch);
    k_dot_final<<<1, 256, 0, s>>>(DOT_BLOCKS, scratch, out);
    VFEM_HIP(hipGetLastError());
}

// x += alpha d, r -= alpha Ad and ||r||^2 of the new residual in one pass (alpha = sc[0] / sc[2]); the sum is partitioned and
// ordered as k_dot_partial's
template <bool VEC>
__global__ void __launch_bounds__(256) k_pcg_step_dot(long long n, double *__restrict__ x, double *__restrict__ r,
                                                      const double *__restrict__ dv, const double *__restrict__ Ad,
                                                      const double *__restrict__ sc, double *__restrict__ partial) {
    __shared__ double sh[4];
    const double alpha = sc[0] / sc[2];
    double acc = 0.0;
    const long long t0 = (long long) blockIdx.x * blockDim.x + threadIdx.x, nt = (long long) gridDim.x * blockDim.x;
    for (long long i = 2 * t0; i + 1 < n; i += 2 * nt) {
        double x0, x1, d0, d1, r0, r1, q0, q1;
        load2<VEC>(x, i, x0, x1); load2<VEC>(dv, i, d0, d1); load2<VEC>(r, i, r0, r1); load2<VEC>(Ad, i, q0, q1);
        store2<VEC>(x, i, fma(alpha, d0, x0), fma(alpha, d1, x1));
        const double n0 = fma(-alpha, q0, r0), n1 = fma(-alpha, q1, r1);
        store2<VEC>(r, i, n0, n1);
        acc = fma(n0, n0, acc);
        acc = fma(n1, n1, acc);
    }
    if ((n & 1) && t0 == 0) {
        const long long i = n - 1;
        x[i] = fma(alpha, dv[i], x[i]);
        const double rn = fma(-alpha, Ad[i], r[i]);
        r[i] = rn;
        acc = fma(rn, rn, acc);
    }
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
void launch_pcg_step_dot(long long n, double *x, double *r, const double *dv, const double *Ad, const double *sc, double *scratch,
                         double *rr_out, hipStream_t s) {
    if (aligned16(x) && aligned16(r) && aligned16(dv) && aligned16(Ad)) k_pcg_step_dot<true><<<DOT_BLOCKS, 256, 0, s>>>(n, x, r, dv, Ad, sc, scratch);
    else k_pcg_step_dot<false><<<DOT_BLOCKS, 256, 0, s>>>(n, x, r, dv, Ad, sc, scratch);
    k_dot_final<<<1, 256, 0, s>>>(DOT_BLOCKS, scratch, rr_out);
    VFEM_HIP(hipGetLastError());
}

void launch_dot(long long n, const double *a, const double *b, double *scratch, double *out, hipStream_t s) {
    if (aligned16(a) && aligned16(b)) k_dot_partial<true><<<DOT_BLOCKS, 256, 0, s>>>(n, a, b, scratch);
    else k_dot_partial<false><<<DOT_BLOCKS, 256, 0, s>>>(n, a, b, scratch);
    k_dot_final<<<1, 256, 0, s>>>(DOT_BLOCKS, scratch, out);
    VFEM_HIP(hipGetLastError());
}

// d = s + (rMr / rMr_old) d   (MG.hh:717-718); scalars live in HBM so the host never stalls on them
template <bool VEC>
__global__ void __launch_bounds__(256) k_pcg_direction(long long n, const double *__restrict__ sv, double *__restrict__ dv,
                                                       const double *__restrict__ sc, int first) {
    const double beta = first ? 0.0 : sc[0] / sc[1];
    const long long t0 = (long long) blockIdx.x * blockDim.x + threadIdx.x, nt = (long long) gridDim.x * blockDim.x;
    for (long long i = 2 * t0; i + 1 < n; i += 2 * nt) {
        double s0, s1, d0 = 0.0, d1 = 0.0;
        load2<VEC>(sv, i, s0, s1);
        if (!first) load2<VEC>(dv, i, d0, d1);
        store2<VEC>(dv, i, first ? s0 : fma(beta, d0, s0), first ? s1 : fma(beta, d1, s1));
    }
    if ((n & 1) && t0 == 0) dv[n - 1] = first ? sv[n - 1] : fma(beta, dv[n - 1], sv[n - 1]);
}
void launch_pcg_direction(long long n, const double *sv, double *dv, const double *sc, int first, hipStream_t s) {
    const unsigned grid = (unsigned) std::min<long long>((n / 2 + 255) / 256 + 1, 1 << 20);
    if (aligned16(sv) && aligned16(dv)) k_pcg_direction<true><<<grid, 256, 0, s>>>(n, sv, dv, sc, first);
    else k_pcg_direction<false><<<grid, 256, 0, s>>>(n, sv, dv, sc, first);
    VFEM_HIP(hipGetLastError());
}

// alpha = rMr / dAd ; x += alpha d ; r -= alpha Ad   (MG.hh:723-725)
__global__ void __launch_bounds__(256) k_pcg_step(long long n, double *__restrict__ x, double *__restrict__ r,
                                                  const double *__restrict__ dv, const double *__restrict__ Ad,
                                                  const double *__restrict__ sc) {
    const double alpha = sc[0] / sc[2];
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        x[i] = fma(alpha, dv[i], x[i]);
        r[i] = fma(-alpha, Ad[i], r[i]);
    }
}
void launch_pcg_step(long long n, double *x, double *r, const double *dv, const double *Ad, const double *sc, hipStream_t s) {
    k_pcg_step<<<grid_for(n, 256), 256, 0, s>>>(n, x, r, dv, Ad, sc);
    VFEM_HIP(hipGetLastError());
}

__global__ void k_shift_scalar(double *sc) { sc[1] = sc[0]; }
void launch_shift_scalar(double *sc, hipStream_t s) {
    k_shift_scalar<<<1, 1, 0, s>>>(sc);
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// complianceGradient (TPS.hh:730-751): g_e = -1/2 gamma rho^(gamma-1) (E0-Emin) u_e^T K0 u_e
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_compliance_gradient(Dims d, const double *__restrict__ K0, const double *__restrict__ rho,
                                                             double E0, double Emin, double gamma,
                                                             const double *__restrict__ u, double *__restrict__ g) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int i = blockIdx.z;
    if (k >= d.nz || j >= d.ny) return;
    const long long sx = (long long) d.NY * d.NZ, sy = d.NZ;
    const long long base = ((long long) i * d.NY + j) * d.NZ + k;
    double ue[24];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const long long nm = base + ((m >> 2) & 1) * sx + ((m >> 1) & 1) * sy + (m & 1);
        ue[3 * m] = u[3 * nm]; ue[3 * m + 1] = u[3 * nm + 1]; ue[3 * m + 2] = u[3 * nm + 2];
    }
    double en = 0.0;
#pragma unroll
    for (int r = 0; r < 24; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 24; ++c) acc = fma(K0[r * 24 + c], ue[c], acc);
        en = fma(ue[r], acc, en);
    }
    const long long e = ((long long) i * d.ny + j) * d.nz + k;
    const double r0 = rho[e];
    const double p = (gamma == 3.0) ? r0 * r0 : pow(r0, gamma - 1.0);
    g[e] = -0.5 * gamma * p * (E0 - Emin) * en;
}

void launch_compliance_gradient(const Dims &d, const double *K0, const double *rho, double E0, double Emin,
                                double gamma, const double *u, double *g, hipStream_t s) {
    dim3 blk(64, 4, 1), grd((d.nz + 63) / 64, (d.ny + 3) / 4, d.nx);
    k_compliance_gradient<<<grd, blk, 0, s>>>(d, K0, rho, E0, Emin, gamma, u, g);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem

// ------------------------------------------------------------------------------------------
// design-update path: SmoothingFilter / ProjectionFilter (TopologyOptimizationFilter.hh:55-79,
// 105-162), TotalVolumeConstraint (TopologyOptimizationConstraint.hh:21-34) and the OC candidate step
// (OptimalityCriterion.hh:47-50) as elementwise / small-stencil kernels on the element grid.
// ------------------------------------------------------------------------------------------
namespace vfem {

// box filter of radius r clipped to the grid; every output row is normalised by its in-bounds neighbour count.
// transpose = 0:  out_i = (1/c_i) sum_{k in N(i)} in_k        (A x)
// transpose = 1:  out_k = sum_{i in N(k)} in_i / c_i          (A^T g; the neighbourhood relation is symmetric)
__global__ void __launch_bounds__(256) k_box_filter(int nx, int ny, int nz, int r, const double *__restrict__ in,
                                                    double *__restrict__ out, int transpose) {
    const long long n = (long long) nx * ny * nz;
    for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long) gridDim.x * blockDim.x) {
        const int k = (int) (e % nz), j = (int) ((e / nz) % ny), i = (int) (e / ((long long) nz * ny));
        const int i0 = max(i - r, 0), i1 = min(i + r, nx - 1), j0 = max(j - r, 0), j1 = min(j + r, ny - 1);
        const int k0 = max(k - r, 0), k1 = min(k + r, nz - 1);
        double acc = 0.0;
        for (int a = i0; a <= i1; ++a)
            for (int b = j0; b <= j1; ++b)
                for (int c = k0; c <= k1; ++c) {
                    double v = in[((long long) a * ny + b) * nz + c];
                    if (transpose) {
                        const int ca = min(a + r, nx - 1) - max(a - r, 0) + 1, cb = min(b + r, ny - 1) - max(b - r, 0) + 1;
                        const int cc = min(c + r, nz - 1) - max(c - r, 0) + 1;
                        v /= (double) (ca * cb * cc);
                    }
                    acc += v;
                }
        if (!transpose) acc /= (double) ((i1 - i0 + 1) * (j1 - j0 + 1) * (k1 - k0 + 1));
        out[e] = acc;
    }
}
void launch_box_filter(int nx, int ny, int nz, int r, const double *in, double *out, int transpose, hipStream_t s) {
    k_box_filter<<<grid_for((long long) nx * ny * nz, 256), 256, 0, s>>>(nx, ny, nz, r, in, out, transpose);
    VFEM_HIP(hipGetLastError());
}

// mode 0: out = 0.5 (tanh(b/2) + tanh(b (x - 1/2))) / tanh(b/2);  mode 1: out = g * 0.5 b (1 - tanh^2(b (x - 1/2))) / tanh(b/2)
__global__ void __launch_bounds__(256) k_projection(long long n, double beta, const double *__restrict__ x,
                                                    const double *__restrict__ g, double *__restrict__ out, int mode) {
    const double th = tanh(0.5 * beta);
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const double t = tanh(beta * (x[i] - 0.5));
        out[i] = mode == 0 ? 0.5 * (th + t) / th : g[i] * 0.5 * beta * (1.0 - t * t) / th;
    }
}
void launch_projection(long long n, double beta, const double *x, const double *g, double *out, int mode, hipStream_t s) {
    k_projection<<<grid_for(n, 256), 256, 0, s>>>(n, beta, x, g, out, mode);
    VFEM_HIP(hipGetLastError());
}

// OC candidate: clip(x0 sqrt(dJ / (dc lambda)), max(x0 - m, 0), min(x0 + m, 1))
__global__ void __launch_bounds__(256) k_oc_candidate(long long n, const double *__restrict__ x0, const double *__restrict__ dJ,
                                                      const double *__restrict__ dc, double lambda, double m, double *__restrict__ out) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const double x = x0[i];
        double v = x * sqrt(dJ[i] / (dc[i] * lambda));
        v = fmax(fmax(v, x - m), 0.0);
        v = fmin(fmin(v, x + m), 1.0);
        out[i] = v;
    }
}
void launch_oc_candidate(long long n, const double *x0, const double *dJ, const double *dc, double lambda, double m, double *out,
                         hipStream_t s) {
    k_oc_candidate<<<grid_for(n, 256), 256, 0, s>>>(n, x0, dJ, dc, lambda, m, out);
    VFEM_HIP(hipGetLastError());
}

__global__ void __launch_bounds__(256) k_sum_partial(long long n, const double *__restrict__ a, double *__restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) acc += a[i];
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
void launch_sum(long long n, const double *a, double *scratch, double *out, hipStream_t s) {
    k_sum_partial<<<DOT_BLOCKS, 256, 0, s>>>(n, a, scratch);
    k_dot_final<<<1, 256, 0, s>>>(DOT_BLOCKS, scratch, out);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
