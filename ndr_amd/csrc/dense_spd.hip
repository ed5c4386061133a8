// Dense symmetric-positive-definite inverse for the coarsest multigrid level (the reference's exact coarsest solve,
// VoxelFEM/TensorProductSimulator.hh:834-865, is a CHOLMOD factorisation; here the inverse is formed once per operator
// update and applied as a GEMV).
//
// Everything is the build's own code on one stream with a FIXED summation order -- no atomics, no split-K, no library
// workspace shared between processes -- so the same matrix gives the same inverse bit for bit in every run, on every rank and
// whatever else runs on the device (rounds 1-2 called rocSOLVER potrf/potri here, which was measured non-reproducible when
// several processes factorised on one GPU at the same moment).
//
//   1. A = L L^T         blocked right-looking Cholesky, 64 x 64 tiles: diagonal tile (one workgroup, in LDS, together with the
//                        tile's inverse), panel L_ik = A_ik L_kk^-T, trailing update A_ij -= L_ik L_jk^T
//   2. X = L^-1          recursive halving, level by level: [[L11,0],[L21,L22]]^-1 = [[X11,0],[-X22 L21 X11, X22]]; all products
//                        of one level are independent tiles of ONE launch (two launches per level, log2(n/64) levels)
//   3. A^-1 = X^T X      one launch, tile (i,j) sums over the tile rows m >= max(i,j)
// All three share one 64 x 64 x 64 tile product (operands staged through LDS, 4 x 4 results per thread).
#include "vfem_internal.h"

namespace vfem {

namespace dense {
constexpr int T = 64;        // tile edge
constexpr int S = 65;        // LDS row stride in doubles (odd: column-wise stores are conflict-free)

// s[k][i] <- tile element with global row r and column c (c contiguous in memory); ROWS_ARE_K: the tile's rows are the
// summation index k (s[r][c]), otherwise its columns are (s[c][r])
template <bool ROWS_ARE_K>
__device__ __forceinline__ void stage(double (*s)[S], const double *__restrict__ g, long long ld) {
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = r0 + 4 * q;
        const double v = g[(long long) r * ld + c];
        if (ROWS_ARE_K) s[r][c] = v;
        else            s[c][r] = v;
    }
}
// acc[x][y] += sum_k sA[k][ti + 16 x] * sB[k][tj + 16 y], k ascending
__device__ __forceinline__ void mac(double acc[4][4], const double (*sA)[S], const double (*sB)[S]) {
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
#pragma unroll 4
    for (int k = 0; k < T; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { a[m] = sA[k][ti + 16 * m]; b[m] = sB[k][tj + 16 * m]; }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = fma(a[x], b[y], acc[x][y]);
    }
}
// acc += opA(A) opB(B) for one pair of 64 x 64 tiles; opA(A)[i][k] = TA ? A[k][i] : A[i][k], opB(B)[k][j] = TB ? B[j][k] : B[k][j]
template <bool TA, bool TB>
__device__ __forceinline__ void tile_product(double acc[4][4], const double *__restrict__ A, const double *__restrict__ B, long long ld,
                                             double (*sA)[S], double (*sB)[S]) {
    __syncthreads();                       // the previous product has been consumed
    stage<TA>(sA, A, ld);
    stage<!TB>(sB, B, ld);
    __syncthreads();
    mac(acc, sA, sB);
}
__device__ __forceinline__ void zero_acc(double acc[4][4]) {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
}
// C[ti + 16 x][tj + 16 y] = sign * acc (+ C when ACCUM)
template <bool ACCUM>
__device__ __forceinline__ void store_acc(const double acc[4][4], double *__restrict__ C, long long ld, double sign) {
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            double *p = C + (long long) (ti + 16 * x) * ld + tj + 16 * y;
            *p = ACCUM ? *p + sign * acc[x][y] : sign * acc[x][y];
        }
}
}  // namespace dense

// W (Np x Np, Np a multiple of 64) <- A (n x n) padded with the identity
__global__ void __launch_bounds__(256) k_dense_pad(long long n, long long Np, const double *__restrict__ A, double *__restrict__ W) {
    const long long gid = (long long) blockIdx.x * 256 + threadIdx.x;
    if (gid >= Np * Np) return;
    const long long r = gid / Np, c = gid - r * Np;
    W[gid] = (r < n && c < n) ? A[r * n + c] : (r == c ? 1.0 : 0.0);
}

// diagonal tile k: L_kk (written back, strict upper part zeroed) and D[k] = L_kk^-1, both by right-looking elimination.
// The tile and the inverse under construction live in REGISTERS (thread (r0, c) of CD_RG x 64 owns the rows r0 + CD_RG q of column
// c); a step publishes only column j of A and row j of X through LDS (double-buffered: one workgroup barrier per step)
// and every thread scales them itself (the same products a[r][j] * inv, x[j][c] * inv as an in-place scaling, so the result does not
// depend on the thread layout).  With both matrices in LDS and three barriers per step the tile took 97 us (a chain of dependent
// LDS round trips per row of the trailing update), as one wave without workgroup barriers 259 us; a 2187-dof coarsest level has 35
// such tiles in sequence.  info (0 on entry) receives 1 + the global index of the first non-positive pivot.
#ifndef VFEM_CD_THREADS
#define VFEM_CD_THREADS 1024
#endif
constexpr int CD_THREADS = VFEM_CD_THREADS;                    // 16 waves: the 64 steps of a tile are one chain of latencies, four waves per SIMD overlap them
constexpr int CD_RG = CD_THREADS / 64, CD_Q = 64 / CD_RG;      // thread (r0, c) owns rows r0 + CD_RG q, q < CD_Q, of column c
__global__ void __launch_bounds__(CD_THREADS) k_chol_diag(long long Np, int k, double *__restrict__ L, double *__restrict__ D, int *__restrict__ info) {
    using namespace dense;
    static_assert(T == 64, "one lane per column of the tile");
    __shared__ double col[2][T], row[2][T];
    double *tile = L + ((long long) k * T) * Np + (long long) k * T;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    double a[CD_Q], x[CD_Q];
#pragma unroll
    for (int q = 0; q < CD_Q; ++q) {
        const int r = r0 + CD_RG * q;
        a[q] = tile[(long long) r * Np + c];
        x[q] = r == c ? 1.0 : 0.0;
    }
    for (int j = 0; j < T; ++j) {
        const int buf = j & 1;
        // publish column j of A (pivot included) and row j of X as they stand
        if (c == j) {
#pragma unroll
            for (int q = 0; q < CD_Q; ++q) col[buf][r0 + CD_RG * q] = a[q];
        }
        if (r0 == (j % CD_RG)) {
            double xv = 0.0;
#pragma unroll
            for (int q = 0; q < CD_Q; ++q) xv = (j / CD_RG) == q ? x[q] : xv;
            row[buf][c] = xv;
        }
        __syncthreads();
        const double piv = col[buf][j];
        if (!(piv > 0.0) && threadIdx.x == 0 && *info == 0) *info = k * T + j + 1;      // (also catches NaN)
        // 1 / sqrt(piv) from the hardware estimate and two Newton steps (a short dependent chain: the correctly rounded sqrt and
        // division cost ~60 dependent instructions per step, and the 64 steps of a tile are one chain); deterministic, within an
        // ulp or two of the rounded values
        double inv = __builtin_amdgcn_rsq(piv);
        inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
        inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
        const double ljj = piv * inv;
        const double lcj = col[buf][c] * inv;          // L[c][j] (used where c > j)
        const double xjc = row[buf][c] * inv;          // X[j][c] (used where c <= j)
#pragma unroll
        for (int q = 0; q < CD_Q; ++q) {               // (selects, no lane-dependent branches: the conditions differ from lane to lane)
            if (CD_RG * q + CD_RG - 1 < j) continue;   // all rows of this slot lie above the pivot row: finished (uniform over the workgroup)
            const int r = r0 + CD_RG * q;
            const double lrj = col[buf][r] * inv;      // L[r][j]
            const double an = (c > j && c <= r) ? fma(-lrj, lcj, a[q]) : (c == j ? lrj : a[q]);
            const double xn = c <= j ? fma(-lrj, xjc, x[q]) : x[q];
            a[q] = r > j ? an : ((r == j && c == j) ? ljj : a[q]);
            x[q] = r > j ? xn : ((r == j && c <= j) ? xjc : x[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < CD_Q; ++q) {
        const int r = r0 + CD_RG * q;
        tile[(long long) r * Np + c] = c <= r ? a[q] : 0.0;
        D[((long long) k * T + r) * T + c] = c <= r ? x[q] : 0.0;
    }
}

// panel below diagonal tile k: L_ik = A_ik L_kk^-T = A_ik D[k]^T, i = k + 1 + blockIdx.x
__global__ void __launch_bounds__(256) k_chol_panel(long long Np, int k, double *__restrict__ L, const double *__restrict__ D) {
    using namespace dense;
    __shared__ double sA[T][S], sB[T][S];
    const int i = k + 1 + blockIdx.x;
    double *Aik = L + ((long long) i * T) * Np + (long long) k * T;
    double acc[4][4];
    zero_acc(acc);
    // opB[m][c] = D[c][m]: the rows of the stored tile are the output columns
    stage<false>(sA, Aik, Np);
    stage<false>(sB, D + (long long) k * T * T, T);
    __syncthreads();
    mac(acc, sA, sB);
    __syncthreads();                                // every thread has read the old A_ik from LDS, none from memory
    store_acc<false>(acc, Aik, Np, 1.0);
}

// trailing update for panel k: A_ij -= L_ik L_jk^T, k < j <= i
__global__ void __launch_bounds__(256) k_chol_update(long long Np, int k, double *__restrict__ L) {
    using namespace dense;
    __shared__ double sA[T][S], sB[T][S];
    const int i = k + 1 + blockIdx.y, j = k + 1 + blockIdx.x;
    if (j > i) return;
    double acc[4][4];
    zero_acc(acc);
    tile_product<false, true>(acc, L + ((long long) i * T) * Np + (long long) k * T, L + ((long long) j * T) * Np + (long long) k * T, Np, sA, sB);
    store_acc<true>(acc, L + ((long long) i * T) * Np + (long long) j * T, Np, -1.0);
}

// X <- diag(D[0], D[1], ...), zero elsewhere
__global__ void __launch_bounds__(256) k_trtri_init(long long Np, const double *__restrict__ D, double *__restrict__ X) {
    const long long gid = (long long) blockIdx.x * 256 + threadIdx.x;
    if (gid >= Np * Np) return;
    const long long r = gid / Np, c = gid - r * Np;
    X[gid] = (r / 64 == c / 64) ? D[(r / 64) * 4096 + (r % 64) * 64 + (c % 64)] : 0.0;
}

// one level of the recursive triangular inverse; groups of 2m tile rows: a = first m, b = the rest (maybe fewer than m).
//   STEP 1:  Tm_{bi,aj} = sum_{kk in a, kk >= aj} L_{bi,kk} X_{kk,aj}
//   STEP 2:  X_{bi,aj}  = - sum_{kk in b, kk <= bi} X_{bi,kk} Tm_{kk,aj}
template <int STEP>
__global__ void __launch_bounds__(256) k_trtri_level(long long Np, int m, const double *__restrict__ L, double *__restrict__ X, double *__restrict__ Tm) {
    using namespace dense;
    __shared__ double sA[T][S], sB[T][S];
    const int bi = blockIdx.y, aj = blockIdx.x;
    const int g = bi / (2 * m);
    if (aj / (2 * m) != g || bi - g * 2 * m < m || aj - g * 2 * m >= m) return;
    const int a0 = g * 2 * m, b0 = a0 + m;
    double acc[4][4];
    zero_acc(acc);
    if (STEP == 1) {
        for (int kk = aj; kk < b0; ++kk)
            tile_product<false, false>(acc, L + ((long long) bi * T) * Np + (long long) kk * T, X + ((long long) kk * T) * Np + (long long) aj * T, Np, sA, sB);
        store_acc<false>(acc, Tm + ((long long) bi * T) * Np + (long long) aj * T, Np, 1.0);
    } else {
        for (int kk = b0; kk <= bi; ++kk)
            tile_product<false, false>(acc, X + ((long long) bi * T) * Np + (long long) kk * T, Tm + ((long long) kk * T) * Np + (long long) aj * T, Np, sA, sB);
        store_acc<false>(acc, X + ((long long) bi * T) * Np + (long long) aj * T, Np, -1.0);
    }
}

// lower tiles of X^T X: Out_ij = sum_{mm >= i} X_{mm,i}^T X_{mm,j}, j <= i
__global__ void __launch_bounds__(256) k_lauum(long long Np, int nb, const double *__restrict__ X, double *__restrict__ Out) {
    using namespace dense;
    __shared__ double sA[T][S], sB[T][S];
    const int i = blockIdx.y, j = blockIdx.x;
    if (j > i) return;
    double acc[4][4];
    zero_acc(acc);
    for (int mm = i; mm < nb; ++mm)
        tile_product<true, false>(acc, X + ((long long) mm * T) * Np + (long long) i * T, X + ((long long) mm * T) * Np + (long long) j * T, Np, sA, sB);
    store_acc<false>(acc, Out + ((long long) i * T) * Np + (long long) j * T, Np, 1.0);
}

// A (n x n, full symmetric) <- lower triangle of W (Np x Np)
__global__ void __launch_bounds__(256) k_dense_unpad_sym(long long n, long long Np, const double *__restrict__ W, double *__restrict__ A) {
    const long long gid = (long long) blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * n) return;
    const long long r = gid / n, c = gid - r * n;
    A[gid] = r >= c ? W[r * Np + c] : W[c * Np + r];
}

void dense_spd_inverse(long long n, double *A, DenseWork &w, hipStream_t s) {
    using namespace dense;
    if (n <= 0) return;
    const long long Np = (n + T - 1) / T * T;
    const int nb = (int) (Np / T);
    w.L.reserve((size_t) Np * Np);
    w.X.reserve((size_t) Np * Np);
    w.Tm.reserve((size_t) Np * Np);
    w.D.reserve((size_t) nb * T * T);
    w.info.reserve(1);
    const unsigned gsq = (unsigned) ((Np * Np + 255) / 256);
    VFEM_HIP(hipMemsetAsync(w.info.p, 0, sizeof(int), s));
    k_dense_pad<<<gsq, 256, 0, s>>>(n, Np, A, w.L.p);
    for (int k = 0; k < nb; ++k) {
        k_chol_diag<<<1, CD_THREADS, 0, s>>>(Np, k, w.L.p, w.D.p, w.info.p);
        const int rest = nb - k - 1;
        if (rest > 0) {
            k_chol_panel<<<rest, 256, 0, s>>>(Np, k, w.L.p, w.D.p);
            k_chol_update<<<dim3(rest, rest), 256, 0, s>>>(Np, k, w.L.p);
        }
    }
    VFEM_HIP(hipGetLastError());
    int info = 0;
    VFEM_HIP(hipMemcpyAsync(&info, w.info.p, sizeof(int), hipMemcpyDeviceToHost, s));
    VFEM_HIP(hipStreamSynchronize(s));
    if (info != 0) throw Error("coarsest-level stiffness matrix is not positive definite (pivot " + std::to_string(info) + " of " + std::to_string(n) + ")");
    k_trtri_init<<<gsq, 256, 0, s>>>(Np, w.D.p, w.X.p);
    for (int m = 1; m < nb; m *= 2) {
        k_trtri_level<1><<<dim3(nb, nb), 256, 0, s>>>(Np, m, w.L.p, w.X.p, w.Tm.p);
        k_trtri_level<2><<<dim3(nb, nb), 256, 0, s>>>(Np, m, w.L.p, w.X.p, w.Tm.p);
    }
    k_lauum<<<dim3(nb, nb), 256, 0, s>>>(Np, nb, w.X.p, w.Tm.p);
    k_dense_unpad_sym<<<(unsigned) ((n * n + 255) / 256), 256, 0, s>>>(n, Np, w.Tm.p, A);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
