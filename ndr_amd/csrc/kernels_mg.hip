// Multigrid / operator kernels for gfx950 (wave64).  Lanes run along z (the contiguous axis) so
// that nodal [numNodes][3] fp64 fields are read in contiguous 24-byte-per-lane runs.
//
// Reference semantics (paths relative to the reference checkout):
//   applyK            VoxelFEM/TensorProductSimulator.hh:905-952
//   m_smoothNode      VoxelFEM/MultigridSolver.hh:193-265   (block GS, component-sequential solve)
//   colour order      VoxelFEM/MultigridSolver.hh:285-326
//   restriction       VoxelFEM/MultigridSolver.hh:146-161
//   interpolation     VoxelFEM/MultigridSolver.hh:116-141
//   buildPESCoarse    VoxelFEM/MultigridSolver.hh:604-669
// The reference scatters per element into thread-private arrays; here every operator is a
// node-centred gather (no atomics, deterministic summation order).
#include "vfem_internal.h"
#include "device_utils.h"
#include "gs_coef.h"


#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>

namespace vfem {

__device__ __forceinline__ long long nidx(const Dims &d, int i, int j, int k) {
    return ((long long) i * d.NY + j) * d.NZ + k;
}
__device__ __forceinline__ long long eidx(const Dims &d, int i, int j, int k) {
    return ((long long) i * d.ny + j) * d.nz + k;
}

// ------------------------------------------------------------------------------------------
// matrix-free node operator: S = sum_e (Ke_e[rows of n] . u_e), M = sum_e Ke_e[n,n block]
//   KIND 0: Ke_e = E[e] * K0                          (TPS.hh:933-951, MG.hh:199-220)
//   KIND 1: Ke_e = sum_f Efine[child f of e] * cK0[f] (MG.hh:639-657), never materialised
// ------------------------------------------------------------------------------------------
template <int KIND, bool WITH_M>
__device__ __forceinline__ void mf_node(const Dims &d, const double *__restrict__ K, const double *__restrict__ E,
                                        const double *__restrict__ u, int i, int j, int k, double S[3], double M[9]) {
    S[0] = S[1] = S[2] = 0.0;
    if (WITH_M) {
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = 0.0;
    }
    const long long sx = (long long) d.NY * d.NZ, sy = d.NZ;
#pragma unroll
    for (int slot = 0; slot < 8; ++slot) {
        const int di = (slot >> 2) & 1, dj = (slot >> 1) & 1, dk = slot & 1;
        const int ex = i - 1 + di, ey = j - 1 + dj, ez = k - 1 + dk;
        if (ex < 0 || ex >= d.nx || ey < 0 || ey >= d.ny || ez < 0 || ez >= d.nz) continue;
        const int li = ((1 - di) * 2 + (1 - dj)) * 2 + (1 - dk);
        const long long base = nidx(d, ex, ey, ez);
        double ue[24];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const long long nm = base + ((m >> 2) & 1) * sx + ((m >> 1) & 1) * sy + (m & 1);
            ue[3 * m + 0] = u[3 * nm + 0];
            ue[3 * m + 1] = u[3 * nm + 1];
            ue[3 * m + 2] = u[3 * nm + 2];
        }
        if (KIND == 0) {
            const double Ee = E[eidx(d, ex, ey, ez)];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double *row = K + (3 * li + r) * 24;
                double acc = 0.0;
#pragma unroll
                for (int c = 0; c < 24; ++c) acc = fma(row[c], ue[c], acc);
                S[r] = fma(Ee, acc, S[r]);
                if (WITH_M) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) M[r * 3 + c] = fma(Ee, row[3 * li + c], M[r * 3 + c]);
                }
            }
        } else {
            const long long nyf = 2LL * d.ny, nzf = 2LL * d.nz;
            for (int f = 0; f < 8; ++f) {
                const int fx = (f >> 2) & 1, fy = (f >> 1) & 1, fz = f & 1;
                const double Ef = E[((2LL * ex + fx) * nyf + (2LL * ey + fy)) * nzf + (2LL * ez + fz)];
                const double *Kf = K + f * 576;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double *row = Kf + (3 * li + r) * 24;
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < 24; ++c) acc = fma(row[c], ue[c], acc);
                    S[r] = fma(Ef, acc, S[r]);
                    if (WITH_M) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) M[r * 3 + c] = fma(Ef, row[3 * li + c], M[r * 3 + c]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// apply / residual, matrix-free gather form
// ------------------------------------------------------------------------------------------
template <int KIND, bool RES>
__global__ void __launch_bounds__(256) k_apply_gather(Dims d, const double *__restrict__ K, const double *__restrict__ E,
                                                      const double *__restrict__ u, const double *__restrict__ b,
                                                      const uint8_t *__restrict__ mask, double *__restrict__ out) {
    const int q = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x;       // lanes packed over the nodes of an x-plane (rows of 2^k + 1 nodes)
    if (q >= d.NY * d.NZ) return;
    const int j = q / d.NZ, k = q - j * d.NZ, i = blockIdx.z;
    double S[3], M[9];
    mf_node<KIND, false>(d, K, E, u, i, j, k, S, M);
    const long long n = nidx(d, i, j, k);
    if (RES) {
        const uint8_t m = mask ? mask[n] : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * n + c] = ((m >> c) & 1) ? 0.0 : b[3 * n + c] - S[c];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * n + c] = S[c];
    }
}

void launch_apply_gather(const Dims &d, OpKind kind, const double *K, const double *E, const double *u,
                         const double *b, const uint8_t *mask, int res, double *out, hipStream_t s) {
    dim3 blk(64, 4, 1), grd((d.NY * d.NZ + 255) / 256, 1, d.NX);
    if (kind == OP_MF0) {
        if (res) k_apply_gather<0, true><<<grd, blk, 0, s>>>(d, K, E, u, b, mask, out);
        else     k_apply_gather<0, false><<<grd, blk, 0, s>>>(d, K, E, u, b, mask, out);
    } else {
        if (res) k_apply_gather<1, true><<<grd, blk, 0, s>>>(d, K, E, u, b, mask, out);
        else     k_apply_gather<1, false><<<grd, blk, 0, s>>>(d, K, E, u, b, mask, out);
    }
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// multicoloured block Gauss-Seidel, matrix-free levels.  One launch per colour (8 per sweep);
// nodes of one colour share no element, so the update order inside a colour is immaterial.
// ------------------------------------------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(256) k_gs_color_mf(Dims d, const double *__restrict__ K, const double *__restrict__ E,
                                                     double *__restrict__ u, const double *__restrict__ b,
                                                     const uint8_t *__restrict__ mask, int cx, int cy, int cz,
                                                     int forward) {
    const int k = 2 * (blockIdx.x * 64 + threadIdx.x) + cz;
    const int j = 2 * (blockIdx.y * 4 + threadIdx.y) + cy;
    const int i = 2 * blockIdx.z + cx;
    if (k >= d.NZ || j >= d.NY || i >= d.NX) return;
    double S[3], M[9];
    mf_node<KIND, true>(d, K, E, u, i, j, k, S, M);
    const long long n = nidx(d, i, j, k);
    double bms[3], ud[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bms[c] = b[3 * n + c] - S[c];
    gs_solve(bms, M, mask[n], forward != 0, ud);
#pragma unroll
    for (int c = 0; c < 3; ++c) u[3 * n + c] += ud[c];
}

// ------------------------------------------------------------------------------------------
// Row-streaming colour sweep for the finest level (Ke = E_e K0).
//
// The plain kernel above has every lane issue ~190 strided 8-byte loads (colour stride 2 nodes = 48 B), which
// saturates the address path long before the 576 FMA per node matter.  Here a wave owns 64 colour nodes of one
// grid row and walks the 9 neighbouring rows (dx,dy): each row segment (129 nodes = 3.1 KB) is fetched with
// dense 8-byte-per-lane loads, staged in a per-wave LDS buffer, and every lane then reads its z-1,z,z+1
// neighbours with conflict-free 16-byte LDS reads.  Per incident element an unscaled 3-vector T_e accumulates
// K0[rows of n] . u_e; the element moduli enter once at the end (S = sum_e E_e T_e, M = sum_e E_e K0_nn).
// The next row's loads are in flight while the current row is consumed.
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Resident coefficients.  For a box voxel with an isotropic tensor K0 has 36 distinct magnitudes (closed form in
// capi.hip, vfem_sim::update_k0):
//   same component a:   K0[(n,a),(m,a)] depends on a and on which of the three index bits of n and m agree   -> 3 x 8 values
//   components a != b:  K0[(n,a),(m,b)] = sign * C,  C chosen by the axis pair, by the agreement of the third axis' bits and by
//                       whether tau1 = s(n_lo) s(m_hi) equals tau2 = s(n_hi) s(m_lo)  (lo < hi the two axes, s(bit) = +-1);
//                       sign = tau1 for a < b, tau2 for a > b                                                -> 3 x 2 x 2 values
// so a sweep can keep the whole matrix in 72 SGPRs and select entry and sign at compile time: no coefficient loads inside
// the node loop (the table version spends one 24-double scalar load per 18 multiply-adds).  build_gs_coef() fills the table
// from K0 and checks that EVERY entry of K0 is reproduced bit for bit; otherwise the table kernels are used.
// ------------------------------------------------------------------------------------------
bool build_gs_coef(const double *K0, double *coef /* 36 */) {
    bool have[36] = {false};
    for (int q = 0; q < 36; ++q) coef[q] = 0.0;
    for (int n = 0; n < 8; ++n)
        for (int a = 0; a < 3; ++a)
            for (int m = 0; m < 8; ++m)
                for (int b = 0; b < 3; ++b) {
                    const KSel k = ksel(n, a, m, b);
                    const double v = k.neg ? -K0[(3 * n + a) * 24 + 3 * m + b] : K0[(3 * n + a) * 24 + 3 * m + b];
                    if (!have[k.idx]) { coef[k.idx] = v; have[k.idx] = true; }
                    else if (coef[k.idx] != v) return false;
                }
    return true;
}

#ifndef VFEM_GS_PF
#define VFEM_GS_PF 9
#endif
#ifndef VFEM_GS_MINW
#define VFEM_GS_MINW 1
#endif
constexpr int GS_PF = VFEM_GS_PF;        // node rows requested ahead of the one being consumed (9 = all of a segment's rows up front)
constexpr int GS_ROWBUF = 448;   // 129 nodes x 3 doubles = 387, padded to 7 x 64 so that every staging store is unconditional

// Host-side layout of the coefficient table consumed by k_gs_rows_mf0 (same loop nest as the kernel):
// 64 groups of 12 doubles: [r][c] = K0[(3 ln + r)*24 + 3 lm + c] for one (incident element, element node) pair,
// padded to 12; then 8 groups of 12 doubles with the diagonal block of local node ln = group index.
void build_gs_table(const double *K0, double *tab /* 72*12 */) {
    int g = 0;
    for (int r9 = 0; r9 < 9; ++r9) {
        const int dx = r9 / 3 - 1, dy = r9 % 3 - 1;
        for (int di = 0; di < 2; ++di)
            for (int mx = 0; mx < 2; ++mx) {
                if (di - 1 + mx != dx) continue;
                for (int dj = 0; dj < 2; ++dj)
                    for (int my = 0; my < 2; ++my) {
                        if (dj - 1 + my != dy) continue;
                        for (int dk = 0; dk < 2; ++dk) {
                            const int ln = (1 - di) * 4 + (1 - dj) * 2 + (1 - dk);
                            for (int mz = 0; mz < 2; ++mz) {
                                const int lm = mx * 4 + my * 2 + mz;
                                for (int r = 0; r < 3; ++r)
                                    for (int c = 0; c < 3; ++c) tab[g * 12 + r * 3 + c] = K0[(3 * ln + r) * 24 + 3 * lm + c];
                                for (int q = 9; q < 12; ++q) tab[g * 12 + q] = 0.0;
                                ++g;
                            }
                        }
                    }
            }
    }
    for (int ln = 0; ln < 8; ++ln) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) tab[(64 + ln) * 12 + r * 3 + c] = K0[(3 * ln + r) * 24 + 3 * ln + c];
        for (int q = 9; q < 12; ++q) tab[(64 + ln) * 12 + q] = 0.0;
    }
}

// one z-segment (64 colour nodes from colour index l0) of the colour row (x, y): the work of one wave
// RES: `tab` holds the 36 resident coefficients (build_gs_coef) instead of the 72 x 12 table (build_gs_table); the arithmetic
// (order of the multiply-adds, operands) is the same in both forms, so the results agree bit for bit
template <bool RES>
__device__ __forceinline__ void gs_row_segment_mf0(const Dims &d, const double *__restrict__ tab, const GsCoef &ck, const double *__restrict__ E,
                                                   double *__restrict__ u, const double *__restrict__ b,
                                                   const uint8_t *__restrict__ mask, int x, int y, int l0, int cz, int forward,
                                                   double *buf) {
    auto coef = [&](auto nn, auto aa, auto mm, auto bb) -> double {   // K0[(n,a),(m,b)] from the resident table, compile-time selection
        constexpr KSel k = ksel(decltype(nn)::value, decltype(aa)::value, decltype(mm)::value, decltype(bb)::value);
        constexpr int i = k.idx;
        const double v = i < 32 ? ck.c[i < 32 ? i / 8 : 0][i < 32 ? i % 8 : 0] : ck.t[i >= 32 ? i - 32 : 0];
        return k.neg ? -v : v;
    };
    const int lane = threadIdx.x;
    const int z = 2 * (l0 + lane) + cz;
    const int zlo = 2 * l0 + cz - 1;                    // first node of the staged segment
    const bool node_ok = z < d.NZ;

    // staged doubles q = lane + 64 s; loads are unconditional with the offset clamped into the row: values
    // outside the grid only ever meet elements outside the grid, whose modulus is 0
    unsigned qc[7];                                     // byte offsets: scalar row base + 32-bit lane offset addressing
#pragma unroll
    for (int s7 = 0; s7 < 7; ++s7) {
        int q = lane + 64 * s7;
        const int lo = -3 * zlo > 0 ? -3 * zlo : 0, hi = 3 * (d.NZ - zlo) - 1;
        q = q < lo ? lo : (q > hi ? hi : q);
        qc[s7] = 8u * (unsigned) q;
    }

    double T[8][3];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) { T[sl][0] = 0.0; T[sl][1] = 0.0; T[sl][2] = 0.0; }

    // everything the epilogue needs from memory is requested here, before the row loop: left in the epilogue these were eleven
    // dependent round trips (8 moduli one by one behind the coefficient loads, then b and the mask, then u) and the waves
    // spent 58 % of their life in s_waitcnt (SQ_WAIT_ANY)
    const int zc = node_ok ? z : d.NZ - 1;
    const long long n = nidx(d, x, y, zc);
    double Ev8[8], bv[3], uself[3] = {0.0, 0.0, 0.0};
    // The two moduli of an element column (ez = z - 1, z) are neighbours in memory: ONE 16-byte load per column, and with the lanes
    // two elements apart the four loads of a wave are dense (as eight 8-byte loads every instruction used half of each line it
    // touched; the moduli, right-hand side and mask loads were a quarter of the sweep's time, profiles/r02_gs_experiments.json).
    // Loads are unconditional at clamped indices and masked afterwards.  (needs nz >= 2; the callers use the plain kernels below)
    typedef double d2u_t __attribute__((ext_vector_type(2), aligned(8)));
    const int ez0 = zc - 1;
    const int ez0c = ez0 < 0 ? 0 : (ez0 > d.nz - 2 ? d.nz - 2 : ez0), esh = ez0c - ez0;     // +1 at the lower face, -1 at the upper one
#pragma unroll
    for (int xy = 0; xy < 4; ++xy) {
        const int di = xy >> 1, dj = xy & 1;
        const int ex = x - 1 + di, ey = y - 1 + dj;
        const bool okxy = ex >= 0 && ex < d.nx && ey >= 0 && ey < d.ny;
        const int exc = ex < 0 ? 0 : (ex > d.nx - 1 ? d.nx - 1 : ex), eyc = ey < 0 ? 0 : (ey > d.ny - 1 ? d.ny - 1 : ey);
        const d2u_t pr = *reinterpret_cast<const d2u_t *>(E + eidx(d, exc, eyc, ez0c));
        const double lo = esh == 0 ? pr[0] : (esh < 0 ? pr[1] : 0.0);      // modulus of ez = z - 1 (none below the grid)
        const double hi = esh == 0 ? pr[1] : (esh > 0 ? pr[0] : 0.0);      // modulus of ez = z     (none above the grid)
        Ev8[di * 4 + dj * 2 + 0] = (okxy && ez0 >= 0) ? lo : 0.0;
        Ev8[di * 4 + dj * 2 + 1] = (okxy && ez0 + 1 < d.nz) ? hi : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) bv[c] = b[3 * n + c];
    const uint8_t mk = mask[n];

    // All nine rows are requested before the first one is consumed (63 loads per lane in flight, retired in order by counted
    // vmcnt waits).  With one row requested ahead, as this loop was first written, a segment cost nine dependent memory round
    // trips (~12 us per segment and wave against 1.5 us of arithmetic: the sweep was latency-bound at a third of the VALU rate).
    double pre[GS_PF][7];
    auto issue = [&](int r9) {
        int gx = x + r9 / 3 - 1, gy = y + r9 % 3 - 1;
        gx = gx < 0 ? 0 : (gx > d.NX - 1 ? d.NX - 1 : gx);
        gy = gy < 0 ? 0 : (gy > d.NY - 1 ? d.NY - 1 : gy);
        // the row is the same for the whole wave: keep its base in SGPRs so that the loads use scalar base + 32-bit lane offset
        const long long ro = 3 * (((long long) gx * d.NY + gy) * d.NZ + zlo);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) (ro & 0xffffffffLL));
        const int hi = __builtin_amdgcn_readfirstlane((int) (ro >> 32));
        const double *rowp = u + (((long long) hi << 32) | (long long) lo);
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) pre[r9 % GS_PF][s7] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(rowp) + qc[s7]);
    };
#pragma unroll
    for (int r9 = 0; r9 < GS_PF; ++r9) issue(r9);
    // row r9 of the 3 x 3 rows (dx, dy) around the node: staged values -> LDS -> the lane's three neighbours in z
    auto stage_row = [&](int r9, double u3[3][3]) {
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) buf[lane + 64 * s7] = pre[r9 % GS_PF][s7];
        if (r9 + GS_PF < 9) issue(r9 + GS_PF);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n3 = 0; n3 < 3; ++n3)
#pragma unroll
            for (int c = 0; c < 3; ++c) u3[n3][c] = buf[6 * lane + 3 * n3 + c];
        __builtin_amdgcn_wave_barrier();
        if (r9 == 4) {                                  // this row holds the node itself (the middle one of the lane's three)
#pragma unroll
            for (int c = 0; c < 3; ++c) uself[c] = u3[1][c];
        }
    };
    if constexpr (RES) {
        // elements touching row (dx,dy): (di,mx) with di-1+mx == dx, (dj,my) with dj-1+my == dy; same nest and order as below
        static_for<9>([&](auto r9c) {
            constexpr int r9 = decltype(r9c)::value, dx = r9 / 3 - 1, dy = r9 % 3 - 1;
            double u3[3][3];
            stage_row(r9, u3);
            static_for<16>([&](auto ec) {
                constexpr int e = decltype(ec)::value, di = (e >> 3) & 1, mx = (e >> 2) & 1, dj = (e >> 1) & 1, my = e & 1;
                if constexpr (di - 1 + mx == dx && dj - 1 + my == dy) {
                    static_for<2>([&](auto dkc) {
                        constexpr int dk = decltype(dkc)::value, sl = di * 4 + dj * 2 + dk;
                        constexpr int ln = (1 - di) * 4 + (1 - dj) * 2 + (1 - dk);
                        static_for<2>([&](auto mzc) {
                            constexpr int mz = decltype(mzc)::value, n3 = dk + mz, lm = mx * 4 + my * 2 + mz;
                            static_for<9>([&](auto rc) {
                                constexpr int r = decltype(rc)::value / 3, c = decltype(rc)::value % 3;
                                const double kv = coef(std::integral_constant<int, ln>{}, std::integral_constant<int, r>{},
                                                       std::integral_constant<int, lm>{}, std::integral_constant<int, c>{});
                                T[sl][r] = fma(kv, u3[n3][c], T[sl][r]);
                            });
                        });
                    });
                }
            });
        });
    } else {
    int hg = 0;
#pragma unroll
    for (int r9 = 0; r9 < 9; ++r9) {
        const int dx = r9 / 3 - 1, dy = r9 % 3 - 1;
        double u3[3][3];
        stage_row(r9, u3);

        // elements touching row (dx,dy): (di,mx) with di-1+mx == dx, (dj,my) with dj-1+my == dy
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int mx = 0; mx < 2; ++mx) {
                if (di - 1 + mx != dx) continue;
#pragma unroll
                for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                    for (int my = 0; my < 2; ++my) {
                        if (dj - 1 + my != dy) continue;
#pragma unroll
                        for (int dk = 0; dk < 2; ++dk) {
                            const int sl = di * 4 + dj * 2 + dk;
                            {   // the two groups (mz = 0, 1) of this element slot in one 24-double scalar load: 18 FMAs per round trip
                                d8_t c0, c1, c2;
                                sload24(tab, hg * 96, c0, c1, c2);
                                hg += 2;
#pragma unroll
                                for (int mz = 0; mz < 2; ++mz) {
                                    const int n3 = dk + mz;            // dz + 1
#pragma unroll
                                    for (int r = 0; r < 3; ++r)
#pragma unroll
                                        for (int c = 0; c < 3; ++c) {
                                            const int q = 12 * mz + r * 3 + c;                   // position in the 24 doubles
                                            const double kv = q < 8 ? c0[q < 8 ? q : 0] : (q < 16 ? c1[(q >= 8 && q < 16) ? q - 8 : 0] : c2[q >= 16 ? q - 16 : 0]);
                                            T[sl][r] = fma(kv, u3[n3][c], T[sl][r]);
                                        }
                                }
                                // pin: these FMAs retire before the next coefficient load is issued
                                asm volatile("" : "+v"(T[sl][0]), "+v"(T[sl][1]), "+v"(T[sl][2]));
                            }
                        }
                    }
            }
    }
    }
    if (!node_ok) return;
    double S[3] = {0.0, 0.0, 0.0}, M[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) M[q] = 0.0;
    if constexpr (RES) {
        static_for<8>([&](auto gc) {
            constexpr int ln = decltype(gc)::value, sl = 7 - ln;      // local index of the node in its element; slot = complement
            const double Ee = Ev8[sl];
            static_for<3>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                S[r] = fma(Ee, T[sl][r], S[r]);
                static_for<3>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    const double kv = coef(std::integral_constant<int, ln>{}, std::integral_constant<int, r>{},
                                           std::integral_constant<int, ln>{}, std::integral_constant<int, c>{});
                    M[3 * r + c] = fma(Ee, kv, M[3 * r + c]);
                });
            });
        });
    } else {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        d8_t k0;
        d4_t k1;
        sload12(tab, (64 + g) * 96, k0, k1);
        double kk[9];
#pragma unroll
        for (int q = 0; q < 8; ++q) kk[q] = k0[q];
        kk[8] = k1[0];
        {
            const int h = 0;
            const int ln = g;                            // local index of the node in its element
            const int sl = 7 - ln;                       // slot (di,dj,dk) = complement of ln
            const double Ee = Ev8[sl];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                S[r] = fma(Ee, T[sl][r], S[r]);
#pragma unroll
                for (int c = 0; c < 3; ++c) M[3 * r + c] = fma(Ee, kk[h * 9 + 3 * r + c], M[3 * r + c]);
            }
        }
        asm volatile("" : "+v"(M[0]), "+v"(M[4]), "+v"(M[8]), "+v"(S[0]));
    }
    }
    double bms[3], ud[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bms[c] = bv[c] - S[c];
    gs_solve(bms, M, mk, forward != 0, ud);
#pragma unroll
    for (int c = 0; c < 3; ++c) u[3 * n + c] = uself[c] + ud[c];      // the node's own value came through the staged row
}

template <bool RES>
__global__ void __launch_bounds__(256, VFEM_GS_MINW) k_gs_rows_mf0(Dims d, const double *__restrict__ tab, const double *__restrict__ E,
                                                     double *__restrict__ u, const double *__restrict__ b,
                                                     const uint8_t *__restrict__ mask, int cx, int cy, int cz, int forward) {
    __shared__ double rowbuf[4][GS_ROWBUF];
    const int x = 2 * blockIdx.z + cx;
    // (threadIdx.y is the wave index: telling the compiler that y is wave-uniform moves the row / column address arithmetic to
    // the scalar unit and out of the vector registers)
    const int y = __builtin_amdgcn_readfirstlane(2 * (blockIdx.y * 4 + threadIdx.y) + cy);
    if (y >= d.NY || x >= d.NX) return;                 // wave-uniform; no block-level barrier below
    GsCoef ck;
    gs_load_coef<RES>(tab, ck);
    gs_row_segment_mf0<RES>(d, tab, ck, E, u, b, mask, x, y, blockIdx.x * 64, cz, forward, rowbuf[threadIdx.y]);
}

// Both z colours of the rows (cx, cy) in one launch: a wave owns its row over the whole z extent and relaxes, segment by segment,
// first the nodes of colour c1 and then those of the other colour between them -- the same arithmetic as two launches of
// k_gs_rows_mf0 (bitwise the same result), but the nine node rows of a segment are fetched from HBM once for both colours (the
// second time they come from L2): the sweep is bound by the HBM traffic of its passes over u and E.  No other wave of the launch
// reads this row (rows of equal parity are two apart), and the order A(s+1) before B(s) (even colour first) resp. A(s), B(s)
// (odd colour first) keeps every first-colour update ahead of the second-colour updates that read it and behind none.
template <bool RES>
__global__ void __launch_bounds__(256, VFEM_GS_MINW) k_gs_rows_mf0_pair(Dims d, const double *__restrict__ tab, const double *__restrict__ E,
                                                          double *__restrict__ u, const double *__restrict__ b,
                                                          const uint8_t *__restrict__ mask, int cx, int cy, int c1, int forward,
                                                          int ystride) {
    __shared__ double rowbuf[4][GS_ROWBUF];
    const int x = 2 * blockIdx.z + cx;
    const int y = __builtin_amdgcn_readfirstlane(ystride * (blockIdx.y * 4 + threadIdx.y) + cy);       // ystride 2: every row of parity cy
    if (y >= d.NY || x >= d.NX) return;                 // wave-uniform; no block-level barrier below
    double *buf = rowbuf[threadIdx.y];
    GsCoef ck;
    gs_load_coef<RES>(tab, ck);
    const int c2 = 1 - c1;
    const int nA = ((d.NZ - 1 - c1) / 2 + 1 + 63) / 64, nB = d.NZ - 1 - c2 < 0 ? 0 : ((d.NZ - 1 - c2) / 2 + 1 + 63) / 64;
    const int lag = c1 == 0 ? 1 : 0;                    // even colour first: the last node of B(s) needs the first node of A(s+1)
    const int steps = nA > nB + lag ? nA : nB + lag;
    for (int s = 0; s < steps; ++s) {
        if (s < nA) gs_row_segment_mf0<RES>(d, tab, ck, E, u, b, mask, x, y, 64 * s, c1, forward, buf);
        if (s - lag >= 0 && s - lag < nB) {
            // the first-colour values this wave has just stored are read back below (same wave, in order; the fence makes the
            // stores complete before the loads are issued)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            gs_row_segment_mf0<RES>(d, tab, ck, E, u, b, mask, x, y, 64 * (s - lag), c2, forward, buf);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
}


// ------------------------------------------------------------------------------------------
// Level-1 colour sweep on the reflection symmetry of the coarsened reference matrices.
//
// cK0[f] = I_f^T K0 I_f (child f = 4fx+2fy+fz of a coarse element) is the mirror image of cK0[0]:
//     cK0[f][(n,a),(m,b)] = s_a(f) s_b(f) cK0[0][(n^f, a),(m^f, b)],   s_a(f) = -1 if f has the bit of axis a
// (box voxels, isotropic tensor; verified numerically at hierarchy creation).  The general kernel streams all eight
// matrices (36.9 KB) through the 16 KB scalar cache and sits at ~35 % of the fp64 rate waiting on scalar-cache
// misses.  Here only cK0[0] (4.6 KB, cache resident) is read: for a node with local index li in its element and child
// f the rows are the rows rho = li ^ f of cK0[0]; the column permutation m -> m ^ f and the signs are resolved at
// compile time (register renaming, per-component partial sums).  Register footprint as the general kernel.
// ------------------------------------------------------------------------------------------
// diagonal block of one incident element of a level-1 node: sum_f E_f s_r s_c cK0[0][(rho,r),(rho,c)], rho = li ^ f, from the
// 8 x 9 table of diagonal blocks (build_mf1_diag_table); li = local index of the node in the element
__device__ __forceinline__ void mf1_diag_slot(const double *__restrict__ Dtab, int li, const double Ef[8], double M[9]) {
    static_for<8>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        const int rho = li ^ f;
        d8_t k0;
        d4_t k1;
        sload12(Dtab, rho * 96, k0, k1);
        constexpr bool ng[3] = {(bool) ((f >> 2) & 1), (bool) ((f >> 1) & 1), (bool) (f & 1)};
        static_for<9>([&](auto qc) {
            constexpr int q = decltype(qc)::value, r = q / 3, c = q % 3;
            const double kv = q < 8 ? k0[q < 8 ? q : 0] : k1[0];
            M[q] = fma((ng[r] != ng[c]) ? -Ef[f] : Ef[f], kv, M[q]);
        });
        asm volatile("" : "+v"(M[0]), "+v"(M[4]), "+v"(M[8]), "+v"(M[1]), "+v"(M[2]), "+v"(M[5]));
    });
}

// The diagonal blocks of the level-1 operator depend on the moduli only: computed once per operator update (same sums, same
// order as inside the sweep: bitwise the same blocks) and read back by the sweeps -- 72 B per node visit instead of 576
// multiply-adds and 64 scalar loads (11 % of the sweep's arithmetic).
__global__ void __launch_bounds__(256) k_mf1_diag(Dims d, const double *__restrict__ Dtab, const double *__restrict__ E,
                                                  double *__restrict__ Mdiag) {
    const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
    if (k >= d.NZ || j >= d.NY) return;
    const long long nyf = 2LL * d.ny, nzf = 2LL * d.nz;
    double M[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) M[q] = 0.0;
#pragma unroll 1
    for (int slot = 0; slot < 8; ++slot) {
        const int li = 7 - slot;
        const int ex = i - 1 + ((slot >> 2) & 1), ey = j - 1 + ((slot >> 1) & 1), ez = k - 1 + (slot & 1);
        const bool ok = ex >= 0 && ex < d.nx && ey >= 0 && ey < d.ny && ez >= 0 && ez < d.nz;
        const int exc = min(max(ex, 0), d.nx - 1), eyc = min(max(ey, 0), d.ny - 1), ezc = min(max(ez, 0), d.nz - 1);
        double Ef[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const double v = E[((2LL * exc + ((f >> 2) & 1)) * nyf + (2LL * eyc + ((f >> 1) & 1))) * nzf + (2LL * ezc + (f & 1))];
            Ef[f] = ok ? v : 0.0;
        }
        mf1_diag_slot(Dtab, li, Ef, M);
    }
    const long long n = nidx(d, i, j, k);
#pragma unroll
    for (int q = 0; q < 9; ++q) Mdiag[9 * n + q] = M[q];
}

void launch_mf1_diag(const Dims &d, const double *Dtab, const double *E, double *Mdiag, hipStream_t s) {
    k_mf1_diag<<<dim3((d.NZ + 63) / 64, (d.NY + 3) / 4, d.NX), dim3(64, 4, 1), 0, s>>>(d, Dtab, E, Mdiag);
    VFEM_HIP(hipGetLastError());
}

// relaxation of one level-1 node (i, j, k) with the mirror-symmetric child matrices: the work of one lane
// S, M: the sums over the element slots [slot0, slot1) of the node (the whole node: 0, 8)
__device__ __forceinline__ void mf1_sym_partial(const Dims &d, const double *__restrict__ K0c, const double *__restrict__ Dtab,
                                                const double *__restrict__ Mdiag, const double *__restrict__ E, const double *__restrict__ u,
                                                int i, int j, int k, int slot0, int slot1, double S[3], double M[9]) {
    const long long nyf = 2LL * d.ny, nzf = 2LL * d.nz;
    const long long sx = (long long) d.NY * d.NZ, sy = d.NZ;
    S[0] = S[1] = S[2] = 0.0;
    if (Mdiag && slot0 == 0) {                     // precomputed diagonal block, requested ahead of the slot loop
        const long long nq = 9 * nidx(d, i, j, k);
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = Mdiag[nq + q];
    } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = 0.0;
    }
    // the element slot is a run-time loop: fully unrolled the kernel is ~60 KB of straight-line code, about the size of
    // the instruction cache two CUs share; the child index f stays compile-time (it permutes registers and fixes the signs)
#pragma unroll 1
    for (int slot = slot0; slot < slot1; ++slot) {
        const int li = 7 - slot;
        const int di = (slot >> 2) & 1, dj = (slot >> 1) & 1, dk = slot & 1;
        const int ex = i - 1 + di, ey = j - 1 + dj, ez = k - 1 + dk;
        const bool ok = ex >= 0 && ex < d.nx && ey >= 0 && ey < d.ny && ez >= 0 && ez < d.nz;
        // out-of-grid elements (grid faces only) run with clamped indices and zero moduli: no divergent control flow
        const int exc = min(max(ex, 0), d.nx - 1), eyc = min(max(ey, 0), d.ny - 1), ezc = min(max(ez, 0), d.nz - 1);
        const long long base = nidx(d, exc, eyc, ezc);
        double ue[24];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const long long nm = base + ((m >> 2) & 1) * sx + ((m >> 1) & 1) * sy + (m & 1);
            ue[3 * m + 0] = u[3 * nm + 0];
            ue[3 * m + 1] = u[3 * nm + 1];
            ue[3 * m + 2] = u[3 * nm + 2];
        }
        double Ef[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const double v = E[((2LL * exc + ((f >> 2) & 1)) * nyf + (2LL * eyc + ((f >> 1) & 1))) * nzf + (2LL * ezc + (f & 1))];
            Ef[f] = ok ? v : 0.0;
        }
        static_for<24>([&](auto gc) {
            constexpr int f = decltype(gc)::value / 3, r = decltype(gc)::value % 3;
            const int rho = li ^ f;                                   // wave-uniform, run-time
            d8_t c0, c1, c2;
            sload24(K0c, (3 * rho + r) * 24 * 8, c0, c1, c2);
            double coef[24];
            static_for<24>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                coef[q] = q < 8 ? c0[q < 8 ? q : 0] : (q < 16 ? c1[(q >= 8 && q < 16) ? q - 8 : 0] : c2[q >= 16 ? q - 16 : 0]);
            });
            double acc[3] = {0.0, 0.0, 0.0};
            static_for<24>([&](auto tc) {
                constexpr int mq = decltype(tc)::value / 3, c = decltype(tc)::value % 3;
                acc[c] = fma(coef[3 * mq + c], ue[3 * (mq ^ f) + c], acc[c]);
            });
            constexpr bool n0 = (f >> 2) & 1, n1 = (f >> 1) & 1, n2 = f & 1;      // sign flips of the x, y, z components
            constexpr bool nr = (f >> (2 - r)) & 1;
            // s_r * sum_c s_c acc_c
            const double t = ((n0 != nr) ? -acc[0] : acc[0]) + ((n1 != nr) ? -acc[1] : acc[1]) + ((n2 != nr) ? -acc[2] : acc[2]);
            S[r] = fma(Ef[f], t, S[r]);
            asm volatile("" : "+v"(S[r]));                            // retire before the next row load
        });
        if (!Mdiag) mf1_diag_slot(Dtab, li, Ef, M);
    }
}
__device__ __forceinline__ void mf1_relax(const Dims &d, double *__restrict__ u, const double *__restrict__ b, const uint8_t *__restrict__ mask,
                                          int i, int j, int k, int forward, const double S[3], const double M[9]) {
    const long long n = nidx(d, i, j, k);
    double bms[3], ud[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bms[c] = b[3 * n + c] - S[c];
    gs_solve(bms, M, mask[n], forward != 0, ud);
#pragma unroll
    for (int c = 0; c < 3; ++c) u[3 * n + c] += ud[c];
}
// relaxation of one level-1 node (i, j, k) with the mirror-symmetric child matrices: the work of one lane
__device__ __forceinline__ void gs_node_mf1_sym(const Dims &d, const double *__restrict__ K0c, const double *__restrict__ Dtab,
                                                const double *__restrict__ Mdiag, const double *__restrict__ E, double *__restrict__ u,
                                                const double *__restrict__ b, const uint8_t *__restrict__ mask, int i, int j, int k,
                                                int forward) {
    if (k >= d.NZ) return;
    double S[3], M[9];
    mf1_sym_partial(d, K0c, Dtab, Mdiag, E, u, i, j, k, 0, 8, S, M);
    mf1_relax(d, u, b, mask, i, j, k, forward, S, M);
}

__global__ void __launch_bounds__(256, 4) k_gs_color_mf1_sym(Dims d, const double *__restrict__ K0c, const double *__restrict__ Dtab,
                                                             const double *__restrict__ Mdiag, const double *__restrict__ E, double *__restrict__ u,
                                                             const double *__restrict__ b, const uint8_t *__restrict__ mask, int cx,
                                                             int cy, int cz, int forward) {
    // lanes are packed over the colour's nodes of an x-plane (row after row): a row of 2^k + 1 nodes has one even-colour node
    // more than a whole number of waves, and a wave per row segment left every third (second) wave of such a row with a single
    // useful lane doing the full 5184 multiply-adds
    const int cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
    const int q = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x;
    if (q >= cnty * cntz) return;
    const int jq = q / cntz;
    const int k = 2 * (q - jq * cntz) + cz, j = 2 * jq + cy, i = 2 * blockIdx.z + cx;
    gs_node_mf1_sym(d, K0c, Dtab, Mdiag, E, u, b, mask, i, j, k, forward);
}

// The same sweep with the eight element slots of a node shared by TWO waves (slots 0-3 / 4-7; partial sums meet in LDS, added in
// slot order).  A launch whose waves just exceed the chip's resident-wave slots pays for a second, almost empty round with a full
// wave duration (a colour of the 129^3 level: 4291 waves for 4096 slots, 128 us per launch against 94 us by the per-node rate
// of the 257^3 level); with half the work per wave the rounds are half as long.  Used while a colour has fewer than three
// rounds of waves.
template <int SPLIT>      // waves per node group: 2, 4 or 8 (8 / SPLIT element slots per wave)
__global__ void __launch_bounds__(SPLIT == 8 ? 512 : 256) k_gs_color_mf1_sym_split(Dims d, const double *__restrict__ K0c, const double *__restrict__ Dtab,
                                                                   const double *__restrict__ Mdiag, const double *__restrict__ E,
                                                                   double *__restrict__ u, const double *__restrict__ b,
                                                                   const uint8_t *__restrict__ mask, int cx, int cy, int cz, int forward) {
    constexpr int GROUPS = SPLIT == 8 ? 1 : 256 / (64 * SPLIT);
    __shared__ double part[GROUPS][SPLIT - 1][12][64];
    const int lane = threadIdx.x, part_id = __builtin_amdgcn_readfirstlane(threadIdx.y), grp = threadIdx.z;       // (part_id: uniform over the wave)
    const int cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
    const int q = blockIdx.x * (64 * GROUPS) + grp * 64 + lane;
    const bool live = q < cnty * cntz;
    const int qq = live ? q : cnty * cntz - 1;
    const int jq = qq / cntz;
    const int k = 2 * (qq - jq * cntz) + cz, j = 2 * jq + cy, i = 2 * blockIdx.z + cx;
    double S[3], M[9];
    mf1_sym_partial(d, K0c, Dtab, Mdiag, E, u, i, j, k, (8 / SPLIT) * part_id, (8 / SPLIT) * (part_id + 1), S, M);
    if (part_id > 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) part[grp][part_id - 1][c][lane] = S[c];
#pragma unroll
        for (int c = 0; c < 9; ++c) part[grp][part_id - 1][3 + c][lane] = M[c];
    }
    __syncthreads();
    if (part_id > 0 || !live) return;
    for (int o = 0; o < SPLIT - 1; ++o) {
#pragma unroll
        for (int c = 0; c < 3; ++c) S[c] += part[grp][o][c][lane];
#pragma unroll
        for (int c = 0; c < 9; ++c) M[c] += part[grp][o][3 + c][lane];
    }
    mf1_relax(d, u, b, mask, i, j, k, forward, S, M);
}

// diagonal blocks of cK0[0] for k_gs_color_mf1_sym: 8 groups of 12 doubles (9 used)
void build_mf1_diag_table(const double *cK0_0, double *tab /* 8*12 */) {
    for (int rho = 0; rho < 8; ++rho) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) tab[rho * 12 + 3 * r + c] = cK0_0[(3 * rho + r) * 24 + 3 * rho + c];
        for (int q = 9; q < 12; ++q) tab[rho * 12 + q] = 0.0;
    }
}

// true when cK0[f] is the mirror image of cK0[0] to rounding (what k_gs_color_mf1_sym relies on)
bool coarsened_matrices_are_mirror_images(const double *cK0 /* 8 x 576, host */) {
    double scale = 0.0, err = 0.0;
    for (int q = 0; q < 576; ++q) scale = std::max(scale, std::fabs(cK0[q]));
    for (int f = 1; f < 8; ++f)
        for (int n = 0; n < 8; ++n)
            for (int a = 0; a < 3; ++a)
                for (int m = 0; m < 8; ++m)
                    for (int bb = 0; bb < 3; ++bb) {
                        const double sg = (((f >> (2 - a)) & 1) ^ ((f >> (2 - bb)) & 1)) ? -1.0 : 1.0;
                        const double want = sg * cK0[(3 * (n ^ f) + a) * 24 + 3 * (m ^ f) + bb];
                        err = std::max(err, std::fabs(cK0[f * 576 + (3 * n + a) * 24 + 3 * m + bb] - want));
                    }
    return err <= 1e-13 * scale;
}

// mf1_sym: coarsened_matrices_are_mirror_images() holds for the hierarchy (level-1 sweeps read cK0[0] only)
void launch_gs_sweep_mf(const Dims &d, OpKind kind, const double *K, const double *gs_tab, const double *E, double *u,
                        const double *b, const uint8_t *mask, int forward, int xparity, int first, int count, hipStream_t s,
                        const Tuning &tune, bool mf1_sym, const double *mdiag) {
    const int g_gs_variant = tune.gs_variant, g_gs_pair = tune.gs_pair;
    const bool g_mf1_sym = mf1_sym;
    const bool res = tune.gs_resident != 0;       // level 0: the 36 resident coefficients follow the 72 x 12 table in gs_tab
    for (int ci = first; ci < first + count; ++ci) {
        const int lni = forward ? ci : 7 - ci;
        const int cx = ((lni >> 2) & 1) ^ (xparity & 1), cy = (lni >> 1) & 1, cz = lni & 1;   // global -> local x parity
        if (cx > d.NX - 1) continue;
        const int cntx = (d.NX - 1 - cx) / 2 + 1, cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
        dim3 blk(64, 4, 1), grd((cntz + 63) / 64, (cnty + 3) / 4, cntx);
        if (kind == OP_MF0 && g_gs_variant == 0 && gs_tab && g_gs_pair && ((ci - first) % 2 == 0) && ci % 2 == 0 && ci + 1 < first + count && d.NZ >= 3) {
            // colours 2m and 2m+1 of the sweep order differ in cz only: one launch, the wave walks its row through both
            if (res) k_gs_rows_mf0_pair<true><<<dim3(1, (cnty + 3) / 4, cntx), blk, 0, s>>>(d, gs_tab + GS_TABLE_DOUBLES, E, u, b, mask, cx, cy, cz, forward, 2);
            else     k_gs_rows_mf0_pair<false><<<dim3(1, (cnty + 3) / 4, cntx), blk, 0, s>>>(d, gs_tab, E, u, b, mask, cx, cy, cz, forward, 2);
            ++ci;
            continue;
        }
        if (kind == OP_MF0 && g_gs_variant == 0 && gs_tab && res && d.nz >= 2) k_gs_rows_mf0<true><<<grd, blk, 0, s>>>(d, gs_tab + GS_TABLE_DOUBLES, E, u, b, mask, cx, cy, cz, forward);
        else if (kind == OP_MF0 && g_gs_variant == 0 && gs_tab && d.nz >= 2) k_gs_rows_mf0<false><<<grd, blk, 0, s>>>(d, gs_tab, E, u, b, mask, cx, cy, cz, forward);
        else if (kind == OP_MF0) k_gs_color_mf<0><<<grd, blk, 0, s>>>(d, K, E, u, b, mask, cx, cy, cz, forward);
        else if (g_mf1_sym && g_gs_variant == 0 && gs_tab && tune.l1_split == 2)
            k_gs_color_mf1_sym_split<2><<<dim3((cnty * cntz + 127) / 128, 1, cntx), dim3(64, 2, 2), 0, s>>>(d, K, gs_tab, mdiag, E, u, b, mask, cx, cy, cz, forward);
        else if (g_mf1_sym && g_gs_variant == 0 && gs_tab && tune.l1_split == 4)
            k_gs_color_mf1_sym_split<4><<<dim3((cnty * cntz + 63) / 64, 1, cntx), dim3(64, 4, 1), 0, s>>>(d, K, gs_tab, mdiag, E, u, b, mask, cx, cy, cz, forward);
        else if (g_mf1_sym && g_gs_variant == 0 && gs_tab && tune.l1_split == 8)
            k_gs_color_mf1_sym_split<8><<<dim3((cnty * cntz + 63) / 64, 1, cntx), dim3(64, 8, 1), 0, s>>>(d, K, gs_tab, mdiag, E, u, b, mask, cx, cy, cz, forward);
        else if (g_mf1_sym && g_gs_variant == 0 && gs_tab)
            k_gs_color_mf1_sym<<<dim3((cnty * cntz + 255) / 256, 1, cntx), blk, 0, s>>>(d, K, gs_tab, mdiag, E, u, b, mask, cx, cy, cz, forward);
        else                k_gs_color_mf<1><<<grd, blk, 0, s>>>(d, K, E, u, b, mask, cx, cy, cz, forward);
    }
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// stored 27-point stencil levels (layout: cm_index below).
// Replaces the block-CSC matrix of TPS::updateBlockK (TPS.hh:649-720): on a regular grid the
// column indices are implicit.
// ------------------------------------------------------------------------------------------
// Colour-major, tile-major stencil storage: the nodes of one Gauss-Seidel colour (parity class) are numbered row-major
// within the colour and cut into tiles of 64 consecutive nodes; a tile holds its 243 entries (27 neighbours x 3 x 3) as 243
// runs of 64 doubles, 124 KB contiguous.  Entry e of the node with colour-local index q lives at
//   St[ 243 * (padded nodes of earlier colours) + (q / 64) * 243 * 64 + e * 64 + (q % 64) ].
// A colour sweep (lanes = every other node in z = consecutive q) reads full cache lines AND each wave's 243 loads fall into
// one or two contiguous 124 KB blocks; with one array per entry across the whole colour (the first layout) a wave's loads
// were 243 separate 512-byte pieces megabytes apart, and the sweeps ran at 3.5 TB/s.
__host__ __device__ inline long long cm_padded(long long cnt) { return (cnt + 63) / 64 * 64; }
__device__ __forceinline__ void cm_index(const Dims &d, int i, int j, int k, long long &base, long long &stride) {
    const int ci = i & 1, cj = j & 1, ck = k & 1;
    const long long nx[2] = {(d.NX + 1) >> 1, d.NX >> 1}, ny[2] = {(d.NY + 1) >> 1, d.NY >> 1}, nz[2] = {(d.NZ + 1) >> 1, d.NZ >> 1};
    long long before = 0;
    const int c = ci * 4 + cj * 2 + ck;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < c) before += cm_padded(nx[(q >> 2) & 1] * ny[(q >> 1) & 1] * nz[q & 1]);
    const long long q = ((long long) (i >> 1) * ny[cj] + (j >> 1)) * nz[ck] + (k >> 1);
    base = 243 * before + (q >> 6) * (243 * 64) + (q & 63);
    stride = 64;
}
// padded number of nodes of the colours before (ci, cj, ck): cm_index's `before`
__device__ __forceinline__ long long cm_colour_start(const Dims &d, int ci, int cj, int ck) {
    const long long nx[2] = {(d.NX + 1) >> 1, d.NX >> 1}, ny[2] = {(d.NY + 1) >> 1, d.NY >> 1}, nz[2] = {(d.NZ + 1) >> 1, d.NZ >> 1};
    long long before = 0;
    const int c = ci * 4 + cj * 2 + ck;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < c) before += cm_padded(nx[(q >> 2) & 1] * ny[(q >> 1) & 1] * nz[q & 1]);
    return before;
}
long long stencil_storage_doubles(const Dims &d) {
    const long long nx[2] = {(d.NX + 1) >> 1, d.NX >> 1}, ny[2] = {(d.NY + 1) >> 1, d.NY >> 1}, nz[2] = {(d.NZ + 1) >> 1, d.NZ >> 1};
    long long total = 0;
    for (int q = 0; q < 8; ++q) total += cm_padded(nx[(q >> 2) & 1] * ny[(q >> 1) & 1] * nz[q & 1]);
    return 243 * total;
}

template <bool WITH_M>
__device__ __forceinline__ void stencil_node(const Dims &d, const double *__restrict__ St, const double *__restrict__ u,
                                             int i, int j, int k, long long n, double S[3], double M[9], double *uself = nullptr) {
    S[0] = S[1] = S[2] = 0.0;
    long long sbase, scnt;
    cm_index(d, i, j, k, sbase, scnt);
    (void) n;
    if (!WITH_M) {                 // apply / residual (every node, all colours in a wave): the rolled loop keeps the register count low
        for (int nb = 0; nb < 27; ++nb) {
            const int di = nb / 9 - 1, dj = (nb / 3) % 3 - 1, dk = nb % 3 - 1;
            const int ii = i + di, jj = j + dj, kk = k + dk;
            if (ii < 0 || ii >= d.NX || jj < 0 || jj >= d.NY || kk < 0 || kk >= d.NZ) continue;
            const long long m = nidx(d, ii, jj, kk);
            const double u0 = u[3 * m], u1 = u[3 * m + 1], u2 = u[3 * m + 2];
            const double *a = St + sbase + (long long) nb * 9 * scnt;
            double A[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) A[q] = a[(long long) q * scnt];
#pragma unroll
            for (int r = 0; r < 3; ++r) S[r] += A[3 * r] * u0 + A[3 * r + 1] * u1 + A[3 * r + 2] * u2;
        }
        return;
    }
    // Gauss-Seidel (one colour per launch), branch-free: the stencil entries of neighbours outside the grid are stored as
    // zeros (k_stencil_build), so such a neighbour is read at the clamped (existing) node and contributes exactly 0; without
    // branches the loads of the 27 blocks can be issued ahead of the arithmetic (6 % faster; the same form made the
    // all-node apply 4x slower)
    static_for<27>([&](auto nbc) {
        constexpr int nb = decltype(nbc)::value, di = nb / 9 - 1, dj = (nb / 3) % 3 - 1, dk = nb % 3 - 1;
        int ii = i + di, jj = j + dj, kk = k + dk;
        if (di < 0) ii = ii < 0 ? 0 : ii;
        if (di > 0) ii = ii > d.NX - 1 ? d.NX - 1 : ii;
        if (dj < 0) jj = jj < 0 ? 0 : jj;
        if (dj > 0) jj = jj > d.NY - 1 ? d.NY - 1 : jj;
        if (dk < 0) kk = kk < 0 ? 0 : kk;
        if (dk > 0) kk = kk > d.NZ - 1 ? d.NZ - 1 : kk;
        const long long m = nidx(d, ii, jj, kk);
        const double u0 = u[3 * m], u1 = u[3 * m + 1], u2 = u[3 * m + 2];
        const double *a = St + sbase + (long long) nb * 9 * scnt;
        double A[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) A[q] = a[(long long) q * scnt];
#pragma unroll
        for (int r = 0; r < 3; ++r) S[r] += A[3 * r] * u0 + A[3 * r + 1] * u1 + A[3 * r + 2] * u2;
        if (WITH_M && nb == 13) {
#pragma unroll
            for (int q = 0; q < 9; ++q) M[q] = A[q];
            if (uself) { uself[0] = u0; uself[1] = u1; uself[2] = u2; }      // the centre of the stencil is the node itself
        }
    });
}

template <bool RES>
__global__ void __launch_bounds__(256) k_apply_stencil(Dims d, const double *__restrict__ St, const double *__restrict__ u,
                                                       const double *__restrict__ b, const uint8_t *__restrict__ mask,
                                                       double *__restrict__ out) {
    const int q = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x;       // lanes packed over the nodes of an x-plane
    if (q >= d.NY * d.NZ) return;
    const int j = q / d.NZ, k = q - j * d.NZ, i = blockIdx.z;
    const long long n = nidx(d, i, j, k);
    double S[3], M[9];
    stencil_node<false>(d, St, u, i, j, k, n, S, M);
    if (RES) {
        const uint8_t m = mask ? mask[n] : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * n + c] = ((m >> c) & 1) ? 0.0 : b[3 * n + c] - S[c];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * n + c] = S[c];
    }
}

void launch_apply_stencil(const Dims &d, const double *S, const double *u, const double *b, const uint8_t *mask,
                          int res, double *out, hipStream_t s) {
    dim3 blk(64, 4, 1), grd((d.NY * d.NZ + 255) / 256, 1, d.NX);
    if (res) k_apply_stencil<true><<<grd, blk, 0, s>>>(d, S, u, b, mask, out);
    else     k_apply_stencil<false><<<grd, blk, 0, s>>>(d, S, u, b, mask, out);
    VFEM_HIP(hipGetLastError());
}

__global__ void __launch_bounds__(256) k_gs_color_stencil(Dims d, const double *__restrict__ St, double *__restrict__ u,
                                                          const double *__restrict__ b, const uint8_t *__restrict__ mask,
                                                          int cx, int cy, int cz, int forward) {
    // lanes follow the colour-local node index of the stencil storage (cm_index): a wave is one 64-node tile, whole waves
    // whatever the row length
    const int cntx = (d.NX - 1 - cx) / 2 + 1, cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
    const long long q = (long long) blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x;
    if (q >= (long long) cntx * cnty * cntz) return;
    const int iq = (int) (q / ((long long) cnty * cntz)), rem = (int) (q - (long long) iq * cnty * cntz), jq = rem / cntz;
    const int i = 2 * iq + cx, j = 2 * jq + cy, k = 2 * (rem - jq * cntz) + cz;
    const long long n = nidx(d, i, j, k);
    // right-hand side and mask requested ahead of the stencil loads, the node's own value taken from the stencil centre: on the
    // small levels a launch lasts little more than its chain of dependent round trips
    double bv[3], uself[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bv[c] = b[3 * n + c];
    const uint8_t mk = mask[n];
    double S[3], M[9];
    stencil_node<true>(d, St, u, i, j, k, n, S, M, uself);
    double bms[3], ud[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bms[c] = bv[c] - S[c];
    gs_solve(bms, M, mk, forward != 0, ud);
#pragma unroll
    for (int c = 0; c < 3; ++c) u[3 * n + c] = uself[c] + ud[c];
}

// The node-per-lane relaxation with the 27 neighbour blocks shared by THREE waves (one x-plane of neighbours each; partial
// row sums through LDS, the wave of the centre plane holds the diagonal block and finishes): a third of the dependent loads per
// lane and three times the waves -- the mid-size levels (65^3: 537 waves per colour) are latency-bound, the large one
// (129^3) gets more loads in flight against its stencil traffic.
template <int W>
__device__ __forceinline__ void stencil_plane(const Dims &d, const double *__restrict__ St, const double *__restrict__ u, int i, int j, int k,
                                              long long sbase, long long scnt, double S[3], double M[9], double uself[3]) {
    static_for<9>([&](auto tc) {
        constexpr int nb = 9 * W + decltype(tc)::value, di = W - 1, dj = (nb / 3) % 3 - 1, dk = nb % 3 - 1;
        int ii = i + di, jj = j + dj, kk = k + dk;
        if (di < 0) ii = ii < 0 ? 0 : ii;
        if (di > 0) ii = ii > d.NX - 1 ? d.NX - 1 : ii;
        if (dj < 0) jj = jj < 0 ? 0 : jj;
        if (dj > 0) jj = jj > d.NY - 1 ? d.NY - 1 : jj;
        if (dk < 0) kk = kk < 0 ? 0 : kk;
        if (dk > 0) kk = kk > d.NZ - 1 ? d.NZ - 1 : kk;
        const long long m = nidx(d, ii, jj, kk);
        const double u0 = u[3 * m], u1 = u[3 * m + 1], u2 = u[3 * m + 2];
        const double *a = St + sbase + (long long) nb * 9 * scnt;
        double A[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) A[q] = a[(long long) q * scnt];
#pragma unroll
        for (int r = 0; r < 3; ++r) S[r] += A[3 * r] * u0 + A[3 * r + 1] * u1 + A[3 * r + 2] * u2;
        if (nb == 13) {
#pragma unroll
            for (int q = 0; q < 9; ++q) M[q] = A[q];
            uself[0] = u0; uself[1] = u1; uself[2] = u2;
        }
    });
}
__global__ void __launch_bounds__(192) k_gs_color_stencil_split(Dims d, const double *__restrict__ St, double *__restrict__ u,
                                                                const double *__restrict__ b, const uint8_t *__restrict__ mask,
                                                                int cx, int cy, int cz, int forward) {
    __shared__ double part[2][3][64];
    const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int cntx = (d.NX - 1 - cx) / 2 + 1, cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
    // (32-bit index arithmetic: the launcher checks that a colour has fewer than 2^31 nodes; the 64-bit divisions were ~200 of the
    // kernel's ~600 instructions per lane)
    const unsigned q0 = blockIdx.x * 64u + (unsigned) lane, total = (unsigned) cntx * (unsigned) cnty * (unsigned) cntz;
    const bool live = q0 < total;
    const unsigned q = live ? q0 : total - 1u, pc = (unsigned) cnty * (unsigned) cntz;
    const int iq = (int) (q / pc), rem = (int) (q - (unsigned) iq * pc), jq = rem / cntz;
    const int i = 2 * iq + cx, j = 2 * jq + cy, k = 2 * (rem - jq * cntz) + cz;
    const long long n = nidx(d, i, j, k);
    long long sbase, scnt;
    cm_index(d, i, j, k, sbase, scnt);
    double S[3] = {0.0, 0.0, 0.0}, M[9], uself[3] = {0.0, 0.0, 0.0};
    if (w == 0) stencil_plane<0>(d, St, u, i, j, k, sbase, scnt, S, M, uself);
    else if (w == 1) stencil_plane<1>(d, St, u, i, j, k, sbase, scnt, S, M, uself);
    else stencil_plane<2>(d, St, u, i, j, k, sbase, scnt, S, M, uself);
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

// The same relaxation with ONE WAVE PER NODE, for the small levels (a colour of a 33^3 level has 4.5 k nodes, of a 17^3 level 614):
// there a launch of the node-per-lane kernel lasts ~10 us whatever its size -- the chain of a node's 243 + 81 loads -- and a
// sweep is eight such launches.  Here the 243 stencil entries of the node are spread over the lanes (four each, all loads of
// the node in flight at once), every lane multiplies its entries with the matching neighbour component, a fixed xor tree adds
// the three row sums, the centre block comes from the lanes that hold it.  Same arithmetic in a different (fixed) order.
// node-major copy of a level's stencil for the wave-per-node sweep: the 243 entries of the node with padded colour-major number
// lin (all 64-node tiles of the earlier colours, then the node's place in its own) are contiguous, Sn[243 lin + e]
__global__ void __launch_bounds__(256) k_stencil_node_major(long long ntiles, const double *__restrict__ St, double *__restrict__ Sn) {
    const long long t = (long long) blockIdx.x * 256 + threadIdx.x;              // (tile, entry, lane) -> lane fastest: coalesced reads
    if (t >= ntiles * 243 * 64) return;
    const long long tile = t / (243 * 64);
    const int e = (int) ((t / 64) % 243), lane = (int) (t % 64);
    Sn[(tile * 64 + lane) * 243 + e] = St[t];
}
void launch_stencil_node_major(const Dims &d, const double *St, double *Sn, hipStream_t s) {
    const long long ntiles = stencil_storage_doubles(d) / (243 * 64);
    k_stencil_node_major<<<dim3((unsigned) ((ntiles * 243 * 64 + 255) / 256)), dim3(256), 0, s>>>(ntiles, St, Sn);
    VFEM_HIP(hipGetLastError());
}

__global__ void __launch_bounds__(256) k_gs_color_stencil_wave(Dims d, const double *__restrict__ Sn, double *__restrict__ u,
                                                               const double *__restrict__ b, const uint8_t *__restrict__ mask,
                                                               int cx, int cy, int cz, int forward) {
    // the grid is (groups of four nodes along z, rows, planes) of the colour: no division on the way to the node -- a launch of
    // this kernel is one latency chain, and the 64-bit divisions of a flat node number were a microsecond of it
    const int cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
    const int kq = blockIdx.x * 4 + threadIdx.y, jq = blockIdx.y, iq = blockIdx.z;
    if (kq >= cntz) return;
    const int lane = threadIdx.x;
    const int i = 2 * iq + cx, j = 2 * jq + cy, k = 2 * kq + cz;
    const long long n = nidx(d, i, j, k);
    // padded colour-major number of the node (cm_index: the 64-node tiles of the earlier colours, then its place in its own)
    const double *row = Sn + (cm_colour_start(d, cx, cy, cz) + ((long long) iq * cnty + jq) * cntz + kq) * 243;
    // what the solve of lane 0 needs is requested with the row (wave-uniform addresses): a launch of this kernel is one latency
    // chain, and loads issued after the reduction would add a second memory round trip to it
    const double bn0 = b[3 * n], bn1 = b[3 * n + 1], bn2 = b[3 * n + 2], un0 = u[3 * n], un1 = u[3 * n + 1], un2 = u[3 * n + 2];
    const uint8_t fixed = mask[n];
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, centre = 0.0;
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        const int e = lane + 64 * sidx;
        if (e < 243) {
            const int nb = e / 9, qq = e - 9 * nb, r = qq / 3, c = qq - 3 * r;
            // neighbours outside the grid have zero entries (k_stencil_build): read the clamped node
            const int ii = min(max(i + nb / 9 - 1, 0), d.NX - 1), jj = min(max(j + (nb / 3) % 3 - 1, 0), d.NY - 1), kk = min(max(k + nb % 3 - 1, 0), d.NZ - 1);
            const double a = row[e];
            const double t = a * u[3 * nidx(d, ii, jj, kk) + c];
            p0 += r == 0 ? t : 0.0; p1 += r == 1 ? t : 0.0; p2 += r == 2 ? t : 0.0;
            if (sidx == 1) centre = a;                                           // entries 117..125 (the node's own block): lanes 53..61
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { p0 += __shfl_xor(p0, o); p1 += __shfl_xor(p1, o); p2 += __shfl_xor(p2, o); }
    double M[9];
#pragma unroll
    for (int qq = 0; qq < 9; ++qq) M[qq] = __shfl(centre, 53 + qq);
    if (lane != 0) return;
    double bms[3] = {bn0 - p0, bn1 - p1, bn2 - p2}, ud[3];
    gs_solve(bms, M, fixed, forward != 0, ud);
    u[3 * n] = un0 + ud[0]; u[3 * n + 1] = un1 + ud[1]; u[3 * n + 2] = un2 + ud[2];
}

void launch_gs_sweep_stencil(const Dims &d, const double *S, double *u, const double *b, const uint8_t *mask,
                             int forward, int xparity, int first, int count, hipStream_t s, const double *Sn, int stencil_split) {
    for (int ci = first; ci < first + count; ++ci) {
        const int lni = forward ? ci : 7 - ci;
        const int cx = ((lni >> 2) & 1) ^ (xparity & 1), cy = (lni >> 1) & 1, cz = lni & 1;
        if (cx > d.NX - 1) continue;
        const int cntx = (d.NX - 1 - cx) / 2 + 1, cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
        const long long cnt = (long long) cntx * cnty * cntz;
        dim3 blk(64, 4, 1), grd((unsigned) ((cnt + 255) / 256), 1, 1);
        if (Sn) k_gs_color_stencil_wave<<<dim3((unsigned) ((cntz + 3) / 4), (unsigned) cnty, (unsigned) cntx), blk, 0, s>>>(d, Sn, u, b, mask, cx, cy, cz, forward);
        else if (stencil_split && cnt < (1LL << 31)) k_gs_color_stencil_split<<<dim3((unsigned) ((cnt + 63) / 64)), dim3(64, 3, 1), 0, s>>>(d, S, u, b, mask, cx, cy, cz, forward);
        else k_gs_color_stencil<<<grd, blk, 0, s>>>(d, S, u, b, mask, cx, cy, cz, forward);
    }
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// grid transfer (trilinear weights 1, 1/2, 1/4, 1/8; no 1/2^N scaling), gather form
// ------------------------------------------------------------------------------------------
// Lanes run along z.  Of the three fine nodes 2k-1, 2k, 2k+1 a coarse node reads in each of its nine fine rows, lane k loads 2k and
// 2k+1 (48 contiguous bytes: a wave reads 3 KB of the row in one piece) and receives 2k-1 from lane k-1, which loaded it as its own
// second node (DPP wave shift; the first lane of a wave loads it itself).  Loading all three per lane -- 54 load instructions per
// coarse node for 24 bytes of result -- ran at the rate of its memory instructions, 2.7 TB/s at 512^3.  The sum runs over the
// rows and, inside a row, over 2k-1, 2k, 2k+1 as before: same values bit for bit.
__global__ void __launch_bounds__(256) k_restrict(Dims c, int FX, int shift, const double *__restrict__ fine,
                                                  double *__restrict__ coarse, double *__restrict__ zeroed) {
    // a wave = 64 coarse nodes along z of the rows (i, j) and (i + 1, j), i = 2 blockIdx.z: the fine plane between the two coarse
    // planes is loaded once for both (the x-neighbours of a launch lie a plane apart: what one wave does not share comes back
    // through the fabric, profiles/r03_transfers_pmc.json)
    const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, i = 2 * blockIdx.z;
    if (j >= c.NY) return;                                                 // (wave-uniform)
    const int FY = 2 * c.ny + 1, FZ = 2 * c.nz + 1;
    const bool live = k < c.NZ, lo = live && k > 0, hi = live && 2 * k + 1 < FZ, first = threadIdx.x == 0;
    const bool second = i + 1 < c.NX;                                       // (wave-uniform)
    double a[3] = {0.0, 0.0, 0.0}, a2[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int pl = 0; pl < 5; ++pl) {                                        // fine planes 2 i + shift - 1 ... + 3
        const int fi = 2 * i + shift - 1 + pl;
        if (fi < 0 || fi >= FX || (pl > 2 && !second)) continue;            // (wave-uniform)
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
            const int fj = 2 * j + dj;
            if (fj < 0 || fj >= FY) continue;                               // (wave-uniform)
            const double *p = fine + 3 * (((long long) fi * FY + fj) * FZ + (live ? 2 * k : 0));
            double f0[3], f1[3], fm[3];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                f0[cc] = p[cc];
                f1[cc] = hi ? p[3 + cc] : 0.0;
            }
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                fm[cc] = lane_below(f1[cc]);
                if (first && lo) fm[cc] = p[cc - 3];
            }
            if (pl <= 2) {                                                  // di = pl - 1 for coarse plane i
                const double w = (pl != 1 ? 0.5 : 1.0) * (dj ? 0.5 : 1.0);
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    if (lo) a[cc] = fma(0.5 * w, fm[cc], a[cc]);
                    a[cc] = fma(w, f0[cc], a[cc]);
                    if (hi) a[cc] = fma(0.5 * w, f1[cc], a[cc]);
                }
            }
            if (pl >= 2 && second) {                                        // di = pl - 3 for coarse plane i + 1
                const double w = (pl != 3 ? 0.5 : 1.0) * (dj ? 0.5 : 1.0);
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    if (lo) a2[cc] = fma(0.5 * w, fm[cc], a2[cc]);
                    a2[cc] = fma(w, f0[cc], a2[cc]);
                    if (hi) a2[cc] = fma(0.5 * w, f1[cc], a2[cc]);
                }
            }
        }
    }
    if (!live) return;
    const long long n = nidx(c, i, j, k);
    coarse[3 * n] = a[0]; coarse[3 * n + 1] = a[1]; coarse[3 * n + 2] = a[2];
    if (zeroed) { zeroed[3 * n] = 0.0; zeroed[3 * n + 1] = 0.0; zeroed[3 * n + 2] = 0.0; }      // the coarse initial guess of the V-cycle
    if (second) {
        const long long n2 = nidx(c, i + 1, j, k);
        coarse[3 * n2] = a2[0]; coarse[3 * n2 + 1] = a2[1]; coarse[3 * n2 + 2] = a2[2];
        if (zeroed) { zeroed[3 * n2] = 0.0; zeroed[3 * n2 + 1] = 0.0; zeroed[3 * n2 + 2] = 0.0; }
    }
}

void launch_restrict(const Dims &c, int fineNX, int shift, const double *fine, double *coarse, hipStream_t s, double *zeroed) {
    dim3 blk(64, 4, 1), grd((c.NZ + 63) / 64, (c.NY + 3) / 4, (c.NX + 1) / 2);
    k_restrict<<<grd, blk, 0, s>>>(c, fineNX, shift, fine, coarse, zeroed);
    VFEM_HIP(hipGetLastError());
}

// Interpolation (MG.hh:116-142) row by row with whole-line stores.  fixed (optional, !ACC): Dirichlet mask of the fine level, the
// interpolated field gets zeros at its components (the residual system's Dirichlet values, MG.hh:521-523, without a pass of their
// own).  A wave owns 128 consecutive nodes of a fine row (i, j): lane l loads
// coarse node k0 = 64 chunk + l of the (up to) four coarse rows the fine row depends on (24 bytes each; k0 + 1 comes from the next
// lane by a DPP wave shift), forms fine nodes 2 k0 and 2 k0 + 1, and the 3 KB of results go through LDS so that every store
// instruction writes 1 KB of consecutive bytes.  The node-per-lane kernel of rounds 1-2 issued 16 loads and 2 strided stores per
// fine node (2.8 TB/s of useful bytes at 512^3, bound by the rate of its memory instructions); this one 6 loads and 3 stores per
// 128.  Every fine node is summed in that kernel's order (coarse neighbours lexicographic in (x, y, z)): same values bit for bit.
template <bool ACC>
__global__ void __launch_bounds__(256) k_prolong_rows(Dims c, int shift, const double *__restrict__ coarse, double *__restrict__ fine,
                                                      const uint8_t *__restrict__ fixed) {
    __shared__ __align__(16) double sh[4][384];
    const int FY = 2 * c.ny + 1, FZ = 2 * c.nz + 1;
    const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int j = blockIdx.y * 4 + wv, i = blockIdx.z, chunk = blockIdx.x;
    if (j >= FY) return;                                                    // (wave-uniform)
    const int ig = i - shift;                                               // >= 0: shift is 0 or -1
    const int i0 = ig >> 1, j0 = j >> 1, oi = ig & 1, oj = j & 1;
    const int k0 = 64 * chunk + lane;
    const bool live = k0 < c.NZ, up = k0 + 1 < c.NZ;
    const double w0 = (oi ? 0.5 : 1.0) * (oj ? 0.5 : 1.0), w1 = 0.5 * w0;
    double v0[3] = {0.0, 0.0, 0.0}, v1[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int qi = t >> 1, qj = t & 1;
        if ((qi && !oi) || (qj && !oj) || i0 + qi > c.NX - 1) continue;     // (wave-uniform; beyond the local slab: ghost planes only)
        const double *p = coarse + 3 * nidx(c, i0 + qi, j0 + qj, live ? k0 : 0);
        double c0[3], c1[3];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) c0[cc] = p[cc];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            c1[cc] = lane_above(c0[cc]);
            if (lane == 63 && up) c1[cc] = p[3 + cc];
        }
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            v0[cc] = fma(w0, c0[cc], v0[cc]);
            v1[cc] = fma(w1, c0[cc], v1[cc]);
            v1[cc] = fma(w1, c1[cc], v1[cc]);
        }
    }
    const long long row = ((long long) i * FY + j) * FZ;
    if (!ACC && fixed != nullptr && live) {
        const unsigned f0 = fixed[row + 2 * k0], f1 = up ? fixed[row + 2 * k0 + 1] : 0u;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            if ((f0 >> cc) & 1) v0[cc] = 0.0;
            if ((f1 >> cc) & 1) v1[cc] = 0.0;
        }
    }
    double *s = sh[wv];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) { s[6 * lane + cc] = v0[cc]; s[6 * lane + 3 + cc] = v1[cc]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nd = 3 * min(128, FZ - 128 * chunk);                           // doubles of this wave's piece of the fine row
    double *out = fine + 3 * row + 384 * chunk;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int d = 2 * (lane + 64 * r);
        if (d + 1 < nd) {
            double2 x = *reinterpret_cast<const double2 *>(s + d);
            if (ACC) { x.x += out[d]; x.y += out[d + 1]; out[d] = x.x; out[d + 1] = x.y; }
            else { __builtin_nontemporal_store(x.x, out + d); __builtin_nontemporal_store(x.y, out + d + 1); }
        } else if (d < nd) {
            out[d] = ACC ? out[d] + s[d] : s[d];
        }
    }
}

void launch_prolong(const Dims &c, int fineNX, int shift, const double *coarse, double *fine, int accumulate, hipStream_t s,
                    const uint8_t *fixed) {
    const int FY = 2 * c.ny + 1;
    dim3 blk(64, 4, 1), grd((c.NZ + 63) / 64, (FY + 3) / 4, fineNX);
    if (accumulate) k_prolong_rows<true><<<grd, blk, 0, s>>>(c, shift, coarse, fine, nullptr);
    else            k_prolong_rows<false><<<grd, blk, 0, s>>>(c, shift, coarse, fine, fixed);
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// Galerkin coarse element matrices (MG.hh:604-669).  One 576-thread block per coarse element,
// thread t owns entry t of the 24x24 result.  phi[g][i][j] = coarse shape function j at node i of
// child g (MG.hh:559-583), children indexed g = 4 gx + 2 gy + gz here.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double phi_val(int g, int fn, int cn) {
    double v = 1.0;
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) {
        const int sh = 2 - dd;
        const double p = 0.5 * ((fn >> sh) & 1) + 0.5 * ((g >> sh) & 1);
        v *= ((cn >> sh) & 1) ? p : (1.0 - p);
    }
    return v;
}

template <int MODE>
__global__ void __launch_bounds__(576) k_coarsen_ke(Dims c, const double *__restrict__ cK0, const double *__restrict__ Ef,
                                                    const double *__restrict__ Kef, double *__restrict__ Kec) {
    __shared__ double Kf[576];
    __shared__ double T[576];
    __shared__ double ph[64];
    const int t = threadIdx.x;
    const long long ec = blockIdx.x;
    const int ez = (int) (ec % c.nz), ey = (int) ((ec / c.nz) % c.ny), ex = (int) (ec / ((long long) c.nz * c.ny));
    const long long ny1 = 2LL * c.ny, nz1 = 2LL * c.nz;          // child-level element dims
    double acc = 0.0;
    // stored child matrices: the entry of child g + 1 is requested before child g is multiplied (one after the other the block was a
    // chain of eight exposed memory round trips)
    auto child_entry = [&](int g) {
        const long long cx_ = 2LL * ex + ((g >> 2) & 1), cy_ = 2LL * ey + ((g >> 1) & 1), cz_ = 2LL * ez + (g & 1);
        return Kef[((cx_ * ny1 + cy_) * nz1 + cz_) * 576 + t];
    };
    double kf_cur = MODE != 1 ? child_entry(0) : 0.0;
    for (int g = 0; g < 8; ++g) {
        const int gx = (g >> 2) & 1, gy = (g >> 1) & 1, gz = g & 1;
        const long long cx_ = 2LL * ex + gx, cy_ = 2LL * ey + gy, cz_ = 2LL * ez + gz;   // child element index
        const double kf_next = (MODE != 1 && g < 7) ? child_entry(g + 1) : 0.0;
        if (MODE == 1) {
            // child matrix = sum_f Efine[f] * cK0[f]; the fine grid is 4x this level
            const long long ny0 = 2 * ny1, nz0 = 2 * nz1;
            double v = 0.0;
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const int fx = (f >> 2) & 1, fy = (f >> 1) & 1, fz = f & 1;
                const double e = Ef[((2 * cx_ + fx) * ny0 + (2 * cy_ + fy)) * nz0 + (2 * cz_ + fz)];
                v = fma(e, cK0[f * 576 + t], v);
            }
            Kf[t] = v;
        } else {
            Kf[t] = kf_cur;
            kf_cur = kf_next;
        }
        if (t < 64) ph[t] = phi_val(g, t >> 3, t & 7);
        __syncthreads();
        {   // T = Kf * I : T[a][3j+dd] = sum_i Kf[a][3i+dd] phi[i][j]
            const int a = t / 24, col = t % 24, j = col / 3, dd = col % 3;
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) v = fma(Kf[a * 24 + 3 * i + dd], ph[i * 8 + j], v);
            T[t] = v;
        }
        __syncthreads();
        {   // Kc += I^T T : Kc[3j+cc][bcol] += sum_i phi[i][j] T[3i+cc][bcol]
            const int row = t / 24, bcol = t % 24, j = row / 3, cc = row % 3;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = fma(ph[i * 8 + j], T[(3 * i + cc) * 24 + bcol], acc);
        }
        __syncthreads();
    }
    Kec[ec * 576 + t] = acc;
}

// level-2 element matrices straight from the 64 fine moduli inside the element: Ke = sum_{g,f} E[g,f] * c2K0[g][f] with
// c2K0[g][f] = I_g^T cK0[f] I_g precomputed on the host (the same sum as MODE 1 of k_coarsen_ke with the two triple products
// folded into the table: 37 k instead of 250 k multiply-adds per element)
constexpr int CK2_NB = 16;      // elements per block: every table entry fetched from L2 serves 16 elements (8: the 295 KB table per block made 11 TB/s of L2 traffic)
__global__ void __launch_bounds__(576) k_coarsen_ke_two_levels(Dims c, const double *__restrict__ c2K0, const double *__restrict__ Ef,
                                                               double *__restrict__ Kec) {
    __shared__ __align__(16) double Es[64][CK2_NB];        // [grandchild][element]: the eight moduli a table entry multiplies are four 16-byte reads
    const int t = threadIdx.x;
    const long long e0 = (long long) blockIdx.x * CK2_NB;
    const long long ny0 = 4LL * c.ny, nz0 = 4LL * c.nz;          // fine element dims
    for (int tq = t; tq < 64 * CK2_NB; tq += 576) {
        const int b = tq >> 6, q = tq & 63, g = q >> 3, f = q & 7;
        const long long ec = e0 + b;
        double v = 0.0;
        if (ec < c.ne) {
            const int ez = (int) (ec % c.nz), ey = (int) ((ec / c.nz) % c.ny), ex = (int) (ec / ((long long) c.nz * c.ny));
            const long long fx = 4LL * ex + 2 * ((g >> 2) & 1) + ((f >> 2) & 1), fy = 4LL * ey + 2 * ((g >> 1) & 1) + ((f >> 1) & 1),
                            fz = 4LL * ez + 2 * (g & 1) + (f & 1);
            v = Ef[(fx * ny0 + fy) * nz0 + fz];
        }
        Es[q][b] = v;
    }
    __syncthreads();
    double acc[CK2_NB];
#pragma unroll
    for (int b = 0; b < CK2_NB; ++b) acc[b] = 0.0;
#pragma unroll 4
    for (int q = 0; q < 64; ++q) {
        const double k = c2K0[q * 576 + t];
        typedef double d2v_t __attribute__((ext_vector_type(2), aligned(16)));
#pragma unroll
        for (int b = 0; b < CK2_NB; b += 2) {
            const d2v_t e = *reinterpret_cast<const d2v_t *>(&Es[q][b]);
            acc[b] = fma(e.x, k, acc[b]);
            acc[b + 1] = fma(e.y, k, acc[b + 1]);
        }
    }
#pragma unroll
    for (int b = 0; b < CK2_NB; ++b)
        if (e0 + b < c.ne) Kec[(e0 + b) * 576 + t] = acc[b];
}

void launch_coarsen_ke(const Dims &c, int mode, const double *cK0, const double *Efine, const double *Kef,
                       double *Kec, hipStream_t s) {
    if (mode == 3) k_coarsen_ke_two_levels<<<dim3((unsigned) ((c.ne + CK2_NB - 1) / CK2_NB)), dim3(576), 0, s>>>(c, cK0, Efine, Kec);
    else if (mode == 1) k_coarsen_ke<1><<<dim3((unsigned) c.ne), dim3(576), 0, s>>>(c, cK0, Efine, Kef, Kec);
    else           k_coarsen_ke<2><<<dim3((unsigned) c.ne), dim3(576), 0, s>>>(c, cK0, Efine, Kef, Kec);
    VFEM_HIP(hipGetLastError());
}

// stencil block (n, n+off) = sum over elements containing both of Ke[e][ln rows][lm cols]
// SRC 0: Ke = E*K0, SRC 1: virtual level-1, SRC 2: stored Ke
template <int SRC>
__global__ void __launch_bounds__(256) k_stencil_build(Dims d, const double *__restrict__ K, const double *__restrict__ E,
                                                       double *__restrict__ St) {
    const long long gid = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= d.nn * 27) return;
    const long long n = gid % d.nn;
    const int nb = (int) (gid / d.nn);
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
            if (SRC == 0) {
                const double Ee = E[eidx(d, e3[0], e3[1], e3[2])];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) A[3 * r + c] = fma(Ee, K[(3 * ln + r) * 24 + 3 * lm + c], A[3 * r + c]);
            } else if (SRC == 1) {
                const long long nyf = 2LL * d.ny, nzf = 2LL * d.nz;
                for (int f = 0; f < 8; ++f) {
                    const int fx = (f >> 2) & 1, fy = (f >> 1) & 1, fz = f & 1;
                    const double Ef = E[((2LL * e3[0] + fx) * nyf + (2LL * e3[1] + fy)) * nzf + (2LL * e3[2] + fz)];
                    const double *Kf = K + f * 576;
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 3; ++c) A[3 * r + c] = fma(Ef, Kf[(3 * ln + r) * 24 + 3 * lm + c], A[3 * r + c]);
                }
            } else {
                const double *Ke = K + eidx(d, e3[0], e3[1], e3[2]) * 576;
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) A[3 * r + c] += Ke[(3 * ln + r) * 24 + 3 * lm + c];
            }
        }
    }
    long long sbase, scnt;
    cm_index(d, i, j, k, sbase, scnt);
#pragma unroll
    for (int q = 0; q < 9; ++q) St[sbase + ((long long) nb * 9 + q) * scnt] = A[q];
}

// The same stencil from stored element matrices, one workgroup per 64-node tile of the colour-major storage.  k_stencil_build<2>
// gives a thread one (node, neighbour) pair: its nine reads per incident element are 8 bytes out of lines 4.6 KB apart from the next
// lane's, and the other neighbours of the node -- the rest of those lines -- are visited by launches-worth of other threads much
// later: 12.6 ms for the 129^3 level (9.7 GB of element matrices, 4.2 GB of stencil).  Here the rows of a node in one incident
// element are read as what they are, 72 contiguous doubles (K_e[3 ln .. 3 ln + 2][0 .. 23]), by 72 consecutive threads, summed
// into the tile image in LDS (243 x 64 doubles: the storage layout itself) element by element in the old order -- bit for bit the
// same stencil -- and the tile leaves as one contiguous 124 KB block.
struct TileStarts { int v[9]; };
constexpr int ST_LD = 65;            // LDS row length of the tile image (64 + 1: the 72 entries of a node's chunk would otherwise share a bank)
__global__ void __launch_bounds__(256) k_stencil_tiles(Dims d, const double *__restrict__ Ke, double *__restrict__ St, TileStarts ts) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *st = reinterpret_cast<double *>(smem_raw);               // [243][ST_LD]
    __shared__ int sn[64][4];                                         // i, j, k of the tile's nodes (k < 0: padding beyond the colour)
    int c = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) c += (int) blockIdx.x >= ts.v[q];
    const int t = (int) blockIdx.x - ts.v[c];
    const int ci = (c >> 2) & 1, cj = (c >> 1) & 1, ck = c & 1;
    const int cny = cj ? d.NY >> 1 : (d.NY + 1) >> 1, cnz = ck ? d.NZ >> 1 : (d.NZ + 1) >> 1, cnx = ci ? d.NX >> 1 : (d.NX + 1) >> 1;
    const long long cnt = (long long) cnx * cny * cnz;
    for (int q = threadIdx.x; q < 243 * ST_LD; q += 256) st[q] = 0.0;
    if (threadIdx.x < 64) {
        const long long q = 64LL * t + threadIdx.x;
        const bool ok = q < cnt;
        const long long qq = ok ? q : 0;
        sn[threadIdx.x][0] = 2 * (int) (qq / ((long long) cnz * cny)) + ci;
        sn[threadIdx.x][1] = 2 * (int) ((qq / cnz) % cny) + cj;
        sn[threadIdx.x][2] = ok ? 2 * (int) (qq % cnz) + ck : -1000;
    }
    __syncthreads();
    // thread = one of the 72 doubles of a node's chunk (x: row r of the node, column (lm, cc)) for every third node of the tile
    const int x = threadIdx.x % 72, g = threadIdx.x / 72;
    const int r = x / 24, col = x - 24 * r, lm = col / 3, cc = col - 3 * lm;
    const int lmx = (lm >> 2) & 1, lmy = (lm >> 1) & 1, lmz = lm & 1, qe = 3 * r + cc;
    for (int sel = 0; sel < 8; ++sel) {
        const int bx = (sel >> 2) & 1, by = (sel >> 1) & 1, bz = sel & 1;
        const int ln = 4 * (1 - bx) + 2 * (1 - by) + (1 - bz);       // the node's corner in the element i - 1 + bit
        // neighbour the column belongs to, relative to the node: element origin offset (bit - 1) + corner of the column
        const int nb = (bx - 1 + lmx + 1) * 9 + (by - 1 + lmy + 1) * 3 + (bz - 1 + lmz + 1);
        if (g < 3) {
            // all 22 loads of the element slot in flight before the first is added (one at a time the tile was a chain of 176 round trips)
            double v[22];
#pragma unroll
            for (int it = 0; it < 22; ++it) {
                const int p = g + 3 * it;
                v[it] = 0.0;
                if (p < 64) {
                    const int ex = sn[p][0] - 1 + bx, ey = sn[p][1] - 1 + by, ez = sn[p][2] - 1 + bz;
                    if (ex >= 0 && ex < d.nx && ey >= 0 && ey < d.ny && ez >= 0 && ez < d.nz) v[it] = Ke[eidx(d, ex, ey, ez) * 576 + (3 * ln) * 24 + x];
                }
            }
#pragma unroll
            for (int it = 0; it < 22; ++it) {
                const int p = g + 3 * it;
                if (p < 64) st[(nb * 9 + qe) * ST_LD + p] += v[it];
            }
        }
        __syncthreads();
    }
    long long before = 0;
    for (int q = 0; q < c; ++q) {
        const long long qx = ((q >> 2) & 1) ? d.NX >> 1 : (d.NX + 1) >> 1, qy = ((q >> 1) & 1) ? d.NY >> 1 : (d.NY + 1) >> 1, qz = (q & 1) ? d.NZ >> 1 : (d.NZ + 1) >> 1;
        before += cm_padded(qx * qy * qz);
    }
    double *dst = St + 243 * before + (long long) t * (243 * 64);
    for (int q = threadIdx.x; q < 243 * 64; q += 256) dst[q] = st[(q >> 6) * ST_LD + (q & 63)];
}

void launch_stencil_from_ke(const Dims &d, const double *Ke, double *S, hipStream_t s) {
    TileStarts ts;
    int total = 0;
    for (int c = 0; c < 8; ++c) {
        const long long cx = ((c >> 2) & 1) ? d.NX >> 1 : (d.NX + 1) >> 1, cy = ((c >> 1) & 1) ? d.NY >> 1 : (d.NY + 1) >> 1, cz = (c & 1) ? d.NZ >> 1 : (d.NZ + 1) >> 1;
        ts.v[c] = total;
        total += (int) (cm_padded(cx * cy * cz) / 64);
    }
    ts.v[8] = total;
    static bool attr = false;
    if (!attr) {
        VFEM_HIP(hipFuncSetAttribute((const void *) k_stencil_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 243 * ST_LD * 8));
        attr = true;
    }
    if (total > 0) k_stencil_tiles<<<dim3((unsigned) total), dim3(256), 243 * ST_LD * 8, s>>>(d, Ke, S, ts);
    VFEM_HIP(hipGetLastError());
}

void launch_stencil_from_mf(const Dims &d, OpKind kind, const double *K, const double *E, double *S, hipStream_t s) {
    const long long total = d.nn * 27;
    if (kind == OP_MF0) k_stencil_build<0><<<dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s>>>(d, K, E, S);
    else                k_stencil_build<1><<<dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s>>>(d, K, E, S);
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// coarsest level: dense matrix with fixed rows/columns replaced by identity (the reference removes
// them, TPS.hh:834-865), inverted once per operator update, applied as a symmetric GEMV.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dense_from_stencil(Dims d, const double *__restrict__ St,
                                                            const uint8_t *__restrict__ mask, double *__restrict__ A) {
    const long long gid = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= d.nn * 27) return;
    const long long n = gid % d.nn;
    const int nb = (int) (gid / d.nn);
    const int k = (int) (n % d.NZ), j = (int) ((n / d.NZ) % d.NY), i = (int) (n / ((long long) d.NZ * d.NY));
    const int ii = i + nb / 9 - 1, jj = j + (nb / 3) % 3 - 1, kk = k + nb % 3 - 1;
    if (ii < 0 || ii >= d.NX || jj < 0 || jj >= d.NY || kk < 0 || kk >= d.NZ) return;
    const long long m = nidx(d, ii, jj, kk);
    const long long N3 = 3 * d.nn;
    long long sbase, scnt;
    cm_index(d, i, j, k, sbase, scnt);
    const uint8_t mn = mask[n], mm = mask[m];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double v = St[sbase + ((long long) nb * 9 + 3 * r + c) * scnt];
            const bool fr = (mn >> r) & 1, fc = (mm >> c) & 1;
            if (fr || fc) v = (n == m && r == c) ? 1.0 : 0.0;
            A[(3 * n + r) * N3 + 3 * m + c] = v;
        }
}

void launch_dense_from_stencil(const Dims &d, const double *S, const uint8_t *mask, double *A, hipStream_t s) {
    const long long total = d.nn * 27;
    k_dense_from_stencil<<<dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s>>>(d, S, mask, A);
    VFEM_HIP(hipGetLastError());
}

// after potrf+potri(lower, column-major == upper, row-major): mirror the computed triangle and zero the
// rows/columns of fixed dofs so that x_fixed = 0 and rhs_fixed is ignored.
__global__ void __launch_bounds__(256) k_dense_finish(long long n, const uint8_t *__restrict__ mask, double *__restrict__ A) {
    const long long gid = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * n) return;
    const long long r = gid / n, c = gid % n;
    const bool fr = (mask[r / 3] >> (r % 3)) & 1, fc = (mask[c / 3] >> (c % 3)) & 1;
    if (fr || fc) { A[gid] = 0.0; return; }
    if (c < r) A[gid] = A[c * n + r];   // row-major upper triangle (c >= r) holds the result
}

void launch_dense_finish_inverse(long long n, const uint8_t *mask, double *A, hipStream_t s) {
    k_dense_finish<<<dim3((unsigned) ((n * n + 255) / 256)), dim3(256), 0, s>>>(n, mask, A);
    VFEM_HIP(hipGetLastError());
}

// y = A x, one wave per row; four independent 16-byte loads of the row in flight per lane (the rows come from L2: with one
// 8-byte load per round trip a 2187-column row took 34 dependent trips and the launch 370 us)
__global__ void __launch_bounds__(256) k_gemv(long long n, const double *__restrict__ A, const double *__restrict__ x,
                                              double *__restrict__ y) {
    const long long row = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const double *a = A + row * n;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    long long c = lane;
    for (; c + 192 < n; c += 256) {
        const double a0 = a[c], a1 = a[c + 64], a2 = a[c + 128], a3 = a[c + 192];
        const double x0 = x[c], x1 = x[c + 64], x2 = x[c + 128], x3 = x[c + 192];
        acc[0] = fma(a0, x0, acc[0]); acc[1] = fma(a1, x1, acc[1]); acc[2] = fma(a2, x2, acc[2]); acc[3] = fma(a3, x3, acc[3]);
    }
    for (; c < n; c += 64) acc[0] = fma(a[c], x[c], acc[0]);
    double r = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o);
    if (lane == 0) y[row] = r;
}

void launch_gemv_sym(long long n, const double *A, const double *x, double *y, hipStream_t s) {
    k_gemv<<<dim3((unsigned) ((n + 3) / 4)), dim3(256), 0, s>>>(n, A, x, y);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
