// Level-0 multicoloured block Gauss-Seidel (MG.hh:193-340, matrix-free Ke = E_e K0) as an x-MARCH with the node planes
// resident in LDS.
//
// The eight colours of the reference sweep are (cx, cy, cz) with cx slowest (MG.hh:292-310): during the four colours of one cx
// only the planes of that x-parity change, and they couple to each other only THROUGH the planes of the other parity, which
// stand still.  Every plane of the active parity is therefore an independent two-dimensional four-colour problem, and a
// workgroup that owns a (y, z) tile can march along x in steps of two planes: plane x-1 and x+1 (fixed) and plane x (relaxed)
// sit in LDS, all four in-plane colours of the tile are relaxed there, the finished plane is written out once.  A half sweep
// then reads every plane once and writes half of them -- the row-streaming kernels (kernels_mg.hip: k_gs_rows_mf0_pair) read
// the nine neighbour rows of every row again in every colour pass (measured 36 GB per sweep at 512^3 against 10.7 GB).
//
// * Staging.  Node planes and element-moduli layers are brought in by `global_load_lds_dwordx4` (one dedicated DMA wave; the
//   idiom and the 16-byte-grid addressing of kernels_apply_dma.hip): 5 node-plane slots (x-1, x, x+1 resident, x+2, x+3 in
//   flight) and 4 moduli slots, 147 KB of LDS, one workgroup of 7 compute waves + 1 DMA wave per CU.
// * Tile seams.  A tile cannot see its neighbours' updates, so it recomputes what it needs of them: with the colour order
//   (p,p), (p,q), (q,p), (q,q) (p = 0 forward, 1 reverse; q = 1 - p) the nodes a tile OWNS are 2R rows x 2C columns starting at
//   a row / column of parity p; colour k is relaxed on  rows [yb + (k>>1), yb + 2R - (k>>1)]  and  columns [zb - 2 + k, zb + 2C +
//   2 - k]  of its parity, and one more ring of old values is loaded (15 x 65 staged node columns for 12 x 58 owned ones).
//   Recomputed nodes run the same instruction sequence on the same inputs in every tile, so they agree bit for bit.
// * Out of place.  Tiles read OLD halo values of their neighbours, so a half sweep must not overwrite its input: it reads the
//   relaxed parity from `uR`, the other parity from `uO` and writes the relaxed planes to `dst` (!= uR); the caller
//   ping-pongs between the field and one scratch vector (capi.hip: mg_smooth_n), two sweeps end where they began.
// * Arithmetic.  A wave relaxes one row of the active colour: lanes 0-31 hold the nodes' four element slots on the low-x side,
//   lanes 32-63 the four on the high-x side, computed by the SAME instructions on x-mirrored data (K0 commutes with the
//   reflection: u_x -> -u_x, S_x -> -S_x, exact), 288 multiply-adds each with K0 in 72 SGPRs (gs_coef.h); the halves meet by
//   lane shuffle and lanes 0-31 do the 3x3 component-sequential solve (MG.hh:254-264).  The summation order differs from the
//   row kernels' (planes x-1, x | x+1, x instead of x-1, x, x+1 per slot), so the two agree to rounding, not bit for bit.
#include "vfem_internal.h"
#include "device_utils.h"
#include "gs_coef.h"

namespace vfem {

namespace gsm {
constexpr int R = 6;                          // owned row pairs of a tile (2R owned node rows)
constexpr int C = 29;                         // owned column pairs (2C owned node columns); C + 3 = 32 lanes in the widest colour
constexpr int CW = R + 1;                     // compute waves
constexpr int LY = 2 * R + 3, LZ = 2 * C + 7; // staged node rows / columns (15 x 65)
constexpr int EY = 2 * R + 2, EZ = 2 * C + 6; // staged element rows / columns (14 x 64)
constexpr int PU = (LZ * 24 + 8 + 15) / 16;   // 16-byte pieces per staged node row incl. the alignment shift (98)
constexpr int PE = (EZ * 8 + 8 + 15) / 16;    // per staged element row (33)
constexpr int ROW_D = 2 * PU, EROW_D = 2 * PE;
constexpr int U_INSTR = (LY * PU + 63) / 64;  // DMA instructions per node plane (23)
constexpr int E_INSTR = (EY * PE + 63) / 64;  // per element layer (8)
constexpr int U_SLOT_D = U_INSTR * 128, E_SLOT_D = E_INSTR * 128;      // doubles per slot
constexpr int NU = 5, NE = 4;
constexpr size_t LDS_BYTES = (size_t) (NU * U_SLOT_D + NE * E_SLOT_D) * 8;
static_assert(C + 3 == 32, "the widest colour fills a half wave");
static_assert(LDS_BYTES <= 160 * 1024, "ring must fit the LDS of a CU");
static_assert(2 * (U_INSTR + E_INSTR) <= 63, "one step of DMA must fit the 6-bit vmcnt");
}  // namespace gsm

struct GsMarchArgs {
    Dims d;
    const double *coef;            // 36 resident coefficients (build_gs_coef)
    const double *E;               // moduli of the level's elements, [nx][ny][nz]
    const char *e_first, *e_last;  // first / last admissible 16-byte piece of the moduli allocation
    const double *uR, *uO;         // current values of the planes of the relaxed parity / of the other parity
    const char *uR_first, *uR_last, *uO_first, *uO_last;
    double *dst;                   // receives the relaxed planes (other planes untouched)
    const double *b;
    const uint8_t *mask;
    int cxl;                       // local x parity of the relaxed planes
    int forward;                   // component order of the 3x3 solve (MG.hh:254-264)
    int steps_per_chunk;           // relaxed planes per block
};

typedef double d2a_t __attribute__((ext_vector_type(2), aligned(16)));

__device__ __forceinline__ void gsm_glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                     (__attribute__((address_space(3))) void *) l, 16, 0, 0);
}
__device__ __forceinline__ double flip(double v, unsigned long long sgn) {
    return __longlong_as_double(__double_as_longlong(v) ^ (long long) sgn);
}
// nine consecutive doubles from LDS; `odd`: the (wave-uniform) parity of idx, so that the 16-byte reads are aligned
__device__ __forceinline__ void read9(const double *s, int idx, bool odd, double v[9]) {
    if (odd) {
        v[0] = s[idx];
        const d2a_t a = *reinterpret_cast<const d2a_t *>(s + idx + 1), b = *reinterpret_cast<const d2a_t *>(s + idx + 3);
        const d2a_t c = *reinterpret_cast<const d2a_t *>(s + idx + 5), e = *reinterpret_cast<const d2a_t *>(s + idx + 7);
        v[1] = a[0]; v[2] = a[1]; v[3] = b[0]; v[4] = b[1]; v[5] = c[0]; v[6] = c[1]; v[7] = e[0]; v[8] = e[1];
    } else {
        const d2a_t a = *reinterpret_cast<const d2a_t *>(s + idx), b = *reinterpret_cast<const d2a_t *>(s + idx + 2);
        const d2a_t c = *reinterpret_cast<const d2a_t *>(s + idx + 4), e = *reinterpret_cast<const d2a_t *>(s + idx + 6);
        v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1]; v[4] = c[0]; v[5] = c[1]; v[6] = e[0]; v[7] = e[1];
        v[8] = s[idx + 8];
    }
}

template <int P>      // parity of the first in-plane colour: 0 forward colour order, 1 reverse
__global__ void __launch_bounds__(64 * (gsm::CW + 1)) k_gs_march_mf0(GsMarchArgs A) {
    using namespace gsm;
    extern __shared__ __align__(16) unsigned char smem[];
    double *sU = reinterpret_cast<double *>(smem);
    double *sE = sU + NU * U_SLOT_D;
    const Dims &d = A.d;
    const int lane = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.y);

    // relaxed planes x = cxl + 2 mm, mm in [m0, m1)
    const int M = (d.NX - 1 - A.cxl) / 2 + 1;
    const int m0 = blockIdx.x * A.steps_per_chunk;
    if (A.cxl > d.NX - 1 || m0 >= M) return;                          // block-uniform, before any barrier
    const int m1 = m0 + A.steps_per_chunk < M ? m0 + A.steps_per_chunk : M;
    const int nsteps = m1 - m0;
    const int x0 = A.cxl + 2 * m0;
    // plane stream of this block: j = 0, 1, 2, ... <-> planes x0 - 1 + j; even j: fixed planes (uO), odd j: relaxed planes (uR).
    // a plane outside the grid is replaced by the nearest one of its parity (its values only ever meet elements outside the
    // grid, whose modulus is taken as 0)
    auto plane_of = [&](int j) { int i = x0 - 1 + j; if (i < 0) i += 2; if (i > d.NX - 1) i -= 2; return i; };

    const int yb = 2 * R * (int) blockIdx.z - P, zb = 2 * C * (int) blockIdx.y - P;     // first owned row / column (parity P)
    const int yl = yb - 1, zl = zb - 3;                                 // node row / column of staged index 0
    const int k0u = zl < -1 ? -1 : zl;                                  // first node column held by a staged node row
    const int cshift = k0u - zl;
    const int ek0u = zl < 0 ? 0 : zl;                                   // first element column held by a staged element row
    const int ecshift = ek0u - zl;

    const long long plane = (long long) d.NY * d.NZ, elayer = (long long) d.ny * d.nz;
    const int ppar = (int) ((3 * plane) & 1), epar = (int) (elayer & 1);
    const int bparR = (int) ((reinterpret_cast<uintptr_t>(A.uR) >> 3) & 1), bparO = (int) ((reinterpret_cast<uintptr_t>(A.uO) >> 3) & 1);
    const int bparE = (int) ((reinterpret_cast<uintptr_t>(A.E) >> 3) & 1);
    auto row_par = [&](int ry) {                                        // parity of the first double of staged node row ry (before base / plane)
        int jj = yl + ry; jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
        return ((jj & d.NZ) ^ k0u) & 1;                                 // (3 (jj NZ + k0u)) & 1
    };

    // =========================== DMA wave ===========================
    if (wave == CW) {
        unsigned ugo[U_INSTR], ego[E_INSTR];          // (double offset of the lane's piece from the plane / layer start) * 2 + row-start parity
#pragma unroll
        for (int t = 0; t < U_INSTR; ++t) {
            const int Pc = 64 * t + lane;
            int r = Pc / PU, c = Pc - r * PU;
            if (r > LY - 1) { r = LY - 1; c = PU - 1; }
            int jj = yl + r; jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
            const long long rs = 3LL * ((long long) jj * d.NZ + k0u);   // >= -3
            ugo[t] = (unsigned) ((rs + 3 + 2LL * c) * 2 + (rs & 1));    // offset biased by +3 doubles so that it is never negative
        }
#pragma unroll
        for (int t = 0; t < E_INSTR; ++t) {
            const int Pc = 64 * t + lane;
            int r = Pc / PE, c = Pc - r * PE;
            if (r > EY - 1) { r = EY - 1; c = PE - 1; }
            int jj = yl + r; jj = jj < 0 ? 0 : (jj > d.ny - 1 ? d.ny - 1 : jj);
            const long long rs = (long long) jj * d.nz + ek0u;
            ego[t] = (unsigned) ((rs + 2LL * c) * 2 + (rs & 1));
        }
        auto issueU = [&](int j) {
            const int i = plane_of(j);
            const bool rel = j & 1;
            const double *buf = rel ? A.uR : A.uO;
            const char *first = rel ? A.uR_first : A.uO_first, *last = rel ? A.uR_last : A.uO_last;
            const int par0 = (rel ? bparR : bparO) + (i & ppar);
            const double *pb = buf + 3LL * i * plane - 3;               // (the bias of ugo)
            unsigned char *slot = reinterpret_cast<unsigned char *>(sU + (j % NU) * U_SLOT_D);
#pragma unroll
            for (int t = 0; t < U_INSTR; ++t) {
                const char *g = reinterpret_cast<const char *>(pb + (long long) (ugo[t] >> 1) - (long long) ((par0 + (int) (ugo[t] & 1)) & 1));
                g = g > last ? last : (g < first ? first : g);
                gsm_glds16(g, slot + 1024 * t);
            }
        };
        auto issueE = [&](int j) {
            int il = x0 - 1 + j; il = il < 0 ? 0 : (il > d.nx - 1 ? d.nx - 1 : il);
            const int par0 = bparE + (il & epar);
            const double *pb = A.E + (long long) il * elayer;
            unsigned char *slot = reinterpret_cast<unsigned char *>(sE + (j % NE) * E_SLOT_D);
#pragma unroll
            for (int t = 0; t < E_INSTR; ++t) {
                const char *g = reinterpret_cast<const char *>(pb + (long long) (ego[t] >> 1) - (long long) ((par0 + (int) (ego[t] & 1)) & 1));
                g = g > A.e_last ? A.e_last : (g < A.e_first ? A.e_first : g);
                gsm_glds16(g, slot + 1024 * t);
            }
        };
        issueU(0); issueU(1); issueU(2); issueE(0); issueE(1);
        for (int m = 0; m < nsteps; ++m) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // B0: the planes of step m have landed; step m-1 is finished
            if (m + 1 < nsteps) { issueU(2 * m + 3); issueU(2 * m + 4); issueE(2 * m + 2); issueE(2 * m + 3); }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // =========================== compute waves ===========================
    GsCoef ck;
    gs_load_coef<true>(A.coef, ck);
    const int h = lane >> 5, cl = lane & 31;                            // half (0: low-x element slots, 1: high-x, mirrored), column index
    const unsigned long long sgn = h ? 0x8000000000000000ull : 0ull;

    for (int m = 0; m < nsteps; ++m) {
        const int x = x0 + 2 * m;
        __builtin_amdgcn_s_barrier();                                   // B0
        const int midoff = ((2 * m + 1) % NU) * U_SLOT_D;
        const int faroff = ((2 * m + 2 * h) % NU) * U_SLOT_D;           // per lane: plane x-1 (h = 0) or x+1 (h = 1)
        const int shM = bparR + (x & ppar);                             // + row parity = alignment shift of the staged rows
        const int shF = bparO + (plane_of(2 * m) & ppar);               // (planes x-1 and x+1 have the same parity)
        const int il = x - 1 + h;                                       // element layer of this half
        const bool layer_ok = il >= 0 && il < d.nx;
        const int ilc = il < 0 ? 0 : (il > d.nx - 1 ? d.nx - 1 : il);
        const int eoff = ((2 * m + h) % NE) * E_SLOT_D;
        const int esh0 = bparE + (ilc & epar);

        static_for<4>([&](auto kc) {
            constexpr int k = decltype(kc)::value, ro = k >> 1;
            constexpr int nrows = R + 1 - ro, ncols = C + 3 - k;
            if (wave < nrows) {
                const int ry = 1 + ro + 2 * wave;                       // staged row of the node
                const int y = yl + ry;
                const int ce = cl < ncols ? cl : ncols - 1;             // lanes beyond the colour's columns shadow the last one
                const int czn = 1 + k + 2 * ce;                         // staged column
                const int z = zl + czn;
                const bool node_ok = cl < ncols && y >= 0 && y < d.NY && z >= 0 && z < d.NZ;
                const bool mine = node_ok && h == 0;
                // right-hand side and mask (consumed at the end of the phase)
                const long long n = ((long long) x * d.NY + (y < 0 ? 0 : (y > d.NY - 1 ? d.NY - 1 : y))) * d.NZ + (z < 0 ? 0 : (z > d.NZ - 1 ? d.NZ - 1 : z));
                double bv[3] = {0.0, 0.0, 0.0};
                uint8_t mk = 0;
                if (mine) {
                    bv[0] = A.b[3 * n]; bv[1] = A.b[3 * n + 1]; bv[2] = A.b[3 * n + 2];
                    mk = A.mask[n];
                }
                // the four moduli of this half: elements (y - 1 + dj, z - 1 + dk) of layer il
                double e4[4];
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    const int dj = sl >> 1, dk = sl & 1;
                    const int ey = y - 1 + dj, ez = z - 1 + dk;
                    const bool ok = node_ok && layer_ok && ey >= 0 && ey < d.ny && ez >= 0 && ez < d.nz;
                    const int eyc = ey < 0 ? 0 : (ey > d.ny - 1 ? d.ny - 1 : ey);
                    int ec = czn - 1 + dk - ecshift; ec = ec < 0 ? 0 : ec;
                    const int esh = (esh0 + (((eyc & d.nz) ^ ek0u) & 1)) & 1;
                    const double v = sE[eoff + (ry - 1 + dj) * EROW_D + ec + esh];
                    e4[sl] = ok ? v : 0.0;
                }
                int ni = czn - 1 - cshift;                              // staged node index of the window's first node (z - 1)
                if (ni < 0) ni &= 1;                                    // (outside the grid: any window of the right alignment)
                const int nipar = (k + cshift) & 1;                     // parity of ni, wave-uniform

                double T[4][3];
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) { T[sl][0] = 0.0; T[sl][1] = 0.0; T[sl][2] = 0.0; }
                double uself[3] = {0.0, 0.0, 0.0};
                static_for<6>([&](auto rc6) {
                    constexpr int t = decltype(rc6)::value / 3, dy = decltype(rc6)::value % 3 - 1;      // t = 0: far plane, 1: the node's plane
                    const int rr = ry + dy;
                    const int sh = ((t ? shM : shF) + row_par(rr)) & 1;
                    const int idx = (t ? midoff : faroff) + rr * ROW_D + 3 * ni + sh;
                    double v[9];
                    read9(sU, idx, ((nipar + sh) & 1) != 0, v);
                    if (t == 1 && dy == 0) { uself[0] = v[3]; uself[1] = v[4]; uself[2] = v[5]; }
                    v[0] = flip(v[0], sgn); v[3] = flip(v[3], sgn); v[6] = flip(v[6], sgn);
                    // elements of this half touching the row: dj - 1 + my == dy
                    static_for<4>([&](auto ec) {
                        constexpr int dj = decltype(ec)::value >> 1, my = decltype(ec)::value & 1;
                        if constexpr (dj - 1 + my == dy) {
                            static_for<4>([&](auto zc) {
                                constexpr int dk = decltype(zc)::value >> 1, mz = decltype(zc)::value & 1;
                                constexpr int n3 = dk + mz;
                                constexpr int ln = 4 + 2 * (1 - dj) + (1 - dk), lm = 4 * t + 2 * my + mz;
                                static_for<9>([&](auto qc) {
                                    constexpr int r = decltype(qc)::value / 3, c = decltype(qc)::value % 3;
                                    T[2 * dj + dk][r] = fma(gs_coef_at<ln, r, lm, c>(ck), v[3 * n3 + c], T[2 * dj + dk][r]);
                                });
                            });
                        }
                    });
                });
                // partial sums of this half (in the mirrored frame for h = 1), then back to the node's frame
                double Sp[3] = {0.0, 0.0, 0.0}, Mp[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) Mp[q] = 0.0;
                static_for<4>([&](auto sc) {
                    constexpr int sl = decltype(sc)::value, dj = sl >> 1, dk = sl & 1, ln = 4 + 2 * (1 - dj) + (1 - dk);
                    static_for<3>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        Sp[r] = fma(e4[sl], T[sl][r], Sp[r]);
                        static_for<3>([&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            Mp[3 * r + c] = fma(e4[sl], gs_coef_at<ln, r, ln, c>(ck), Mp[3 * r + c]);
                        });
                    });
                });
                Sp[0] = flip(Sp[0], sgn);
                Mp[1] = flip(Mp[1], sgn); Mp[2] = flip(Mp[2], sgn); Mp[3] = flip(Mp[3], sgn); Mp[6] = flip(Mp[6], sgn);
#pragma unroll
                for (int q = 0; q < 3; ++q) Sp[q] += __shfl_down(Sp[q], 32, 64);
#pragma unroll
                for (int q = 0; q < 9; ++q) Mp[q] += __shfl_down(Mp[q], 32, 64);
                if (mine) {
                    double bms[3], ud[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) bms[q] = bv[q] - Sp[q];
                    gs_solve(bms, Mp, mk, A.forward != 0, ud);
                    const int iself = midoff + ry * ROW_D + 3 * (czn - cshift) + ((shM + row_par(ry)) & 1);
#pragma unroll
                    for (int q = 0; q < 3; ++q) sU[iself + q] = uself[q] + ud[q];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        });

        // the finished plane: owned rows [yb, yb + 2R) x columns [zb, zb + 2C), dense 8-byte stores row by row
        {
            const int z0 = zb < 0 ? 0 : zb;
            int z1 = zb + 2 * C - 1; z1 = z1 > d.NZ - 1 ? d.NZ - 1 : z1;
            const int nd = 3 * (z1 - z0 + 1);
            for (int row = wave; row < 2 * R; row += CW) {
                const int y = yb + row;
                if (y < 0 || y > d.NY - 1) continue;
                const int ry = 1 + row;
                const double *src = sU + midoff + ry * ROW_D + 3 * (z0 - k0u) + ((shM + row_par(ry)) & 1);
                double *dp = A.dst + 3 * (((long long) x * d.NY + y) * d.NZ + z0);
                for (int i = lane; i < nd; i += 64) dp[i] = src[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// planes of local parity `par` copied from src to dst (the odd sweep left them in the scratch vector)
__global__ void __launch_bounds__(256) k_copy_planes(Dims d, int par, const double *__restrict__ src, double *__restrict__ dst) {
    const long long per = 3LL * d.NY * d.NZ;
    const int i = 2 * blockIdx.y + par;
    if (i > d.NX - 1) return;
    for (long long q = (long long) blockIdx.x * 256 + threadIdx.x; q < per; q += (long long) gridDim.x * 256) dst[i * per + q] = src[i * per + q];
}
void launch_copy_planes(const Dims &d, int par, const double *src, double *dst, hipStream_t s) {
    const long long per = 3LL * d.NY * d.NZ;
    unsigned gx = (unsigned) ((per + 255) / 256);
    if (gx > 64) gx = 64;
    k_copy_planes<<<dim3(gx, (unsigned) ((d.NX + 1) / 2)), 256, 0, s>>>(d, par, src, dst);
    VFEM_HIP(hipGetLastError());
}

// One half sweep (the four colours of one x parity) of the level-0 Gauss-Seidel.  forward: colour order (0,0),(0,1),(1,0),(1,1)
// and components 0,1,2; otherwise the reverse of both.  Reads the relaxed planes from uR and the others from uO, writes the
// relaxed planes to dst (must differ from uR).  Returns false when the kernel cannot run on these buffers.
bool launch_gs_march_mf0(const Dims &d, const double *coef36, const double *E, const double *E_alloc_begin, const double *E_alloc_end,
                         const double *uR, const double *uO, double *dst, const double *b, const uint8_t *mask,
                         int cxl, int forward, int chunks, hipStream_t s) {
    using namespace gsm;
    if (dst == uR) return false;
    if ((reinterpret_cast<uintptr_t>(uR) & 7u) || (reinterpret_cast<uintptr_t>(uO) & 7u) || (reinterpret_cast<uintptr_t>(E) & 7u)) return false;
    if (d.NX < 2 || d.NY < 2 || d.NZ < 2) return false;
    if (cxl > d.NX - 1) return true;
    GsMarchArgs a;
    a.d = d;
    a.coef = coef36;
    a.E = E;
    auto first_piece = [](const void *p) { return reinterpret_cast<const char *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t) 15); };
    auto last_piece = [](const void *end) { return reinterpret_cast<const char *>((reinterpret_cast<uintptr_t>(end) - 1) & ~(uintptr_t) 15); };
    a.e_first = first_piece(E_alloc_begin); a.e_last = last_piece(E_alloc_end);
    a.uR = uR; a.uO = uO;
    a.uR_first = first_piece(uR); a.uR_last = last_piece(uR + 3 * d.nn);
    a.uO_first = first_piece(uO); a.uO_last = last_piece(uO + 3 * d.nn);
    a.dst = dst; a.b = b; a.mask = mask;
    a.cxl = cxl; a.forward = forward;
    const int M = (d.NX - 1 - cxl) / 2 + 1;
    const int P = forward ? 0 : 1;
    const int nty = (d.NY + P + 2 * R - 1) / (2 * R), ntz = (d.NZ + P + 2 * C - 1) / (2 * C);
    if (chunks <= 0) {
        // enough blocks for a few rounds of the 256 CUs, chunks of at least 8 steps
        chunks = 1;
        while ((long long) chunks * nty * ntz < 1536 && M / (chunks + 1) >= 8) ++chunks;
    }
    if (chunks > M) chunks = M;
    a.steps_per_chunk = (M + chunks - 1) / chunks;
    const unsigned gx = (unsigned) ((M + a.steps_per_chunk - 1) / a.steps_per_chunk);
    const dim3 grd(gx, (unsigned) ntz, (unsigned) nty), blk(64, CW + 1, 1);
    static bool attr[2] = {false, false};
    if (P == 0) {
        if (!attr[0]) { VFEM_HIP(hipFuncSetAttribute((const void *) k_gs_march_mf0<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES)); attr[0] = true; }
        k_gs_march_mf0<0><<<grd, blk, LDS_BYTES, s>>>(a);
    } else {
        if (!attr[1]) { VFEM_HIP(hipFuncSetAttribute((const void *) k_gs_march_mf0<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES)); attr[1] = true; }
        k_gs_march_mf0<1><<<grd, blk, LDS_BYTES, s>>>(a);
    }
    VFEM_HIP(hipGetLastError());
    return true;
}

}  // namespace vfem
