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
#include "coef_rows.h"

namespace vfem {

namespace gsm {
constexpr int R = 6;                          // owned row pairs of a tile (2R owned node rows)
constexpr int C = 29;                         // owned column pairs (2C owned node columns); C + 3 = 32 lanes in the widest colour
constexpr int CW = R + 1;                     // compute waves, form 1 (a wave = one row of the colour, as two x-mirrored half waves)
constexpr int CW2 = (R + 2) / 2;               // form 2 (a wave = two rows of the colour, one node per lane)
constexpr int compute_waves(int form) { return form == 2 ? CW2 : CW; }

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
    const double *tab;             // form 2: K0 by neighbour kind, signs folded in (l1m::build_table of K0, class 0: 8 rows of 12 doubles)
    const double *coef;            // 36 resident coefficients (build_gs_coef) followed by the two 24-entry part tables (build_gs_coef_parts)
    const double *E;               // moduli of the level's elements, [nx][ny][nz]
    const char *e_first, *e_last;  // first / last admissible 16-byte piece of the moduli allocation
    const double *uR, *uO;         // current values of the planes of the relaxed parity / of the other parity
    const char *uR_first, *uR_last, *uO_first, *uO_last;
    double *dst;                   // receives the relaxed planes (other planes untouched)
    const double *b;
    const double *sd;              // solve data per node (k_gs_solve_data): inverse diagonal (mask folded in) [+ strict lower part of the diagonal block:
                                   // 6 doubles per node in form 1, 3 in form 2]
    int cxl;                       // local x parity of the relaxed planes
    int forward;                   // component order of the 3x3 solve (MG.hh:254-264)
    int steps_per_chunk;           // relaxed planes per block
    int first_plane, num_planes;   // relaxed planes: first_plane + 2 mm, mm in [0, num_planes)  (first_plane has the parity cxl)
    long long *stamps;             // diagnostic (normally null): s_memtime stamps of one block, [wave][step < 8][16]
};

typedef double d2a_t __attribute__((ext_vector_type(2), aligned(16)));
typedef unsigned int u4q_t __attribute__((ext_vector_type(4)));
typedef unsigned int u2q_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mkd(unsigned lo, unsigned hi) { return __longlong_as_double(((unsigned long long) hi << 32) | lo); }


__device__ __forceinline__ void gsm_glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                     (__attribute__((address_space(3))) void *) l, 16, 0, 0);
}
__device__ __forceinline__ double flip(double v, unsigned long long sgn) {
    return __longlong_as_double(__double_as_longlong(v) ^ (long long) sgn);
}
// The nine doubles of a node window start at an index whose parity depends on the row and the plane (the alignment shift of the
// staged image).  Ten doubles from the 16-byte-aligned index at or below it are read instead, always as five 16-byte reads in the
// same registers whatever the parity; the consumer then indexes w[off + j] with off = the parity, selected by a wave-uniform
// branch around the multiply-adds.  (Reading by parity -- 8 + 4 x 16 bytes or 4 x 16 + 8 -- left the two paths with different
// register layouts, which the compiler reconciled with moves behind a wait for the data: every row's latency was exposed.)
__device__ __forceinline__ void read10(const double *s, int idx_aligned, double w[10]) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const d2a_t a = *reinterpret_cast<const d2a_t *>(s + idx_aligned + 2 * q);
        w[2 * q] = a[0]; w[2 * q + 1] = a[1];
    }
}

__device__ __forceinline__ long long gsm_now() {
    long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");      // 100 MHz wall clock (s_memtime does not advance at the shader clock here)
    return t;
}

// The parity of a window's first double (the consumer's offset into its ten doubles, see read10) is
//     (k + (P+1) + base parity of the buffer + (plane & ppar) + ALT ((P+1) + staged row)) & 1,     ALT = NZ & 1, ppar = (3 NY NZ) & 1:
// tile origins have the parity of P, so nothing in it depends on the tile, and the planes of a launch have fixed parities.  It is
// therefore QF + k + ALT (ro + dy) for the far planes and QM + k + ALT (ro + dy) for the relaxed plane with launch-uniform QF, QM,
// which are template parameters: every register index in the multiply-adds is a compile-time constant.
//
// FORM 2 (round 3, second half): one node per LANE.  Form 1 spends 2 x 270 fp64 operations per node (two mirrored half waves, each
// with its own products against K0 and its own moduli sums); summed per NEIGHBOUR instead -- the moduli of the elements that share
// a neighbour are combined first (sums and differences over the sides: l1_merged_core.h, the level-1 arithmetic with a single
// mirror class) -- a node costs 26 x 9 + 63 + 18 multiply-adds and ~100 additions = ~400 operations.  A wave relaxes TWO rows of
// the colour (lanes 0-31 / 32-63), four compute waves instead of seven, one per SIMD.
template <int ALT, int QF, int QM, int FORM>
__global__ void __launch_bounds__(64 * (gsm::compute_waves(FORM) + 1)) k_gs_march_mf0(GsMarchArgs A) {
    using namespace gsm;
    constexpr int CWv = compute_waves(FORM);
    const int P = A.forward ? 0 : 1;              // parity of the first in-plane colour: 0 forward colour order, 1 reverse
    extern __shared__ __align__(16) unsigned char smem[];
    double *sU = reinterpret_cast<double *>(smem);
    double *sE = sU + NU * U_SLOT_D;
    const Dims &d = A.d;
    const int lane = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.y);

    // relaxed planes x = first_plane + 2 mm, mm in [m0, m1)
    const int M = A.num_planes;
    const int m0 = blockIdx.x * A.steps_per_chunk;
    if (m0 >= M) return;                                              // block-uniform, before any barrier
    const int m1 = m0 + A.steps_per_chunk < M ? m0 + A.steps_per_chunk : M;
    const int nsteps = m1 - m0;
    const int x0 = A.first_plane + 2 * m0;
    // plane stream of this block: j = 0, 1, 2, ... <-> planes x0 - 1 + j; even j: fixed planes (uO), odd j: relaxed planes (uR).
    // a plane outside the grid is replaced by the nearest one of its parity (its values only ever meet elements outside the
    // grid, whose modulus is taken as 0)
    auto plane_of = [&](int j) { int i = x0 - 1 + j; if (i < 0) i += 2; if (i > d.NX - 1) i -= 2; return i; };

    const int yb = 2 * R * (int) blockIdx.z - P, zb = 2 * C * (int) blockIdx.y - P;     // first owned row / column (parity P)
    const int yl = yb - 1, zl = zb - 3;                                 // node row / column of staged index 0
    const int k0u = zl < -1 ? -1 : zl;                                  // first node column held by a staged node row
    const int cshift = k0u - zl;
    const int ek0u = k0u;                                               // first element column held by a staged element row
    const int ecshift = cshift;

    const long long plane = (long long) d.NY * d.NZ, elayer = (long long) d.ny * d.nz;
    const int ppar = (int) ((3 * plane) & 1), epar = (int) (elayer & 1);
    const int bparR = (int) ((reinterpret_cast<uintptr_t>(A.uR) >> 3) & 1), bparO = (int) ((reinterpret_cast<uintptr_t>(A.uO) >> 3) & 1);
    const int bparE = (int) ((reinterpret_cast<uintptr_t>(A.E) >> 3) & 1);
    auto row_par = [&](int ry) {                                        // parity of the first double of staged node row ry (before base / plane)
        int jj = yl + ry; jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
        return ((jj & d.NZ) ^ k0u) & 1;                                 // (3 (jj NZ + k0u)) & 1
    };

    // a tile whose staged element rows / columns reach beyond the grid: the compute waves zero those moduli in LDS after they land
    const bool fix_yz = yl < 0 || yl + EY - 1 > d.ny - 1 || zl < 0 || zl + EZ - 1 > d.nz - 1;

    // =========================== DMA wave ===========================
    if (wave == CWv) {
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
            const long long rs = (long long) jj * d.nz + ek0u;       // >= -1
            ego[t] = (unsigned) ((rs + 1 + 2LL * c) * 2 + (rs & 1));   // biased by +1 double
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
            const double *pb = A.E + (long long) il * elayer - 1;     // (the bias of ego)
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
            if (fix_yz || x0 + 2 * m - 1 < 0 || x0 + 2 * m > d.nx - 1) __builtin_amdgcn_s_barrier();   // (the compute waves zero moduli outside the grid)
            __builtin_amdgcn_s_barrier();                               // B1
        }
        return;
    }

    // =========================== compute waves ===========================
    const int h = lane >> 5, cl = lane & 31;                            // half (0: low side, 1: high side, computed on mirrored data), column index
    const int hq = h, clq = cl;
    const unsigned long long sgn = h ? 0x8000000000000000ull : 0ull;
    const int z0s = zb < 0 ? 0 : zb;                                    // owned columns inside the grid: [z0s, z1s]
    const int z1s = zb + 2 * C - 1 > d.NZ - 1 ? d.NZ - 1 : zb + 2 * C - 1;
    const int nd_store = 3 * (z1s - z0s + 1);
    const bool stamping = A.stamps && blockIdx.x == 0 && blockIdx.y == 1 && blockIdx.z == 1;
    auto stamp = [&](int m, int slot) {
        if (stamping && m < 8) {
            const long long t = gsm_now();
            if (lane == 0) A.stamps[(wave * 8 + m) * 16 + slot] = t;
        }
    };
    // right-hand side and solve data (inverse diagonal with the Dirichlet mask folded in, strict lower part of the node's
    // diagonal block; k_gs_solve_data) of the node this lane relaxes in colour k of plane xx, requested one colour ahead of
    // their use (bd[k & 1])
    double bd[2][9];
    auto request = [&](auto kc, int xx) {
        if constexpr (FORM == 2) return;                                // (form 2 requests at the start of the colour itself)
        constexpr int k = decltype(kc)::value, ro = k >> 1;
        constexpr int nrows = R + 1 - ro, ncols = C + 3 - k;
        const int rwq = FORM == 2 ? 2 * wave + h : wave;                   // row of the colour this lane works on
        const int y = yl + 1 + ro + 2 * rwq, z = zl + 1 + k + 2 * cl;
        const bool mine = (FORM == 2 || h == 0) && rwq < nrows && cl < ncols && y >= 0 && y < d.NY && z >= 0 && z < d.NZ;
#pragma unroll
        for (int q = 0; q < 9; ++q) bd[k & 1][q] = 0.0;
        if (mine) {
            const long long n = ((long long) xx * d.NY + y) * d.NZ + z;
#pragma unroll
            for (int q = 0; q < 3; ++q) bd[k & 1][q] = A.b[3 * n + q];
#pragma unroll
            for (int q = 0; q < 6; ++q) bd[k & 1][3 + q] = A.sd[6 * n + q];
        }
    };
    request(std::integral_constant<int, 0>{}, x0);

    for (int m = 0; m < nsteps; ++m) {
        const int x = x0 + 2 * m;
        stamp(m, 0);
        __builtin_amdgcn_s_barrier();                                   // B0
        stamp(m, 1);
        const int midoff = ((2 * m + 1) % NU) * U_SLOT_D;
        const int faroff = ((2 * m + 2 * h) % NU) * U_SLOT_D;           // per lane: plane x-1 (h = 0) or x+1 (h = 1)
        const int shM = bparR + (x & ppar);                             // + row parity = alignment shift of the staged rows
        const int shF = bparO + (plane_of(2 * m) & ppar);               // (planes x-1 and x+1 have the same parity)
        // element layers x-1 (ring slot of stream index 2m) and x (2m + 1)
        const bool lay_ok[2] = {x - 1 >= 0, x <= d.nx - 1};
        const int eoffs[2] = {((2 * m) % NE) * E_SLOT_D, ((2 * m + 1) % NE) * E_SLOT_D};
        const int eshl[2] = {bparE + ((x - 1 < 0 ? 0 : x - 1) & epar), bparE + ((x > d.nx - 1 ? d.nx - 1 : x) & epar)};
        auto erow_shift = [&](int layer, int r) {                       // alignment shift of staged element row r of a layer
            int jj = yl + r; jj = jj < 0 ? 0 : (jj > d.ny - 1 ? d.ny - 1 : jj);
            return (eshl[layer] + (((jj & d.nz) ^ ek0u) & 1)) & 1;
        };
        if (fix_yz || !lay_ok[0] || !lay_ok[1]) {
            // moduli of elements outside the grid are zero: written over whatever the (clamped) DMA brought, once per layer
            for (int q = threadIdx.y * 64 + lane; q < 2 * EY * (EZ + 1); q += CWv * 64) {
                const int layer = q / (EY * (EZ + 1)), q2 = q - layer * (EY * (EZ + 1));
                const int r = q2 / (EZ + 1), ci = q2 - r * (EZ + 1);
                const int ey = yl + r, ez = ek0u + ci;
                if (!lay_ok[layer] || ey < 0 || ey > d.ny - 1 || ez < 0 || ez > d.nz - 1) sE[eoffs[layer] + r * EROW_D + ci + erow_shift(layer, r)] = 0.0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }

        // ---- form 2: one node per lane, rows 2 wave + h of the colour ----
        auto phase2 = [&](auto kc) {
            constexpr int k = decltype(kc)::value, ro = k >> 1;
            constexpr int nrows = R + 1 - ro, ncols = C + 3 - k;
            // (the row / column arithmetic below does not depend on the step: the compiler hoists it out of the march for all four colours,
            // ~45 registers held throughout -- affordable at one wave per SIMD, and 4 % faster than redoing it per step)
            const int h = hq, cl = clq;
            const int rw = 2 * wave + h;
            const int rwe = rw < nrows ? rw : nrows - 1;                // half waves beyond the colour's rows shadow the last one
            const int ry = 1 + ro + 2 * rwe;
            const int y = yl + ry;
            const int ce = cl < ncols ? cl : ncols - 1;
            const int czn = 1 + k + 2 * ce;
            const int z = zl + czn;
            const bool mine = rw < nrows && cl < ncols && y >= 0 && y < d.NY && z >= 0 && z < d.NZ;
            // right-hand side and inverse diagonal of the node (the rest of the diagonal block is formed below): requested now, used after
            // the ~400 operations of the colour.  Buffer loads on the relaxed plane: lanes that relax nothing (tile halo, shadow rows) get an
            // out-of-range offset and move no data -- the sweep is close enough to the HBM rate (14 GB per half sweep at 512^3 in 3.7 ms,
            // block lives of 180 us for ten 9.5 us steps of arithmetic) for 72 bytes per idle lane to matter
            double B[3], D[3];
            {
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.b + 3LL * x * plane), 0, (int) (24 * plane), 0x00020000);
                const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.sd + 3LL * x * plane), 0, (int) (24 * plane), 0x00020000);
                const unsigned off = mine ? (unsigned) (y * d.NZ + z) * 24u : 0x7ffffff0u;
                const u4q_t vb = __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 0), vd = __builtin_amdgcn_raw_buffer_load_b128(rd, off, 0, 0);
                const u2q_t wb = __builtin_amdgcn_raw_buffer_load_b64(rb, off + 16, 0, 0), wd = __builtin_amdgcn_raw_buffer_load_b64(rd, off + 16, 0, 0);
                B[0] = mkd(vb.x, vb.y); B[1] = mkd(vb.z, vb.w); B[2] = mkd(wb.x, wb.y);
                D[0] = mkd(vd.x, vd.y); D[1] = mkd(vd.z, vd.w); D[2] = mkd(wd.x, wd.y);
            }
            int ni = czn - 1 - cshift;
            ni = max(ni, ni & 1);
            const int rlo = y - 1 >= 0 ? ry - 1 : ry + 1, rhi = y + 1 <= d.NY - 1 ? ry + 1 : ry - 1;
            constexpr int OFC = (QF + k + ALT * ro) & 1, OFS = (QF + k + ALT * (ro + 1)) & 1;
            constexpr int OMC = (QM + k + ALT * ro) & 1, OMS = (QM + k + ALT * (ro + 1)) & 1;
            const int lowoff = ((2 * m) % NU) * U_SLOT_D, highoff = ((2 * m + 2) % NU) * U_SLOT_D;
            const int shc = (shF + row_par(ry)) & 1, shs = (shF + row_par(rlo)) & 1;
            const int mhc = (shM + row_par(ry)) & 1, mhs = (shM + row_par(rlo)) & 1;
            // the three rows of a plane as the 3 x 3 window of l1m::side_class / mid_class
            auto window = [&](int slotoff, int sh_c, int sh_s, auto offc, auto offs, double (&un)[3][9]) {
                constexpr int oc = decltype(offc)::value, os = decltype(offs)::value;
                double w0[10], w1[10], w2[10];
                read10(sU, slotoff + rlo * ROW_D + 3 * ni + sh_s - os, w0);
                read10(sU, slotoff + ry * ROW_D + 3 * ni + sh_c - oc, w1);
                read10(sU, slotoff + rhi * ROW_D + 3 * ni + sh_s - os, w2);
#pragma unroll
                for (int c = 0; c < 9; ++c) { un[0][c] = w0[os + c]; un[1][c] = w1[oc + c]; un[2][c] = w2[os + c]; }
            };
            // moduli of the eight incident elements (layer, y - 1 + dj, z - 1 + dk); zero outside the grid (fixed up in LDS above)
            const int ecol = max(czn - 1 - ecshift, 0);
            double a0[2][2], a1[2][2];
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const int r = ry - 1 + dj;
                const int e0 = eoffs[0] + r * EROW_D + ecol + erow_shift(0, r), e1 = eoffs[1] + r * EROW_D + ecol + erow_shift(1, r);
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) { a0[dj][dk] = sE[e0 + dk]; a1[dj][dk] = sE[e1 + dk]; }
            }
            // Software pipeline over the three planes, fenced by scheduling barriers: the windows of the next plane are read while the
            // current one is multiplied, and no more (left alone the scheduler issues all nine rows' reads and all twelve coefficient
            // rows up front: 180 registers of windows, 600 spilled scalars).  Coefficient rows come through the scalar cache one row
            // ahead (coef_rows.h); as LDS reads at a wave-uniform address they were slower (a colour 2.4-3.4 us against 1.8).
            double S[3] = {0.0, 0.0, 0.0}, M6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, uself[3];
            double unA[3][9], unB[3][9];
            __builtin_amdgcn_sched_barrier(0);
            window(lowoff, shc, shs, std::integral_constant<int, OFC>{}, std::integral_constant<int, OFS>{}, unA);
            RowPipe<12, L0NodeRows> rows{A.tab};
            asm volatile("" : "+v"(S[0]) : "v"(a0[0][0]));               // (the chain of row waits starts behind the moduli reads)
            rows.prime();
            __builtin_amdgcn_sched_barrier(0);
            // the next plane's windows are requested right BEHIND the wait for a part's first coefficient row: a row wait is
            // lgkmcnt(0) and would drain them (requested in front of it, every part paid the full LDS latency: waves 44 % of their
            // time in s_waitcnt, gpurun_out/r03_gsm_pmc3)
            {
                auto cf = l0_coef<false, 0>(rows, S[0], [&] {
                    window(midoff, mhc, mhs, std::integral_constant<int, OMC>{}, std::integral_constant<int, OMS>{}, unB);
                });
                l1m::side_class<0, 0>(a0, unA, cf, S);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                uself[0] = unB[1][3]; uself[1] = unB[1][4]; uself[2] = unB[1][5];
                auto cf = l0_coef<true, 4>(rows, S[0], [&] {
                    window(highoff, shc, shs, std::integral_constant<int, OFC>{}, std::integral_constant<int, OFS>{}, unA);
                });
                l1m::mid_class<0>(a0, a1, unB, cf, S, M6);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                auto cf = l0_coef<false, 8>(rows, S[0], [] {});
                l1m::side_class<1, 0>(a1, unA, cf, S);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // residual form of m_smoothNode (MG.hh:199-264): S also takes the node's own block; component-sequential solve with the
                // stored inverse diagonal (0 for a fixed component) and the strict lower part of the block just formed (M6: xx xy xz yy yz zz)
                const double s0 = S[0] + (M6[0] * uself[0] + M6[1] * uself[1] + M6[2] * uself[2]);
                const double s1 = S[1] + (M6[1] * uself[0] + M6[3] * uself[1] + M6[4] * uself[2]);
                const double s2 = S[2] + (M6[2] * uself[0] + M6[4] * uself[1] + M6[5] * uself[2]);
                const double b0 = B[0] - s0, b1 = B[1] - s1, b2 = B[2] - s2;
                double ud0, ud1, ud2;
                if (A.forward) {
                    ud0 = b0 * D[0];
                    ud1 = (b1 - M6[1] * ud0) * D[1];
                    ud2 = (b2 - (M6[2] * ud0 + M6[4] * ud1)) * D[2];
                } else {
                    ud2 = b2 * D[2];
                    ud1 = (b1 - M6[4] * ud2) * D[1];
                    ud0 = (b0 - (M6[1] * ud1 + M6[2] * ud2)) * D[0];
                }
                if (mine) {
                    const int iself = midoff + ry * ROW_D + 3 * (czn - cshift) + mhc;
                    sU[iself] = uself[0] + ud0; sU[iself + 1] = uself[1] + ud1; sU[iself + 2] = uself[2] + ud2;
                }
            }
        };

        auto phase = [&](auto kc) {
            if constexpr (FORM == 2) { phase2(kc); return; }
            constexpr int k = decltype(kc)::value, ro = k >> 1;
            constexpr int nrows = R + 1 - ro, ncols = C + 3 - k;
            if (wave >= nrows) return;
            const int ry = 1 + ro + 2 * wave;                           // staged row of the node
            const int y = yl + ry;
            const int ce = cl < ncols ? cl : ncols - 1;                 // lanes beyond the colour's columns shadow the last one
            const int czn = 1 + k + 2 * ce;                             // staged column
            const int z = zl + czn;
            const bool mine = h == 0 && cl < ncols && y >= 0 && y < d.NY && z >= 0 && z < d.NZ;
            int ni = czn - 1 - cshift;                                  // staged node index of the window's first node (z - 1)
            ni = max(ni, ni & 1);                                       // (outside the grid: any window of the right alignment)
            // Rows read: the far plane's rows y-1, y, y+1; the node's own plane: row y and the row on this half's side.
            // A row outside the grid (only ever multiplied by zero moduli) is replaced by its mirror image on the other side,
            // which has the same parity
            const int rlo = y - 1 >= 0 ? ry - 1 : ry + 1, rhi = y + 1 <= d.NY - 1 ? ry + 1 : ry - 1;
            const int rside = h ? rhi : rlo;
            constexpr int OFC = (QF + k + ALT * ro) & 1, OFS = (QF + k + ALT * (ro + 1)) & 1;      // window offsets: far row y, far rows y -+ 1
            constexpr int OMC = (QM + k + ALT * ro) & 1, OMS = (QM + k + ALT * (ro + 1)) & 1;      // own plane: row y, side row
            double wf[3][10], wm[2][10];
            {
                const int shc = (shF + row_par(ry)) & 1, shs = (shF + row_par(rlo)) & 1;
                read10(sU, faroff + rlo * ROW_D + 3 * ni + shs - OFS, wf[0]);
                read10(sU, faroff + ry * ROW_D + 3 * ni + shc - OFC, wf[1]);
                read10(sU, faroff + rhi * ROW_D + 3 * ni + shs - OFS, wf[2]);
                const int mhc = (shM + row_par(ry)) & 1, mhs = (shM + row_par(rlo)) & 1;
                read10(sU, midoff + rside * ROW_D + 3 * ni + mhs - OMS, wm[0]);
                read10(sU, midoff + ry * ROW_D + 3 * ni + mhc - OMC, wm[1]);
            }
            // moduli of the elements (layer, y - 1 + dj, z - 1 + dk); zero outside the grid (fixed up in LDS above)
            const int ecol = max(czn - 1 - ecshift, 0);
            // layer of this half for the four far slots; both layers for the element row dj = h of the node's own plane
            // (per-lane layer: the ring offset and the alignment shift are selected, not the data)
            const int eo_mine = h ? eoffs[1] : eoffs[0], eo_other = h ? eoffs[0] : eoffs[1];
            double ea[2][2], eb[2];
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const int r = ry - 1 + dj;
                const int s0 = erow_shift(0, r), s1 = erow_shift(1, r);
                const int rowoff = r * EROW_D + ecol;
                const int sm = h ? s1 : s0;
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) ea[dj][dk] = sE[eo_mine + rowoff + dk + sm];
            }
            {
                const int r = ry - 1 + h;
                int jj = yl + r; jj = jj < 0 ? 0 : (jj > d.ny - 1 ? d.ny - 1 : jj);
                const int so = ((h ? eshl[0] : eshl[1]) + (((jj & d.nz) ^ ek0u) & 1)) & 1;
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) eb[dk] = sE[eo_other + r * EROW_D + ecol + dk + so];
            }
            double ep[2], em[2];
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                const double own = h ? ea[1][dk] : ea[0][dk];           // this half's layer of element row y - 1 + h
                // layers x-1, x:  h = 0: (own, other);  h = 1: (other, own)  ->  sum is symmetric, the difference changes sign with h
                ep[dk] = own + eb[dk];
                em[dk] = flip(own - eb[dk], sgn);
            }
            // the node's own value (used by the lanes h = 0, where nothing is mirrored)
            const double uself[3] = {wm[1][OMC + 3], wm[1][OMC + 4], wm[1][OMC + 5]};

            // ---- far plane: the four element slots on this half's side in x (x-mirrored for h = 1) ----
            GsCoef24 ckF;
            gs_load_coef24(A.coef + 36, ckF);                           // (waits for the LDS reads above as well)
            double T[4][3];                                             // (the first term of every accumulator is a plain product)
            static_for<3>([&](auto dc) {
                constexpr int dy = decltype(dc)::value - 1, off = dy == 0 ? OFC : OFS;
                const double *w = wf[dy + 1];
                const double v[9] = {flip(w[off], sgn), w[off + 1], w[off + 2], flip(w[off + 3], sgn), w[off + 4], w[off + 5],
                                     flip(w[off + 6], sgn), w[off + 7], w[off + 8]};
                static_for<4>([&](auto ec) {
                    constexpr int dj = decltype(ec)::value >> 1, my = decltype(ec)::value & 1;
                    if constexpr (dj - 1 + my == dy) {
                        static_for<4>([&](auto zc) {
                            constexpr int dk = decltype(zc)::value >> 1, mz = decltype(zc)::value & 1;
                            constexpr int n3 = dk + mz;
                            constexpr int ln = 4 + 2 * (1 - dj) + (1 - dk), lm = 2 * my + mz;
                            static_for<9>([&](auto qc) {
                                constexpr int r = decltype(qc)::value / 3, c = decltype(qc)::value % 3;
                                if constexpr (my == 0 && mz == 0 && c == 0) T[2 * dj + dk][r] = gs_coef24_at<0, ln, r, lm, c>(ckF) * v[3 * n3 + c];
                                else T[2 * dj + dk][r] = fma(gs_coef24_at<0, ln, r, lm, c>(ckF), v[3 * n3 + c], T[2 * dj + dk][r]);
                            });
                        });
                    }
                });
            });
            // ---- the node's own plane: the element row on this half's side in y (y-mirrored for h = 1), both layers at once.
            // K0[(n ^ 4, a), (m ^ 4, b)] = s_a s_b K0[(n, a), (m, b)] with s_x = -1: the entries with exactly one x index change
            // sign between the two layers ("odd", weighted by E(x-1) - E(x)), the others do not ("even", E(x-1) + E(x)) ----
            GsCoef24 ckM;
            gs_load_coef24(A.coef + 60, ckM);
            double Te[2][3], To[2][3];
            static_for<2>([&](auto mc) {
                constexpr int my = decltype(mc)::value, off = my ? OMC : OMS;
                const double *w = wm[my];
                const double v[9] = {w[off], flip(w[off + 1], sgn), w[off + 2], w[off + 3], flip(w[off + 4], sgn), w[off + 5],
                                     w[off + 6], flip(w[off + 7], sgn), w[off + 8]};
                static_for<4>([&](auto zc) {
                    constexpr int dk = decltype(zc)::value >> 1, mz = decltype(zc)::value & 1;
                    constexpr int n3 = dk + mz;
                    constexpr int ln = 4 + 2 + (1 - dk), lm = 4 + 2 * my + mz;
                    static_for<9>([&](auto qc) {
                        constexpr int r = decltype(qc)::value / 3, c = decltype(qc)::value % 3;
                        constexpr bool even = (r == 0) == (c == 0);
                        constexpr bool first = my == 0 && mz == 0 && c == (even ? (r == 0 ? 0 : 1) : (r == 0 ? 1 : 0));      // first entry of its class in the row
                        if constexpr (even) {
                            if constexpr (first) Te[dk][r] = gs_coef24_at<1, ln, r, lm, c>(ckM) * v[3 * n3 + c];
                            else Te[dk][r] = fma(gs_coef24_at<1, ln, r, lm, c>(ckM), v[3 * n3 + c], Te[dk][r]);
                        } else {
                            if constexpr (first) To[dk][r] = gs_coef24_at<1, ln, r, lm, c>(ckM) * v[3 * n3 + c];
                            else To[dk][r] = fma(gs_coef24_at<1, ln, r, lm, c>(ckM), v[3 * n3 + c], To[dk][r]);
                        }
                    });
                });
            });
            // partial sums of this half in its mirrored frames, then back to the node's frame
            double Sf[3] = {0.0, 0.0, 0.0}, Sm[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int r = 0; r < 3; ++r) Sf[r] = fma(ea[sl >> 1][sl & 1], T[sl][r], Sf[r]);
#pragma unroll
            for (int dk = 0; dk < 2; ++dk)
#pragma unroll
                for (int r = 0; r < 3; ++r) { Sm[r] = fma(ep[dk], Te[dk][r], Sm[r]); Sm[r] = fma(em[dk], To[dk][r], Sm[r]); }
            double Sp[3] = {flip(Sf[0], sgn) + Sm[0], Sf[1] + flip(Sm[1], sgn), Sf[2] + Sm[2]};
#pragma unroll
            for (int q = 0; q < 3; ++q) Sp[q] += __shfl_down(Sp[q], 32, 64);
            if (mine) {
                // component-sequential solve of MG.hh:254-264 with the stored inverse diagonal (0 for a fixed component)
                const double *B = bd[k & 1], *D = bd[k & 1] + 3;        // D: i00 i11 i22 m10 m20 m21
                const double b0 = B[0] - Sp[0], b1 = B[1] - Sp[1], b2 = B[2] - Sp[2];
                double ud0, ud1, ud2;
                if (A.forward) {
                    ud0 = b0 * D[0];
                    ud1 = (b1 - D[3] * ud0) * D[1];
                    ud2 = (b2 - (D[4] * ud0 + D[5] * ud1)) * D[2];
                } else {
                    ud2 = b2 * D[2];
                    ud1 = (b1 - D[5] * ud2) * D[1];
                    ud0 = (b0 - (D[3] * ud1 + D[4] * ud2)) * D[0];
                }
                const int iself = midoff + ry * ROW_D + 3 * (czn - cshift) + ((shM + row_par(ry)) & 1);
                sU[iself] = uself[0] + ud0; sU[iself + 1] = uself[1] + ud1; sU[iself + 2] = uself[2] + ud2;
            }
        };
        // a finished row of this wave (owned rows only), dense 8-byte stores
        auto store_row = [&](int ry) {
            const int y = yl + ry;
            if (ry < 1 || ry > 2 * R || y < 0 || y > d.NY - 1) return;
            const double *src = sU + midoff + ry * ROW_D + 3 * (z0s - k0u) + ((shM + row_par(ry)) & 1);
            double *dp = A.dst + 3 * (((long long) x * d.NY + y) * d.NZ + z0s);
            for (int i = lane; i < nd_store; i += 64) dp[i] = src[i];
        };
        // two finished rows of a wave (form 2): all LDS reads first, then the stores (3 x 58 doubles per row: three 8-byte pieces per lane)
        auto store_rows2 = [&](int ry0) {
            double v[2][3];
            bool ok[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ry = ry0 + 2 * q, y = yl + ry;
                ok[q] = ry >= 1 && ry <= 2 * R && y >= 0 && y <= d.NY - 1;
                const double *src = sU + midoff + (ok[q] ? ry : 1) * ROW_D + 3 * (z0s - k0u) + ((shM + row_par(ok[q] ? ry : 1)) & 1);
#pragma unroll
                for (int t = 0; t < 3; ++t) v[q][t] = (ok[q] && lane + 64 * t < nd_store) ? src[lane + 64 * t] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int y = yl + ry0 + 2 * q;
                double *dp = A.dst + 3 * (((long long) x * d.NY + (ok[q] ? y : 0)) * d.NZ + z0s);
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (ok[q] && lane + 64 * t < nd_store) dp[lane + 64 * t] = v[q][t];
            }
        };
        // rows of parity P: colours (P,P) then (P,Q); the second reads the first's updates of its own row only, so the two are
        // ordered inside the wave (its LDS accesses execute in order) and need no workgroup barrier
        request(std::integral_constant<int, 1>{}, x);
        phase(std::integral_constant<int, 0>{});
        __builtin_amdgcn_wave_barrier();
        stamp(m, 2);
        request(std::integral_constant<int, 2>{}, x);
        phase(std::integral_constant<int, 1>{});
        __builtin_amdgcn_wave_barrier();
        stamp(m, 3);
        if (FORM == 2) store_rows2(1 + 4 * wave);
        else store_row(1 + 2 * wave);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(m, 4);
        __builtin_amdgcn_s_barrier();                                   // B1: the rows of parity P are final
        stamp(m, 5);
        request(std::integral_constant<int, 3>{}, x);
        phase(std::integral_constant<int, 2>{});
        __builtin_amdgcn_wave_barrier();
        stamp(m, 6);
        if (m + 1 < nsteps) request(std::integral_constant<int, 0>{}, x + 2);
        phase(std::integral_constant<int, 3>{});
        __builtin_amdgcn_wave_barrier();
        if (FORM == 2) store_rows2(2 + 4 * wave);
        else if (wave < R) store_row(2 + 2 * wave);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(m, 7);
    }
}

// Solve data of the level-0 sweeps, once per operator update: per node the inverse diagonal of its 3x3 diagonal block
// M = sum_e E_e K0[n-block] (MG.hh:199-220) with the Dirichlet mask folded in (0 for a fixed component, MG.hh:258-262) and the
// strict lower part of M (it is symmetric): sd[n] = { 1/M00, 1/M11, 1/M22, M10, M20, M21 }
// (per = 3: the inverse diagonal only -- form 2 of the marching sweep forms the diagonal block itself)
__global__ void __launch_bounds__(256) k_gs_solve_data(Dims d, const double *__restrict__ K0, const double *__restrict__ E,
                                                       const uint8_t *__restrict__ mask, double *__restrict__ sd, int per) {
    const long long n = (long long) blockIdx.x * 256 + threadIdx.x;
    if (n >= d.nn) return;
    const int k = (int) (n % d.NZ), j = (int) ((n / d.NZ) % d.NY), i = (int) (n / ((long long) d.NZ * d.NY));
    double M[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};           // 00 11 22 10 20 21
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
        const int ex = i - 1 + ((sl >> 2) & 1), ey = j - 1 + ((sl >> 1) & 1), ez = k - 1 + (sl & 1);
        if (ex < 0 || ex >= d.nx || ey < 0 || ey >= d.ny || ez < 0 || ez >= d.nz) continue;
        const double Ee = E[((long long) ex * d.ny + ey) * d.nz + ez];
        const double *blk = K0 + (3 * (7 - sl)) * 24 + 3 * (7 - sl);
        M[0] = fma(Ee, blk[0], M[0]); M[1] = fma(Ee, blk[24 + 1], M[1]); M[2] = fma(Ee, blk[48 + 2], M[2]);
        M[3] = fma(Ee, blk[24], M[3]); M[4] = fma(Ee, blk[48], M[4]); M[5] = fma(Ee, blk[48 + 1], M[5]);
    }
    const uint8_t mk = mask[n];
    sd[per * n + 0] = (mk & 1) ? 0.0 : 1.0 / M[0];
    sd[per * n + 1] = (mk & 2) ? 0.0 : 1.0 / M[1];
    sd[per * n + 2] = (mk & 4) ? 0.0 : 1.0 / M[2];
    if (per == 6) { sd[6 * n + 3] = M[3]; sd[6 * n + 4] = M[4]; sd[6 * n + 5] = M[5]; }
}
void launch_gs_solve_data(const Dims &d, const double *K0, const double *E, const uint8_t *mask, double *sd, hipStream_t s, int per) {
    k_gs_solve_data<<<dim3((unsigned) ((d.nn + 255) / 256)), 256, 0, s>>>(d, K0, E, mask, sd, per);
    VFEM_HIP(hipGetLastError());
}

long long *g_gsm_stamps = nullptr;       // diagnostic: device buffer of 7 x 8 x 8 stamps (vfem_debug_gsm_stamps), null in production

// planes of local parity `par` copied from src to dst (the odd sweep left them in the scratch vector)
__global__ void __launch_bounds__(256) k_copy_planes(Dims d, int first, int last, const double *__restrict__ src, double *__restrict__ dst) {
    const long long per = 3LL * d.NY * d.NZ;
    const int i = first + 2 * blockIdx.y;
    if (i > last) return;
    for (long long q = (long long) blockIdx.x * 256 + threadIdx.x; q < per; q += (long long) gridDim.x * 256) dst[i * per + q] = src[i * per + q];
}
void launch_copy_planes(const Dims &d, int par, const double *src, double *dst, hipStream_t s, int plane_lo, int plane_hi) {
    const long long per = 3LL * d.NY * d.NZ;
    if (plane_hi < 0 || plane_hi > d.NX - 1) plane_hi = d.NX - 1;
    if (plane_lo < 0) plane_lo = 0;
    const int first = plane_lo + (((plane_lo & 1) != (par & 1)) ? 1 : 0);
    if (first > plane_hi) return;
    unsigned gx = (unsigned) ((per + 255) / 256);
    if (gx > 64) gx = 64;
    k_copy_planes<<<dim3(gx, (unsigned) ((plane_hi - first) / 2 + 1)), 256, 0, s>>>(d, first, plane_hi, src, dst);
    VFEM_HIP(hipGetLastError());
}

// One half sweep (the four colours of one x parity) of the level-0 Gauss-Seidel.  forward: colour order (0,0),(0,1),(1,0),(1,1)
// and components 0,1,2; otherwise the reverse of both.  Reads the relaxed planes from uR and the others from uO, writes the
// relaxed planes to dst (must differ from uR).  Returns false when the kernel cannot run on these buffers.
bool launch_gs_march_mf0(const Dims &d, const double *coef36, const double *E, const double *E_alloc_begin, const double *E_alloc_end,
                         const double *uR, const double *uO, double *dst, const double *b, const double *solve_data,
                         int cxl, int forward, int chunks, hipStream_t s, int plane_lo, int plane_hi, const double *tab_form2, int form) {
    using namespace gsm;
    if (!tab_form2 || form != 2) form = 1;
    if (dst == uR) return false;
    if ((reinterpret_cast<uintptr_t>(uR) & 7u) || (reinterpret_cast<uintptr_t>(uO) & 7u) || (reinterpret_cast<uintptr_t>(E) & 7u)) return false;
    if (d.NX < 2 || d.NY < 2 || d.NZ < 2) return false;
    if (plane_hi < 0 || plane_hi > d.NX - 1) plane_hi = d.NX - 1;
    if (plane_lo < 0) plane_lo = 0;
    const int first_plane = plane_lo + (((plane_lo & 1) != (cxl & 1)) ? 1 : 0);
    if (first_plane > plane_hi) return true;                          // no plane of this parity in the range
    GsMarchArgs a;
    a.d = d;
    a.coef = coef36;
    a.tab = tab_form2;
    a.E = E;
    auto first_piece = [](const void *p) { return reinterpret_cast<const char *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t) 15); };
    auto last_piece = [](const void *end) { return reinterpret_cast<const char *>((reinterpret_cast<uintptr_t>(end) - 1) & ~(uintptr_t) 15); };
    a.e_first = first_piece(E_alloc_begin); a.e_last = last_piece(E_alloc_end);
    a.uR = uR; a.uO = uO;
    a.uR_first = first_piece(uR); a.uR_last = last_piece(uR + 3 * d.nn);
    a.uO_first = first_piece(uO); a.uO_last = last_piece(uO + 3 * d.nn);
    a.dst = dst; a.b = b; a.sd = solve_data;
    a.cxl = cxl; a.forward = forward;
    a.stamps = g_gsm_stamps;
    const int M = (plane_hi - first_plane) / 2 + 1;
    a.first_plane = first_plane; a.num_planes = M;
    const int P = forward ? 0 : 1;
    const int nty = (d.NY + P + 2 * R - 1) / (2 * R), ntz = (d.NZ + P + 2 * C - 1) / (2 * C);
    if (chunks <= 0) {
        // many more blocks than CUs (one block per CU is resident: short blocks even out the tail), chunks of at least 8 steps;
        // measured at 512^3 (387 tiles): 4 chunks 8.7 ms per sweep, 9: 7.9, 13-26: 7.7-7.8 (tools/gs_march_probe.py)
        chunks = 1;
        while ((long long) chunks * nty * ntz < 5000 && M / (chunks + 1) >= 8) ++chunks;
    }
    if (chunks > M) chunks = M;
    a.steps_per_chunk = (M + chunks - 1) / chunks;
    const unsigned gx = (unsigned) ((M + a.steps_per_chunk - 1) / a.steps_per_chunk);
    const dim3 grd(gx, (unsigned) ntz, (unsigned) nty), blk(64, compute_waves(form) + 1, 1);
    // launch-uniform window parities (see the kernel's comment)
    const long long plane = (long long) d.NY * d.NZ;
    const int ppar = (int) ((3 * plane) & 1), ALT = d.NZ & 1;
    const int bparR = (int) ((reinterpret_cast<uintptr_t>(uR) >> 3) & 1), bparO = (int) ((reinterpret_cast<uintptr_t>(uO) >> 3) & 1);
    const int QF = ((P + 1) + bparO + ((cxl + 1) & ppar) + ALT * P) & 1, QM = ((P + 1) + bparR + (cxl & ppar) + ALT * P) & 1;
    static bool attr[16] = {false};
#define VFEM_GSM_LAUNCH(A_, F_, M_, V_)                                                                                           \
    do {                                                                                                                          \
        constexpr int v_ = (V_ - 1) * 8 + A_ * 4 + F_ * 2 + M_;                                                                   \
        if (!attr[v_]) {                                                                                                          \
            VFEM_HIP(hipFuncSetAttribute((const void *) k_gs_march_mf0<A_, F_, M_, V_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES)); \
            attr[v_] = true;                                                                                                      \
        }                                                                                                                         \
        k_gs_march_mf0<A_, F_, M_, V_><<<grd, blk, LDS_BYTES, s>>>(a);                                                          \
    } while (0)
#define VFEM_GSM_FORMS(A_, F_, M_) do { if (form == 2) VFEM_GSM_LAUNCH(A_, F_, M_, 2); else VFEM_GSM_LAUNCH(A_, F_, M_, 1); } while (0)
    switch (ALT * 4 + QF * 2 + QM) {
        case 0: VFEM_GSM_FORMS(0, 0, 0); break;
        case 1: VFEM_GSM_FORMS(0, 0, 1); break;
        case 2: VFEM_GSM_FORMS(0, 1, 0); break;
        case 3: VFEM_GSM_FORMS(0, 1, 1); break;
        case 4: VFEM_GSM_FORMS(1, 0, 0); break;
        case 5: VFEM_GSM_FORMS(1, 0, 1); break;
        case 6: VFEM_GSM_FORMS(1, 1, 0); break;
        default: VFEM_GSM_FORMS(1, 1, 1); break;
    }
#undef VFEM_GSM_FORMS
#undef VFEM_GSM_LAUNCH
    VFEM_HIP(hipGetLastError());
    return true;
}

}  // namespace vfem
