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
// * Staging.  Node planes are brought in by `global_load_lds_dwordx4` (one dedicated DMA wave; the idiom and the 16-byte-grid
//   addressing of kernels_apply_dma.hip): 5 node-plane slots (x-1, x, x+1 resident, x+2, x+3 in flight), 135 KB of LDS, one
//   workgroup of 4 compute waves + 1 DMA wave per CU.  The element moduli a node needs (2 layers x 2 x 2) are NOT staged (round 4):
//   they come by buffer loads one colour ahead of their use, out-of-grid elements as out-of-range offsets that read 0.  The 32 KB
//   their LDS ring took buy a seventh row pair per tile: a colour then has exactly the eight rows the four compute waves relax
//   (rounds 2-3: seven rows in eight slots), 37 instead of 43 tiles across 513 rows.
// * Tile seams.  A tile cannot see its neighbours' updates, so it recomputes what it needs of them: with the colour order
//   (p,p), (p,q), (q,p), (q,q) (p = 0 forward, 1 reverse; q = 1 - p) the nodes a tile OWNS are 2R rows x 2C columns starting at
//   a row / column of parity p; colour k is relaxed on  rows [yb + (k>>1), yb + 2R - (k>>1)]  and  columns [zb - 2 + k, zb + 2C +
//   2 - k]  of its parity, and one more ring of old values is loaded (17 x 65 staged node columns for 14 x 58 owned ones).
//   Recomputed nodes run the same instruction sequence on the same inputs in every tile, so they agree bit for bit.
// * Out of place.  Tiles read OLD halo values of their neighbours, so a half sweep must not overwrite its input: it reads the
//   relaxed parity from `uR`, the other parity from `uO` and writes the relaxed planes to `dst` (!= uR); the caller
//   ping-pongs between the field and one scratch vector (capi.hip: mg_smooth_n), two sweeps end where they began.
// * Arithmetic.  One node per LANE, a wave relaxes two rows of the colour (lanes 0-31 / 32-63).  The node row is summed per
//   NEIGHBOUR: the moduli of the elements that share a neighbour are combined first (sums and differences over the sides:
//   l1_merged_core.h, the level-1 arithmetic with a single mirror class), 26 x 9 + 63 + 18 multiply-adds and ~100 additions per
//   node; coefficient rows arrive through the scalar cache one row ahead (coef_rows.h).  (Rounds 2-3 also carried "form 1", a node
//   as two x-mirrored half waves with K0 in SGPRs: 8 % slower, removed in round 4.)
#include "vfem_internal.h"
#include "device_utils.h"
#include "gs_coef.h"
#include "coef_rows.h"

namespace vfem {

namespace gsm {
#ifndef VFEM_GSM_R
#define VFEM_GSM_R 7
#endif
constexpr int R = VFEM_GSM_R;                 // owned row pairs of a tile (2R owned node rows)
constexpr int C = 29;                         // owned column pairs (2C owned node columns); C + 3 = 32 lanes in the widest colour
constexpr int CW = (R + 2) / 2;               // compute waves (a wave = two rows of the colour, one node per lane)

constexpr int LY = 2 * R + 3, LZ = 2 * C + 7; // staged node rows / columns (17 x 65)
constexpr int PU = (LZ * 24 + 8 + 15) / 16;   // 16-byte pieces per staged node row incl. the alignment shift (98)
constexpr int ROW_D = 2 * PU;
constexpr int U_INSTR = (LY * PU + 63) / 64;  // DMA instructions per node plane (27)
constexpr int U_SLOT_D = U_INSTR * 128;       // doubles per slot
constexpr int NU = 5;
constexpr size_t LDS_BYTES = (size_t) (NU * U_SLOT_D) * 8;
static_assert(C + 3 == 32, "the widest colour fills a half wave");
static_assert(LDS_BYTES <= 160 * 1024, "ring must fit the LDS of a CU");
static_assert(2 * U_INSTR <= 63, "one step of DMA must fit the 6-bit vmcnt");
}  // namespace gsm

struct GsMarchArgs {
    Dims d;
    const double *tab;             // K0 by neighbour kind, signs folded in (l1m::build_table of K0, class 0: 8 rows of 12 doubles)
    const double *E;               // moduli of the level's elements, [nx][ny][nz]
    const double *uR, *uO;         // current values of the planes of the relaxed parity / of the other parity
    const char *uR_first, *uR_last, *uO_first, *uO_last;
    double *dst;                   // receives the relaxed planes (other planes untouched)
    const double *b;
    const double *sd;              // per node the inverse diagonal of its 3 x 3 block with the Dirichlet mask folded in (k_gs_solve_data, 3 doubles)
    int cxl;                       // local x parity of the relaxed planes
    int forward;                   // component order of the 3x3 solve (MG.hh:254-264)
    int steps_per_chunk;           // relaxed planes per block
    int first_plane, num_planes;   // relaxed planes: first_plane + 2 mm, mm in [0, num_planes)  (first_plane has the parity cxl)
};

typedef double d2a_t __attribute__((ext_vector_type(2), aligned(16)));
typedef unsigned int u4q_t __attribute__((ext_vector_type(4)));
typedef unsigned int u2q_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mkd(unsigned lo, unsigned hi) { return __longlong_as_double(((unsigned long long) hi << 32) | lo); }

__device__ __forceinline__ void gsm_glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                     (__attribute__((address_space(3))) void *) l, 16, 0, 0);
}
// The nine doubles of a node window start at an index whose parity depends on the row and the plane (the alignment shift of the
// staged image).  Ten doubles from the 16-byte-aligned index at or below it are read instead, always as five 16-byte reads in the
// same registers whatever the parity; the consumer then indexes w[off + j] with off = the parity, a compile-time constant.
// (Reading by parity -- 8 + 4 x 16 bytes or 4 x 16 + 8 -- left the two paths with different register layouts, which the compiler
// reconciled with moves behind a wait for the data: every row's latency was exposed.)
__device__ __forceinline__ void read10(const double *s, int idx_aligned, double w[10]) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const d2a_t a = *reinterpret_cast<const d2a_t *>(s + idx_aligned + 2 * q);
        w[2 * q] = a[0]; w[2 * q + 1] = a[1];
    }
}

// The parity of a window's first double (the consumer's offset into its ten doubles, see read10) is
//     (k + (P+1) + base parity of the buffer + (plane & ppar) + ALT ((P+1) + staged row)) & 1,     ALT = NZ & 1, ppar = (3 NY NZ) & 1:
// tile origins have the parity of P, so nothing in it depends on the tile, and the planes of a launch have fixed parities.  It is
// therefore QF + k + ALT (ro + dy) for the far planes and QM + k + ALT (ro + dy) for the relaxed plane with launch-uniform QF, QM,
// which are template parameters: every register index in the multiply-adds is a compile-time constant.
template <int ALT, int QF, int QM>
__global__ void __launch_bounds__(64 * (gsm::CW + 1)) k_gs_march_mf0(GsMarchArgs A) {
    using namespace gsm;
    const int P = A.forward ? 0 : 1;              // parity of the first in-plane colour: 0 forward colour order, 1 reverse
    extern __shared__ __align__(16) unsigned char smem[];
    double *sU = reinterpret_cast<double *>(smem);
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

    const long long plane = (long long) d.NY * d.NZ, elayer = (long long) d.ny * d.nz;
    const int ppar = (int) ((3 * plane) & 1);
    const int bparR = (int) ((reinterpret_cast<uintptr_t>(A.uR) >> 3) & 1), bparO = (int) ((reinterpret_cast<uintptr_t>(A.uO) >> 3) & 1);
    auto row_par = [&](int ry) {                                        // parity of the first double of staged node row ry (before base / plane)
        int jj = yl + ry; jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
        return ((jj & d.NZ) ^ k0u) & 1;                                 // (3 (jj NZ + k0u)) & 1
    };

    // =========================== DMA wave ===========================
    if (wave == CW) {
        unsigned ugo[U_INSTR];                        // (double offset of the lane's piece from the plane start) * 2 + row-start parity
        unsigned dm0[U_INSTR], dm1[U_INSTR];          // byte offset of the piece from 32 bytes in front of the plane, for an even / odd plane start
#pragma unroll
        for (int t = 0; t < U_INSTR; ++t) {
            const int Pc = 64 * t + lane;
            int r = Pc / PU, c = Pc - r * PU;
            if (r > LY - 1) { r = LY - 1; c = PU - 1; }
            int jj = yl + r; jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
            const long long rs = 3LL * ((long long) jj * d.NZ + k0u);   // >= -3
            ugo[t] = (unsigned) ((rs + 3 + 2LL * c) * 2 + (rs & 1));    // offset biased by +3 doubles so that it is never negative
            const unsigned dbl = ugo[t] >> 1, bit = ugo[t] & 1u;
            dm0[t] = 8u * dbl + 8u - 8u * bit;                          // (the piece that holds the row start is the one at or below it)
            dm1[t] = 8u * dbl + 8u * bit;
        }
        auto issueU = [&](int j) {
            const int i = plane_of(j);
            const bool rel = j & 1;
            const double *buf = rel ? A.uR : A.uO;
            const char *first = rel ? A.uR_first : A.uO_first, *last = rel ? A.uR_last : A.uO_last;
            const int par0 = ((rel ? bparR : bparO) + (i & ppar)) & 1;
            unsigned char *slot = reinterpret_cast<unsigned char *>(sU + (j % NU) * U_SLOT_D);
            if (i >= 1 && i <= d.NX - 2 && plane >= 256) {
                // An inner plane (of more than a staged row's bytes): whatever its staging touches in front of it or behind it is still the field -- no clamping, the address
                // is the plane's (uniform) base plus the lane's offset: 2 vector instructions per piece instead of ~14.  The DMA wave
                // shares its SIMD with compute wave 0, which every barrier waits for: what this wave issues, that one cannot
                // (6.15 -> 5.75 ms per sweep at 512^3, 1.04 -> 0.95 ms at 256^3; with the load's scalar-base form, no vector instruction
                // at all, it is no faster: 5.84 / 0.95).
                const char *pp = reinterpret_cast<const char *>(buf + 3LL * i * plane) - 32;
#pragma unroll
                for (int t = 0; t < U_INSTR; ++t) gsm_glds16(pp + (par0 ? dm1[t] : dm0[t]), slot + 1024 * t);
                return;
            }
            const double *pb = buf + 3LL * i * plane - 3;               // (the bias of ugo)
#pragma unroll
            for (int t = 0; t < U_INSTR; ++t) {
                const char *g = reinterpret_cast<const char *>(pb + (long long) (ugo[t] >> 1) - (long long) ((par0 + (int) (ugo[t] & 1)) & 1));
                g = g > last ? last : (g < first ? first : g);
                gsm_glds16(g, slot + 1024 * t);
            }
        };
        issueU(0); issueU(1); issueU(2);
        for (int m = 0; m < nsteps; ++m) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // B0: the planes of step m have landed; step m-1 is finished
            if (m + 1 < nsteps) { issueU(2 * m + 3); issueU(2 * m + 4); }
            __builtin_amdgcn_s_barrier();                               // B1
        }
        return;
    }

    // =========================== compute waves ===========================
    const int h = lane >> 5, cl = lane & 31;                            // row of the wave's pair, column index in the colour
    const int z0s = zb < 0 ? 0 : zb;                                    // owned columns inside the grid: [z0s, z1s]
    const int z1s = zb + 2 * C - 1 > d.NZ - 1 ? d.NZ - 1 : zb + 2 * C - 1;
    const int nd_store = 3 * (z1s - z0s + 1);
    // the eight moduli of the node this lane relaxes in colour k of plane xx: element layers xx - 1 and xx, rows y - 1, y, columns
    // z - 1, z.  Buffer loads: an element outside the grid gets an out-of-range offset, a layer outside the grid a zero-length
    // buffer, and both read 0 -- no clamping, no select, nothing staged.  Requested ONE COLOUR AHEAD of their use (mod[k & 1]).
    u2q_t mod[2][8];
    auto request_moduli = [&](auto kc, int xx) {
        constexpr int k = decltype(kc)::value, ro = k >> 1;
        constexpr int nrows = R + 1 - ro, ncols = C + 3 - k;
        const int rw = 2 * wave + h;
        const int rwe = rw < nrows ? rw : nrows - 1;
        const int y = yl + 1 + ro + 2 * rwe;
        const int ce = cl < ncols ? cl : ncols - 1;
        const int z = zl + 1 + k + 2 * ce;
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            const int ex = xx - 1 + layer;
            const bool lok = ex >= 0 && ex <= d.nx - 1;
            const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.E + (long long) (lok ? ex : 0) * elayer), 0,
                                                                                   lok ? (int) (8 * elayer) : 0, 0x00020000);
#pragma unroll
            for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) {
                    const int ey = y - 1 + dj, ez = z - 1 + dk;
                    const bool ok = ey >= 0 && ey <= d.ny - 1 && ez >= 0 && ez <= d.nz - 1;
                    const unsigned off = ok ? (unsigned) (ey * d.nz + ez) * 8u : 0x7ffffff0u;
                    mod[k & 1][4 * layer + 2 * dj + dk] = __builtin_amdgcn_raw_buffer_load_b64(re, off, 0, 0);
                }
        }
    };
    request_moduli(std::integral_constant<int, 0>{}, x0);

    for (int m = 0; m < nsteps; ++m) {
        const int x = x0 + 2 * m;
        __builtin_amdgcn_s_barrier();                                   // B0
        const int midoff = ((2 * m + 1) % NU) * U_SLOT_D;
        const int shM = bparR + (x & ppar);                             // + row parity = alignment shift of the staged rows
        const int shF = bparO + (plane_of(2 * m) & ppar);               // (planes x-1 and x+1 have the same parity)

        // ---- one colour: one node per lane, rows 2 wave + h of the colour ----
        auto phase = [&](auto kc) {
            constexpr int k = decltype(kc)::value, ro = k >> 1;
            constexpr int nrows = R + 1 - ro, ncols = C + 3 - k;
            // (the row / column arithmetic below does not depend on the step: the compiler hoists it out of the march for all four colours,
            // ~45 registers held throughout -- affordable at one wave per SIMD, and 4 % faster than redoing it per step)
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
            // out-of-range offset and move no data
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
            // the moduli of the NEXT colour (of the next step's first colour after the last one)
            if constexpr (k < 3) request_moduli(std::integral_constant<int, k + 1>{}, x);
            else if (m + 1 < nsteps) request_moduli(std::integral_constant<int, 0>{}, x + 2);
            int ni = czn - 1 - cshift;
            ni = max(ni, ni & 1);
            const int rlo = y - 1 >= 0 ? ry - 1 : ry + 1, rhi = y + 1 <= d.NY - 1 ? ry + 1 : ry - 1;
            constexpr int OFC = (QF + k + ALT * ro) & 1, OFS = (QF + k + ALT * (ro + 1)) & 1;
            constexpr int OMC = (QM + k + ALT * ro) & 1, OMS = (QM + k + ALT * (ro + 1)) & 1;
            const int lowoff = ((2 * m) % NU) * U_SLOT_D, highoff = ((2 * m + 2) % NU) * U_SLOT_D;
            const int shc = (shF + row_par(ry)) & 1, shs = (shF + row_par(rlo)) & 1;
            const int mhc = (shM + row_par(ry)) & 1, mhs = (shM + row_par(rlo)) & 1;
            // the three rows of a plane as the 3 x 3 window of l1m::side_class / mid_class
            // one row (R: 0 below, 1 the node's own, 2 above) of a plane's 3 x 3 window
            auto window_row = [&](auto rc, int slotoff, int sh_c, int sh_s, auto offc, auto offs, double (&un)[3][9]) {
                constexpr int R_ = decltype(rc)::value, oc = decltype(offc)::value, os = decltype(offs)::value, o = R_ == 1 ? oc : os;
                double w[10];
                read10(sU, slotoff + (R_ == 0 ? rlo : R_ == 1 ? ry : rhi) * ROW_D + 3 * ni + (R_ == 1 ? sh_c : sh_s) - o, w);
#pragma unroll
                for (int c = 0; c < 9; ++c) un[R_][c] = w[o + c];
            };
            // the next plane's window, a row behind each of a part's first three row waits, in the order the next part needs them (side
            // rows first): a batch of five reads lands within the ~33 multiply-adds of a row, all fifteen behind the first wait (rounds
            // 3-4) do not
            auto window_by_rows = [&](auto pc, int slotoff, int sh_c, int sh_s, auto offc, auto offs, double (&un)[3][9]) {
                constexpr int p_ = decltype(pc)::value;
                if constexpr (p_ == 0) window_row(std::integral_constant<int, 0>{}, slotoff, sh_c, sh_s, offc, offs, un);
                if constexpr (p_ == 1) window_row(std::integral_constant<int, 2>{}, slotoff, sh_c, sh_s, offc, offs, un);
                if constexpr (p_ == 2) window_row(std::integral_constant<int, 1>{}, slotoff, sh_c, sh_s, offc, offs, un);
            };
            // moduli of the eight incident elements (layer, y - 1 + dj, z - 1 + dk), requested a colour ago; zero outside the grid
            double a0[2][2], a1[2][2];
#pragma unroll
            for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) {
                    a0[dj][dk] = mkd(mod[k & 1][2 * dj + dk].x, mod[k & 1][2 * dj + dk].y);
                    a1[dj][dk] = mkd(mod[k & 1][4 + 2 * dj + dk].x, mod[k & 1][4 + 2 * dj + dk].y);
                }
            // Software pipeline over the three planes, fenced by scheduling barriers: the windows of the next plane are read while the
            // current one is multiplied, and no more (left alone the scheduler issues all nine rows' reads and all twelve coefficient
            // rows up front: 180 registers of windows, 600 spilled scalars).  Coefficient rows come through the scalar cache one row
            // ahead (coef_rows.h); as LDS reads at a wave-uniform address they were slower (a colour 2.4-3.4 us against 1.8).
            double S[3] = {0.0, 0.0, 0.0}, M6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, uself[3];
            double unA[3][9], unB[3][9];
            __builtin_amdgcn_sched_barrier(0);
            static_for<3>([&](auto rc) { window_row(rc, lowoff, shc, shs, std::integral_constant<int, OFC>{}, std::integral_constant<int, OFS>{}, unA); });
            RowPipe<12, L0NodeRows> rows{A.tab};
            asm volatile("" : "+v"(S[0]) : "v"(a0[0][0]));               // (the chain of row waits starts behind the moduli)
            rows.prime();
            __builtin_amdgcn_sched_barrier(0);
            // the next plane's windows are requested right BEHIND the wait for a part's first coefficient row: a row wait is
            // lgkmcnt(0) and would drain them (requested in front of it, every part paid the full LDS latency: waves 44 % of their
            // time in s_waitcnt, profiles/r03_gs_march_form2_pmc.json)
            {
                auto cf = l0_coef<false, 0>(rows, S[0], [&](auto pc) {
                    window_by_rows(pc, midoff, mhc, mhs, std::integral_constant<int, OMC>{}, std::integral_constant<int, OMS>{}, unB);
                });
                l1m::side_class<0, 0>(a0, unA, cf, S);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                uself[0] = unB[1][3]; uself[1] = unB[1][4]; uself[2] = unB[1][5];
                auto cf = l0_coef<true, 4>(rows, S[0], [&](auto pc) {
                    window_by_rows(pc, highoff, shc, shs, std::integral_constant<int, OFC>{}, std::integral_constant<int, OFS>{}, unA);
                });
                l1m::mid_class<0>(a0, a1, unB, cf, S, M6);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                auto cf = l0_coef<false, 8>(rows, S[0], [](auto) {});
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
        // two finished rows of a wave: all LDS reads first, then the stores (3 x 58 doubles per row: three 8-byte pieces per lane)
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
        phase(std::integral_constant<int, 0>{});
        __builtin_amdgcn_wave_barrier();
        phase(std::integral_constant<int, 1>{});
        __builtin_amdgcn_wave_barrier();
        store_rows2(1 + 4 * wave);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // B1: the rows of parity P are final
        phase(std::integral_constant<int, 2>{});
        __builtin_amdgcn_wave_barrier();
        phase(std::integral_constant<int, 3>{});
        __builtin_amdgcn_wave_barrier();
        store_rows2(2 + 4 * wave);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// Solve data of the level-0 marching sweeps, once per operator update: per node the inverse diagonal of its 3x3 diagonal block
// M = sum_e E_e K0[n-block] (MG.hh:199-220) with the Dirichlet mask folded in (0 for a fixed component, MG.hh:258-262); the rest of
// the block is formed inside the sweep
__global__ void __launch_bounds__(256) k_gs_solve_data(Dims d, const double *__restrict__ K0, const double *__restrict__ E,
                                                       const uint8_t *__restrict__ mask, double *__restrict__ sd) {
    const long long n = (long long) blockIdx.x * 256 + threadIdx.x;
    if (n >= d.nn) return;
    const int k = (int) (n % d.NZ), j = (int) ((n / d.NZ) % d.NY), i = (int) (n / ((long long) d.NZ * d.NY));
    double M[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
        const int ex = i - 1 + ((sl >> 2) & 1), ey = j - 1 + ((sl >> 1) & 1), ez = k - 1 + (sl & 1);
        if (ex < 0 || ex >= d.nx || ey < 0 || ey >= d.ny || ez < 0 || ez >= d.nz) continue;
        const double Ee = E[((long long) ex * d.ny + ey) * d.nz + ez];
        const double *blk = K0 + (3 * (7 - sl)) * 24 + 3 * (7 - sl);
        M[0] = fma(Ee, blk[0], M[0]); M[1] = fma(Ee, blk[24 + 1], M[1]); M[2] = fma(Ee, blk[48 + 2], M[2]);
    }
    const uint8_t mk = mask[n];
    sd[3 * n + 0] = (mk & 1) ? 0.0 : 1.0 / M[0];
    sd[3 * n + 1] = (mk & 2) ? 0.0 : 1.0 / M[1];
    sd[3 * n + 2] = (mk & 4) ? 0.0 : 1.0 / M[2];
}
void launch_gs_solve_data(const Dims &d, const double *K0, const double *E, const uint8_t *mask, double *sd, hipStream_t s) {
    k_gs_solve_data<<<dim3((unsigned) ((d.nn + 255) / 256)), 256, 0, s>>>(d, K0, E, mask, sd);
    VFEM_HIP(hipGetLastError());
}

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
bool launch_gs_march_mf0(const Dims &d, const double *tab, const double *E, const double *uR, const double *uO, double *dst, const double *b,
                         const double *solve_data, int cxl, int forward, int chunks, hipStream_t s, int plane_lo, int plane_hi) {
    using namespace gsm;
    if (!tab || dst == uR) return false;
    if ((reinterpret_cast<uintptr_t>(uR) & 7u) || (reinterpret_cast<uintptr_t>(uO) & 7u) || (reinterpret_cast<uintptr_t>(E) & 7u)) return false;
    if (d.NX < 2 || d.NY < 2 || d.NZ < 2) return false;
    if ((long long) d.ny * d.nz * 8 > 0x7fffffffLL || (long long) d.NY * d.NZ * 24 > 0x7fffffffLL) return false;     // (buffer-load offsets are 32-bit)
    if (plane_hi < 0 || plane_hi > d.NX - 1) plane_hi = d.NX - 1;
    if (plane_lo < 0) plane_lo = 0;
    const int first_plane = plane_lo + (((plane_lo & 1) != (cxl & 1)) ? 1 : 0);
    if (first_plane > plane_hi) return true;                          // no plane of this parity in the range
    GsMarchArgs a;
    a.d = d;
    a.tab = tab;
    a.E = E;
    auto first_piece = [](const void *p) { return reinterpret_cast<const char *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t) 15); };
    auto last_piece = [](const void *end) { return reinterpret_cast<const char *>((reinterpret_cast<uintptr_t>(end) - 1) & ~(uintptr_t) 15); };
    a.uR = uR; a.uO = uO;
    a.uR_first = first_piece(uR); a.uR_last = last_piece(uR + 3 * d.nn);
    a.uO_first = first_piece(uO); a.uO_last = last_piece(uO + 3 * d.nn);
    a.dst = dst; a.b = b; a.sd = solve_data;
    a.cxl = cxl; a.forward = forward;
    const int M = (plane_hi - first_plane) / 2 + 1;
    a.first_plane = first_plane; a.num_planes = M;
    const int P = forward ? 0 : 1;
    const int nty = (d.NY + P + 2 * R - 1) / (2 * R), ntz = (d.NZ + P + 2 * C - 1) / (2 * C);
    if (chunks <= 0) {
        // many more blocks than CUs (one block per CU is resident: short blocks even out the tail), chunks of at least 3 steps
        // (profiles/r04_gs_march_chunks.txt: 512^3, 333 tiles: 15 chunks; 160^3, 36 tiles: 26 chunks 0.26 ms against 0.37 with the
        // former floor of 8 steps and 0.44 for the row kernels)
        chunks = 1;
        while ((long long) chunks * nty * ntz < 5000 && M / (chunks + 1) >= 3) ++chunks;
    }
    if (chunks > M) chunks = M;
    a.steps_per_chunk = (M + chunks - 1) / chunks;
    const unsigned gx = (unsigned) ((M + a.steps_per_chunk - 1) / a.steps_per_chunk);
    const dim3 grd(gx, (unsigned) ntz, (unsigned) nty), blk(64, CW + 1, 1);
    // launch-uniform window parities (see the kernel's comment)
    const long long plane = (long long) d.NY * d.NZ;
    const int ppar = (int) ((3 * plane) & 1), ALT = d.NZ & 1;
    const int bparR = (int) ((reinterpret_cast<uintptr_t>(uR) >> 3) & 1), bparO = (int) ((reinterpret_cast<uintptr_t>(uO) >> 3) & 1);
    const int QF = ((P + 1) + bparO + ((cxl + 1) & ppar) + ALT * P) & 1, QM = ((P + 1) + bparR + (cxl & ppar) + ALT * P) & 1;
    static bool attr[8] = {false};
#define VFEM_GSM_LAUNCH(A_, F_, M_)                                                                                               \
    do {                                                                                                                          \
        constexpr int v_ = A_ * 4 + F_ * 2 + M_;                                                                                  \
        if (!attr[v_]) {                                                                                                          \
            VFEM_HIP(hipFuncSetAttribute((const void *) k_gs_march_mf0<A_, F_, M_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES)); \
            attr[v_] = true;                                                                                                      \
        }                                                                                                                         \
        k_gs_march_mf0<A_, F_, M_><<<grd, blk, LDS_BYTES, s>>>(a);                                                              \
    } while (0)
    switch (ALT * 4 + QF * 2 + QM) {
        case 0: VFEM_GSM_LAUNCH(0, 0, 0); break;
        case 1: VFEM_GSM_LAUNCH(0, 0, 1); break;
        case 2: VFEM_GSM_LAUNCH(0, 1, 0); break;
        case 3: VFEM_GSM_LAUNCH(0, 1, 1); break;
        case 4: VFEM_GSM_LAUNCH(1, 0, 0); break;
        case 5: VFEM_GSM_LAUNCH(1, 0, 1); break;
        case 6: VFEM_GSM_LAUNCH(1, 1, 0); break;
        default: VFEM_GSM_LAUNCH(1, 1, 1); break;
    }
#undef VFEM_GSM_LAUNCH
    VFEM_HIP(hipGetLastError());
    return true;
}

}  // namespace vfem
