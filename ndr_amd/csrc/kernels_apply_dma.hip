// k_apply_dma: the production stiffness apply with LDS-DMA staging.
//
// Same algorithm as k_apply_fast (kernels_apply.hip: mode-space element matrix, lanes = z, 12 waves of element rows; a second
// tile shape of 4 x 16 lanes per wave covers the node columns left over by the 63-wide tiles, see dma::Cfg), block marches along x, one barrier per plane), but the node planes and the element moduli are
// brought in by `global_load_lds_dwordx4` straight into a 4-deep LDS ring instead of through registers:
//   * no VGPRs and no ds_write instructions are spent on staging,
//   * three planes stay in flight behind a *counted* `s_waitcnt vmcnt(N)` + raw `s_barrier` (hipcc drains every
//     ordinary load to vmcnt(0) once per plane in the register-staged version, which is what bounded it:
//     its memory skeleton alone ran at 3 TB/s whatever the prefetch depth),
//   * the DMA is issued and retired by NW dedicated waves (ty >= TY) that do no arithmetic and no stores: `vmcnt`
//     retires in issue order, so a wave that both loads and stores waits for its own older stores whenever it waits
//     for a plane; the compute waves never wait on `vmcnt` at all,
//   * a plane is read one phase after the wait+barrier that retires it (MI355X_MICROARCH.md, two-waves item 7).
// A DMA piece is 16 bytes on the absolute 16-byte grid of memory: a tile row (65 nodes = 1560 B) starts on an 8-byte
// boundary, so its image starts at the aligned address at or 8 bytes below it and the consumer adds that one-double
// shift (a parity that depends on the row and, when a plane holds an odd number of doubles, on the plane).  Aligned
// pieces never straddle the 16-byte-aligned end of an allocation, so nothing outside the caller's buffers is read;
// a row image starts at node max(k0, 0) so that no address precedes the array.
#include "vfem_internal.h"
#include "device_utils.h"
#include <type_traits>

namespace vfem {

namespace dma {
#ifndef VFEM_DMA_TY
#define VFEM_DMA_TY 12
#endif
#ifndef VFEM_DMA_RING
#define VFEM_DMA_RING 4
#endif
#ifndef VFEM_DMA_WAVES
#define VFEM_DMA_WAVES 2
#endif
constexpr int WAVES = VFEM_DMA_TY;      // compute waves per block (8: 87.5 % of the element rows emit; 12: 91.7 %)
constexpr int RING = VFEM_DMA_RING;     // planes staged per block; plane ii + RING - 1 is requested while plane ii is consumed
constexpr int PD = RING - 1;
constexpr int NW = VFEM_DMA_WAVES;      // waves that only issue and retire the LDS-DMA
static_assert(NW == 1 || NW == 2, "one or two DMA waves");

// Tile shape.  A wave holds SUB element rows of TZ element columns (SUB * TZ = 64 lanes):
//   Main  = 1 x 64: 12 element rows x 64 columns per block, 11 x 63 complete node columns;
//   Strip = 4 x 16: 48 element rows x 16 columns, 47 x 15 node columns -- for the few node columns a row of 2^k + 1 nodes
//           leaves over after the 63-wide tiles (9 of 513, 5 of 257): with the main shape that remainder was a ninth (fifth)
//           z-tile of blocks marching every plane for 14 % (8 %) of a block's work.
template <int SUB_, int TZ_>
struct Cfg {
    static_assert(SUB_ * TZ_ == 64, "a wave is 64 lanes");
    static constexpr int SUB = SUB_, TZ = TZ_, TY = WAVES * SUB_;
    static constexpr int PU = ((TZ + 1) * 24 + 8 + 15) / 16;       // 16-byte pieces per staged node row incl. the alignment shift (98 / 26)
    static constexpr int PE = (TZ * 8 + 8 + 15) / 16;              // per staged element row (33 / 9)
    static constexpr int ROW_D = 2 * PU, EROW_D = 2 * PE;           // doubles per staged row
    static constexpr int U_INSTR = ((TY + 1) * PU + 63) / 64;       // 64 pieces per DMA instruction (20 / 20)
    static constexpr int E_INSTR = (TY * PE + 63) / 64;             // (7 / 7)
    static constexpr int NQ = U_INSTR + E_INSTR;                    // DMA instructions per plane
    static constexpr int NI = (NQ + NW - 1) / NW;                   // per DMA wave
    static_assert(NI * PD <= 63, "the planes in flight of one DMA wave must fit the 6-bit vmcnt");
    static constexpr int SLOT_BYTES = NQ * 1024;
    static constexpr int SS_DOUBLES = 3 * TY * TZ;                  // one scatter buffer (per component the sum owed to the next row in y)
    static constexpr size_t LDS_BYTES = (size_t) RING * SLOT_BYTES + 2 * SS_DOUBLES * sizeof(double);
    static_assert(LDS_BYTES <= 160 * 1024, "ring + scatter buffers must fit the 160 KB of LDS of a CU");
};
#ifndef VFEM_DMA_MAIN_SUB
#define VFEM_DMA_MAIN_SUB 1
#endif
using Main = Cfg<VFEM_DMA_MAIN_SUB, 64 / VFEM_DMA_MAIN_SUB>;
using Strip = Cfg<4, 16>;
}  // namespace dma

struct DmArgs2 { double v[36]; };

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                     (__attribute__((address_space(3))) void *) l, 16, 0, 0);
}

// RES: out = b - K u with 0 at the fixed components (the residual of the V-cycle) instead of K u: right-hand side and mask of the
// node a thread emits are requested one phase ahead, before the arithmetic of the plane in between
template <int EXP, class C, bool RES = false, bool LX = false>
__device__ __forceinline__ void apply_tile(const Dims &d, const DmArgs2 &dm, const double *__restrict__ E,
                                           const double *__restrict__ u, double *__restrict__ out,
                                           int planes_per_chunk, const char *u_last, const char *e_last,
                                           int plane_lo, int plane_hi, int kz_origin, int ztile, int ytile, int adv, int hi_node,
                                           const double *__restrict__ rhs = nullptr, const uint8_t *__restrict__ fixed = nullptr) {
    using namespace dma;
    constexpr int TY = C::TY, TZ = C::TZ, PU = C::PU, PE = C::PE, ROW_D = C::ROW_D, EROW_D = C::EROW_D;
    constexpr int U_INSTR = C::U_INSTR, NQ = C::NQ, NI = C::NI, SLOT_BYTES = C::SLOT_BYTES, SS_DOUBLES = C::SS_DOUBLES;
    // ablations (-DVFEM_ABLATION builds only, vfem_debug_set(1, n), wrong results): 1 no LDS scatter, 2 no mode-space arithmetic, 3 neither, 4 memory skeleton,
    // 5 DMA + barriers only, 6 stores + barriers only, 7 / 8: variants 6 / 0 with non-temporal stores, 9: variant 4 storing to two
    // planes only, 10: variant 4 loading four planes only, 11: variant 6 with row-contiguous 16-byte stores
    constexpr int X = (EXP == 8) ? 0 : ((EXP == 7 || EXP == 11) ? 6 : (EXP >= 9 ? 4 : EXP));
    constexpr bool NT = EXP == 7 || EXP == 8;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ring = smem;
    double *sS = reinterpret_cast<double *>(smem + (size_t) RING * SLOT_BYTES);

    const int lane = threadIdx.x, wave = threadIdx.y;
    const int tz = lane % TZ, ty = wave * C::SUB + lane / TZ;     // element column / row inside the tile (compute waves)
    // The tile computes complete sums for the node columns k0 + 1 .. k0 + TZ - 1 and EMITS those in [k0 + 1, hi_node) -- the next tile's
    // first column, `adv` columns on (TZ - 1 = 63 in the plain tiling).  Line-exclusive tiling (lx, round 4; VERDICT r03 item 5,
    // profiles/r04_apply_storeprobe.txt): tiles advance by 57 columns, so neighbouring tiles share six complete columns, and the
    // boundary between two tiles of a row is moved up to the next 128-byte boundary of the ADDRESS SPACE (16 doubles): every line of the
    // result is then written whole by one wave -- inside the reference's layout, whose 12 312-byte row pitch shifts the line grid by
    // one node per row and per plane.  (A 64-lane wave has 63 complete columns: 57 + the 6 a line can reach back over.)
    const int k0 = kz_origin + ztile * adv - 1;
    const int j0 = ytile * (TY - 1) - 1;
    const int p0 = plane_lo + blockIdx.x * planes_per_chunk;      // output planes [p0, p1] of [plane_lo, plane_hi]
    int p1 = p0 + planes_per_chunk - 1;
    if (p1 > plane_hi) p1 = plane_hi;
    if (p0 > plane_hi) return;
    const int k0u = k0 < 0 ? 0 : k0;                   // first node column held by a staged row image
    const int cshift = k0u - k0;                       // 0, or 1 in the first z tile

    const long long plane = (long long) d.NY * d.NZ;
    const long long elayer = (long long) d.ny * d.nz;
    // addresses in units of doubles from address 0, so that parities are those of the absolute 16-byte grid
    const long long ubase8 = (long long) (reinterpret_cast<uintptr_t>(u) >> 3);
    const long long ebase8 = (long long) (reinterpret_cast<uintptr_t>(E) >> 3);
    const int ppar = (int) ((3 * plane) & 1), epar = (int) (elayer & 1);       // parity added per plane / element layer
    const int i_start = p0 > 0 ? p0 - 1 : 0;
    const int i_end = p1 + 1 < d.NX - 1 ? p1 + 1 : d.NX - 1;

    // ---- DMA waves (wave >= WAVES): request the planes, retire them in order, keep in step with the barriers of the compute waves ----
    // Stores and loads of one wave retire through the same in-order vmcnt, so a wave that does both waits for the (slow)
    // completion of its older stores whenever it waits for a plane; waves that only load do not.
    if (wave >= WAVES) {
        auto dma_loop = [&](auto Wc) {
            constexpr int W = decltype(Wc)::value;
            constexpr int Q0 = W * NI;
            constexpr int N = (NQ - Q0) < NI ? (NQ - Q0) : NI;     // instructions of this wave per plane
            long long go[N > 0 ? N : 1];        // double offset of this lane's piece from the plane / layer start
            int par[N > 0 ? N : 1];             // parity of the row start at plane 0
#pragma unroll
            for (int t = 0; t < N; ++t) {
                const int q = Q0 + t;
                if (q < U_INSTR) {              // pieces [64 q, 64 q + 64) of the (TY+1) x PU piece image of the node plane
                    const int P = 64 * q + lane;
                    int r = P / PU, c = P - r * PU;
                    if (r > TY) { r = TY; c = PU - 1; }
                    int jj = j0 + r;
                    jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
                    const long long rs = 3LL * ((long long) jj * d.NZ + k0u);
                    go[t] = rs + 2LL * c;
                    par[t] = (int) ((ubase8 + rs) & 1);
                } else {                        // pieces of the TY x PE piece image of the element layer
                    const int P = 64 * (q - U_INSTR) + lane;
                    int r = P / PE, c = P - r * PE;
                    if (r > TY - 1) { r = TY - 1; c = PE - 1; }
                    int jj = j0 + r;
                    jj = jj < 0 ? 0 : (jj > d.ny - 1 ? d.ny - 1 : jj);
                    const long long rs = (long long) jj * d.nz + k0u;
                    go[t] = rs + 2LL * c;
                    par[t] = (int) ((ebase8 + rs) & 1);
                }
            }
            auto issue = [&](int i, int sl) {   // node plane i + element layer min(i, nx-1) -> ring slot sl
                if (X == 6) return;
                unsigned char *slot = ring + (size_t) sl * SLOT_BYTES;
                const int ia = EXP == 10 ? (i & 3) : i;
                const int il = ia < d.nx ? ia : d.nx - 1;
                const int ip = ia & ppar, ie = il & epar;
                const long long ub = 3LL * ia * plane, eb = (long long) il * elayer;
#pragma unroll
                for (int t = 0; t < N; ++t) {
                    const int q = Q0 + t;
                    const char *g;
                    if (q < U_INSTR) {          // aligned piece: one double below when (row start + plane offset) is odd
                        g = reinterpret_cast<const char *>(u + (ub + go[t] - ((par[t] + ip) & 1)));
                        g = g > u_last ? u_last : g;
                    } else {
                        g = reinterpret_cast<const char *>(E + (eb + go[t] - ((par[t] + ie) & 1)));
                        g = g > e_last ? e_last : g;
                    }
                    glds16(g, slot + 1024 * q);
                }
            };
#pragma unroll
            for (int r = 0; r < RING; ++r)
                if (i_start + r <= i_end) issue(i_start + r, r);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
            int cur = 1 % RING;
            for (int ii = i_start + 1; ii <= i_end; ++ii) {
                const bool more = ii + PD <= i_end;
                if (more) issue(ii + PD, cur == 0 ? RING - 1 : cur - 1);
                cur = cur + 1 == RING ? 0 : cur + 1;
                // plane ii + 1 must have landed before the barrier; planes ii+2 .. ii+PD were requested after it
                if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * N) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (p1 == d.NX - 1) __builtin_amdgcn_s_barrier();
        };
        if (wave == WAVES) dma_loop(std::integral_constant<int, 0>{});
        else dma_loop(std::integral_constant<int, (NW > 1 ? 1 : 0)>{});
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------
    const int ej = j0 + ty, ek = k0 + tz;
    const bool elem_ok = ej >= 0 && ej < d.ny && ek >= 0 && ek < d.nz;
    const int hi_n = hi_node < d.NZ ? hi_node : d.NZ;
    const bool out_ok = ty >= 1 && tz >= 1 && ej < d.NY && ek < d.NZ && (LX ? ek < hi_n + 6 : ek < hi_n);
    // line-exclusive emission, in doubles relative to the first emitted column: this thread's first component sits at e_rel, the tile's
    // range is [t_lo, 3 (hi_n - k0 - 1) + t_hi) with t_lo / t_hi = the distance from the tile's nominal boundaries up to the next
    // multiple of 16 doubles of the address space (0 at the ends of a row), which depends on the row and the plane through s
    const int span = 3 * (hi_n - (k0 + 1));
    const int srow = LX ? (int) (((long long) (reinterpret_cast<uintptr_t>(out) >> 3) + 3LL * ((long long) ej * d.NZ + (k0 + 1))) & 15) : 0;
    const int sp15 = (int) ((3 * ((long long) d.NY * d.NZ)) & 15);
    const bool round_lo = LX && k0 + 1 > 0, round_hi = LX && hi_n < d.NZ;
    // parities of the rows this thread consumes (node rows ty, ty+1; element row ty)
    int rpar[2], erpar;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int jj = j0 + ty + s;
        jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
        rpar[s] = (int) ((ubase8 + 3LL * ((long long) jj * d.NZ + k0u)) & 1);
    }
    {
        int jj = ej < 0 ? 0 : (ej > d.ny - 1 ? d.ny - 1 : ej);
        erpar = (int) ((ebase8 + (long long) jj * d.nz + k0u) & 1);
    }

    // ---- per-thread read offsets into a slot (doubles) ----------------------------------------
    int c0 = tz - cshift, c1 = tz + 1 - cshift;
    c0 = c0 < 0 ? 0 : c0; c1 = c1 < 0 ? 0 : c1;
    const int o00 = ty * ROW_D + 3 * c0, o01 = ty * ROW_D + 3 * c1;
    const int o10 = (ty + 1) * ROW_D + 3 * c0, o11 = (ty + 1) * ROW_D + 3 * c1;
    int ce = tz - cshift; ce = ce < 0 ? 0 : ce;
    const int oE0 = (U_INSTR * 1024) / 8 + ty * EROW_D + ce;

    auto face_modes_at = [&](double f[4][3], const double *su, int q00, int q01, int q10, int q11) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double a = su[q00 + c], bb = su[q01 + c], cc = su[q10 + c], dd = su[q11 + c];
            const double s0 = a + bb, d0 = bb - a, s1 = cc + dd, d1 = dd - cc;
            f[0][c] = s0 + s1; f[1][c] = d0 + d1; f[2][c] = s1 - s0; f[3][c] = d1 - d0;
        }
    };
    auto scatter_face = [&](const double acc[4][3], double wa[3], int buf) {
        double *sX = sS + buf * SS_DOUBLES;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double p = acc[0][c] - acc[2][c], q = acc[1][c] - acc[3][c];
            const double r = acc[0][c] + acc[2][c], t = acc[1][c] + acc[3][c];
            wa[c] = p - q;
            if (X == 1 || X >= 3) { wa[c] += (p + q) + (r - t) + (r + t); continue; }
            // the two terms owed to the z-neighbour move one lane up (DPP), only the sum owed to the y-neighbour (next row)
            // goes through LDS: 3 instead of 9 doubles written and read per thread and plane.  In the strip shape lane tz = 0
            // of a sub-row receives the last lane of the sub-row below: harmless, its column is never emitted.
            wa[c] += lane_below(p + q);
            sX[(c * TY + ty) * TZ + tz] = (r - t) + lane_below(r + t);
        }
    };
    auto request_rhs = [&](int i, double rhs_n[3], unsigned &fixed_n) {      // for the node emit_plane(i, ...) will write
        rhs_n[0] = rhs_n[1] = rhs_n[2] = 0.0;
        fixed_n = 0;
        if (!RES || !out_ok) return;
        const long long n = (long long) i * plane + (long long) ej * d.NZ + ek;
#pragma unroll
        for (int c = 0; c < 3; ++c) rhs_n[c] = rhs[3 * n + c];
        if (fixed != nullptr) fixed_n = fixed[n];
    };
    auto emit_plane = [&](int i, const double wa[3], int buf, const double rhs_n[3], unsigned fixed_n) {
        if (X == 5) return;
        if (EXP == 11 ? !(ty >= 1 && ej < d.NY) : !out_ok) return;
        const double *sX = sS + buf * SS_DOUBLES;
        const long long n = (long long) (EXP == 9 ? (i & 1) : i) * plane + (long long) ej * d.NZ + ek;
        double w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            w[c] = (X == 1 || X >= 3) ? wa[c] : wa[c] + sX[(c * TY + ty - 1) * TZ + tz];
        if (RES) {
#pragma unroll
            for (int c = 0; c < 3; ++c) w[c] = ((fixed_n >> c) & 1) ? 0.0 : rhs_n[c] - w[c];
        }
        if (EXP == 11) {      // timing only: the row written as contiguous 16-byte pieces (piece = lane, then 64 + lane), values meaningless
            const long long n0 = (long long) i * plane + (long long) ej * d.NZ + (k0 + 1);     // first output node of the row
            double *row = out + 3 * n0;
            int nd = 3 * (d.NZ - (k0 + 1)); nd = nd > 189 ? 189 : nd;                          // doubles in the row
            const int g0 = 2 * tz, g1 = 128 + 2 * tz;
            if (g0 + 1 < nd) { row[g0] = w[0]; row[g0 + 1] = w[1]; }
            else if (g0 < nd) row[g0] = w[0];
            if (g1 + 1 < nd) { row[g1] = w[2]; row[g1 + 1] = w[0]; }
            else if (g1 < nd) row[g1] = w[2];
            return;
        }
        if (NT) { __builtin_nontemporal_store(w[0], &out[3 * n]); __builtin_nontemporal_store(w[1], &out[3 * n + 1]); __builtin_nontemporal_store(w[2], &out[3 * n + 2]); }
        else if (LX) {
            const int sl = (srow + i * sp15) & 15;                              // first emitted column of this row and plane, in doubles mod 16
            const int t_lo = round_lo ? ((-sl) & 15) : 0, t_hi = span + (round_hi ? ((-(sl + span)) & 15) : 0);
            const int e_rel = 3 * (tz - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (e_rel + c >= t_lo && e_rel + c < t_hi) out[3 * n + c] = w[c];
        }
        else { out[3 * n] = w[0]; out[3 * n + 1] = w[1]; out[3 * n + 2] = w[2]; }
    };

    double fold[4][3], carry[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) carry[q][c] = 0.0;

    auto face_modes = [&](double f[4][3], const double *su, int i) {
        const int ip = i & ppar;
        const int s0r = (rpar[0] + ip) & 1, s1r = (rpar[1] + ip) & 1;   // one-double shift of the aligned row images
        face_modes_at(f, su, o00 + s0r, o01 + s0r, o10 + s1r, o11 + s1r);
    };
    auto process = [&](double Ee, const double *su, const int q[4], int buf, double wa[3]) {
        if (!elem_ok) Ee = 0.0;
        double fnew[4][3];
        if (X >= 4) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int c = 0; c < 3; ++c) fnew[qq][c] = Ee + qq + c;
        } else face_modes_at(fnew, su, q[0], q[1], q[2], q[3]);
        if (X >= 2) {
            double acc2[4][3];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int c = 0; c < 3; ++c) { acc2[qq][c] = Ee * (fnew[qq][c] + fold[qq][c]) + carry[qq][c]; carry[qq][c] = fnew[qq][c]; fold[qq][c] = fnew[qq][c]; }
            scatter_face(acc2, wa, buf);
            return;
        }
        double m[8][3];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                m[q][c] = fold[q][c] + fnew[q][c];
                m[4 + q][c] = fnew[q][c] - fold[q][c];
                fold[q][c] = fnew[q][c];
            }
        double qv[8][3];
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int c = 0; c < 3; ++c) qv[p][c] = (p == 0) ? 0.0 : dm.v[3 * p + c] * m[p][c];
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

    // ---- prologue: planes i_start .. i_start+RING-1 were requested by the DMA waves and are retired once ------------------
    // ring slot of plane i is (i - i_start) mod RING
    __builtin_amdgcn_s_barrier();
    {
        const double *su = reinterpret_cast<const double *>(ring);
        face_modes(fold, su, i_start);
    }
    auto e_offset = [&](int i) { return oE0 + ((erpar + (i & epar)) & 1); };
    double Eprev = reinterpret_cast<const double *>(ring)[e_offset(i_start)];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // slot 0 may now be overwritten (plane i_start+RING)

    // read offsets for the two step parities (the alignment shift alternates with the plane when 3*plane is odd)
    int qoff[2][4], eoff2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = i_start + 1 + k, ip = i & ppar;
        const int s0r = (rpar[0] + ip) & 1, s1r = (rpar[1] + ip) & 1;
        qoff[k][0] = o00 + s0r; qoff[k][1] = o01 + s0r; qoff[k][2] = o10 + s1r; qoff[k][3] = o11 + s1r;
        eoff2[k] = oE0 + ((erpar + (i & epar)) & 1);
    }

    int buf = 0, cur = 1 % RING;          // cur = ring slot of the plane consumed by the next phase
    auto run_phase = [&](int ii, int k) {
        const double *su = reinterpret_cast<const double *>(ring + (size_t) cur * SLOT_BYTES);
        cur = cur + 1 == RING ? 0 : cur + 1;
        double wa[3];
        const double Enext = su[eoff2[k]];             // modulus of layer ii, used in the next phase
        double rhs_n[3];
        unsigned fixed_n;
        request_rhs(ii - 1 >= p0 ? ii - 1 : p0, rhs_n, fixed_n);
        process(Eprev, su, qoff[k], buf, wa);
        Eprev = Enext;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // the DMA waves arrive here once plane ii + 1 has landed
        if (ii - 1 >= p0) emit_plane(ii - 1, wa, buf, rhs_n, fixed_n);
        buf ^= 1;
    };
    for (int ii = i_start + 1; ii <= i_end; ii += 2) {
        run_phase(ii, 0);
        if (ii + 1 <= i_end) run_phase(ii + 1, 1);
    }
    if (p1 == d.NX - 1) {
        double wa[3], rhs_n[3];
        unsigned fixed_n;
        request_rhs(d.NX - 1, rhs_n, fixed_n);
        scatter_face(carry, wa, buf);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        emit_plane(d.NX - 1, wa, buf, rhs_n, fixed_n);
    }
}

// One launch for both tile shapes: blocks with blockIdx.y < n_main are main tiles; blockIdx.y == n_main is the strip over the
// last node columns (origin `strip_origin`), which needs fewer y tiles -- the surplus blocks leave at once.  (As a launch of
// its own the strip's few blocks marched alone at the end and gave back half of what they save.)
template <int EXP, bool LX>
__global__ void __launch_bounds__(64 * (dma::WAVES + dma::NW)) k_apply_dma(Dims d, DmArgs2 dm, const double *__restrict__ E,
                                                      const double *__restrict__ u, double *__restrict__ out,
                                                      int planes_per_chunk, const char *u_last, const char *e_last,
                                                      int plane_lo, int plane_hi, int n_main, int strip_origin, int strip_ytiles, int strip_ppc, int adv) {
    if ((int) blockIdx.y < n_main) {
        if ((int) blockIdx.z * (dma::Main::TY - 1) > d.NY - 1) return;      // the grid's z extent covers the main shape's y tiles
        const int hi = (int) blockIdx.y + 1 < n_main ? ((int) blockIdx.y + 1) * adv : strip_origin;
        apply_tile<EXP, dma::Main, false, LX>(d, dm, E, u, out, planes_per_chunk, u_last, e_last, plane_lo, plane_hi, 0, (int) blockIdx.y, (int) blockIdx.z, adv, hi);
    } else {
        if (EXP != 0 || (int) blockIdx.z >= strip_ytiles) return;
        // the strip may use longer x-chunks than the main tiles (fewer, longer blocks); chunks beyond its last plane leave inside
        if (EXP == 0) apply_tile<0, dma::Strip, false, LX>(d, dm, E, u, out, strip_ppc, u_last, e_last, plane_lo, plane_hi, strip_origin, 0, (int) blockIdx.z, dma::Strip::TZ - 1, d.NZ);
    }
}

template <bool LX>
__global__ void __launch_bounds__(64 * (dma::WAVES + dma::NW)) k_residual_dma(Dims d, DmArgs2 dm, const double *__restrict__ E,
                                                      const double *__restrict__ u, double *__restrict__ out,
                                                      int planes_per_chunk, const char *u_last, const char *e_last,
                                                      int plane_lo, int plane_hi, int n_main, int strip_origin, int strip_ytiles, int strip_ppc, int adv,
                                                      const double *__restrict__ rhs, const uint8_t *__restrict__ fixed) {
    if ((int) blockIdx.y < n_main) {
        if ((int) blockIdx.z * (dma::Main::TY - 1) > d.NY - 1) return;
        const int hi = (int) blockIdx.y + 1 < n_main ? ((int) blockIdx.y + 1) * adv : strip_origin;
        apply_tile<0, dma::Main, true, LX>(d, dm, E, u, out, planes_per_chunk, u_last, e_last, plane_lo, plane_hi, 0, (int) blockIdx.y, (int) blockIdx.z, adv, hi, rhs, fixed);
    } else {
        if ((int) blockIdx.z >= strip_ytiles) return;
        apply_tile<0, dma::Strip, true, LX>(d, dm, E, u, out, strip_ppc, u_last, e_last, plane_lo, plane_hi, strip_origin, 0, (int) blockIdx.z, dma::Strip::TZ - 1, d.NZ, rhs, fixed);
    }
}

// chunks: number of x-chunks of the marching blocks (0 = default); strip: 0 tiles the whole row with the main shape;
// rhs != null: out = rhs - K u, 0 where `fixed` (may be null) has the component's bit set
bool launch_apply_dma(const Dims &d, const double *Dm_host, const double *E, const double *E_alloc_end, const double *u,
                      double *out, hipStream_t s, int plane_lo, int plane_hi, int g_dma_chunks, int g_dma_strip,
                      const double *rhs, const uint8_t *fixed, int g_dma_lx) {
    using namespace dma;
    if (plane_hi < 0 || plane_hi > d.NX - 1) plane_hi = d.NX - 1;
    if (plane_lo < 0) plane_lo = 0;
    if (plane_lo > plane_hi) return true;
    const int np = plane_hi - plane_lo + 1;
    const char *u_end = reinterpret_cast<const char *>(u + 3 * d.nn);
    const char *e_end = reinterpret_cast<const char *>(E_alloc_end);
    if ((reinterpret_cast<uintptr_t>(u) & 7u) || (reinterpret_cast<uintptr_t>(E) & 7u)) return false;
    DmArgs2 dm;
    for (int q = 0; q < 36; ++q) dm.v[q] = Dm_host[q];
    int nchunks = np >= 64 ? 8 : (np >= 16 ? 4 : 1);
    if (np >= 1024) nchunks = 16;
    if (g_dma_chunks > 0 && np >= 4 * g_dma_chunks) nchunks = g_dma_chunks;
    const int ppc = (np + nchunks - 1) / nchunks;
    const unsigned gx = (unsigned) ((np + ppc - 1) / ppc);
    // last admissible (aligned) piece: the one holding the last byte of each array
    auto last_piece = [](const char *end) { return reinterpret_cast<const char *>((reinterpret_cast<uintptr_t>(end) - 1) & ~(uintptr_t) 15); };
    const int g_apply_skeleton = ablate_apply();
    // z tiling: 63 node columns per main tile; a remainder of at most 15 columns goes to the strip shape (47 x 15 per block)
    int wz = Main::TZ - 1;
    int n_main = (d.NZ + wz - 1) / wz, rem = 0;
    if (g_dma_strip && g_apply_skeleton == 0 && d.NZ > wz) {
        const int r = d.NZ - wz * (d.NZ / wz);
        if (r > 0 && r <= Strip::TZ - 1) { n_main = d.NZ / wz; rem = r; }
    }
    // line-exclusive tiling (see apply_tile): main tiles advance by TZ - 7 = 57 columns and split rows on 128-byte boundaries.  g_dma_lx:
    // 0 off, 2 on, 1 (default) on when it needs no more z tiles than the plain tiling has blocks in z (513 = 9 x 57: as many as 8 + strip)
    int lx = 0;
    if (VFEM_DMA_MAIN_SUB == 1 && g_apply_skeleton == 0 && d.NZ > Main::TZ - 1 && g_dma_lx != 0) {
        const int alx = Main::TZ - 7, nlx = (d.NZ + alx - 1) / alx;
        if (g_dma_lx == 2 || nlx <= n_main + (rem > 0 ? 1 : 0)) { lx = 1; wz = alx; n_main = nlx; rem = 0; }
    }
    const int ytiles = (d.NY + Main::TY - 2) / (Main::TY - 1), strip_ytiles = rem > 0 ? (d.NY + Strip::TY - 2) / (Strip::TY - 1) : 0;
    const dim3 grd(gx, (unsigned) (n_main + (rem > 0 ? 1 : 0)), (unsigned) ytiles), blk(64, WAVES + NW, 1);
    // the strip's blocks are few: with x-chunks twice as long the launch at 512^3 fits 12 rounds of 256 blocks instead of 12.1
    const int strip_ppc = (g_dma_strip == 2 || gx < 2) ? ppc : 2 * ppc;
    static_assert(Main::LDS_BYTES == Strip::LDS_BYTES, "both tile shapes use the same dynamic LDS size");
    const char *ul = last_piece(u_end), *el = last_piece(e_end);
#define VFEM_DMA_LAUNCH2(X, L)                                                                                               \
    do {                                                                                                                     \
        static bool attr = false;                                                                                            \
        if (!attr) {                                                                                                         \
            VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<X, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) Main::LDS_BYTES)); \
            attr = true;                                                                                                     \
        }                                                                                                                    \
        k_apply_dma<X, L><<<grd, blk, Main::LDS_BYTES, s>>>(d, dm, E, u, out, ppc, ul, el, plane_lo, plane_hi, n_main, rem > 0 ? wz * n_main : d.NZ, strip_ytiles, strip_ppc, wz); \
    } while (0)
#define VFEM_DMA_LAUNCH(X) VFEM_DMA_LAUNCH2(X, false)
    if (rhs) {
        static bool attr = false;
        if (!attr) {
            VFEM_HIP(hipFuncSetAttribute((const void *) k_residual_dma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) Main::LDS_BYTES));
            VFEM_HIP(hipFuncSetAttribute((const void *) k_residual_dma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) Main::LDS_BYTES));
            attr = true;
        }
        if (lx) k_residual_dma<true><<<grd, blk, Main::LDS_BYTES, s>>>(d, dm, E, u, out, ppc, ul, el, plane_lo, plane_hi, n_main, rem > 0 ? wz * n_main : d.NZ, strip_ytiles, strip_ppc, wz, rhs, fixed);
        else k_residual_dma<false><<<grd, blk, Main::LDS_BYTES, s>>>(d, dm, E, u, out, ppc, ul, el, plane_lo, plane_hi, n_main, rem > 0 ? wz * n_main : d.NZ, strip_ytiles, strip_ppc, wz, rhs, fixed);
        VFEM_HIP(hipGetLastError());
        return true;
    }
    if (lx) { VFEM_DMA_LAUNCH2(0, true); VFEM_HIP(hipGetLastError()); return true; }
#ifdef VFEM_ABLATION
    switch (g_apply_skeleton) {
        case 1: VFEM_DMA_LAUNCH(1); break;
        case 2: VFEM_DMA_LAUNCH(2); break;
        case 3: VFEM_DMA_LAUNCH(3); break;
        case 4: VFEM_DMA_LAUNCH(4); break;
        case 5: VFEM_DMA_LAUNCH(5); break;
        case 6: VFEM_DMA_LAUNCH(6); break;
        case 7: VFEM_DMA_LAUNCH(7); break;
        case 8: VFEM_DMA_LAUNCH(8); break;
        case 9: VFEM_DMA_LAUNCH(9); break;
        case 10: VFEM_DMA_LAUNCH(10); break;
        case 11: VFEM_DMA_LAUNCH(11); break;
        default: VFEM_DMA_LAUNCH(0);
    }
#else
    VFEM_DMA_LAUNCH(0);
#endif
#undef VFEM_DMA_LAUNCH
#undef VFEM_DMA_LAUNCH2
    VFEM_HIP(hipGetLastError());
    return true;
}

}  // namespace vfem
