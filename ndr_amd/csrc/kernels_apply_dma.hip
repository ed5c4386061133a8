// k_apply_dma: the production stiffness apply with LDS-DMA staging.
//
// Same algorithm and tiling as k_apply_fast (kernels_apply.hip: mode-space element matrix, lanes = z, TY waves = TY
// element rows, block marches along x, one barrier per plane), but the node planes and the element moduli are
// brought in by `global_load_lds_dwordx4` straight into a 4-deep LDS ring instead of through registers:
//   * no VGPRs and no ds_write instructions are spent on staging,
//   * three planes stay in flight behind a *counted* `s_waitcnt vmcnt(N)` + raw `s_barrier` (hipcc drains every
//     ordinary load to vmcnt(0) once per plane in the register-staged version, which is what bounded it:
//     its memory skeleton alone ran at 3 TB/s whatever the prefetch depth),
//   * the DMA is issued and retired by NW dedicated waves (ty >= TY) that do no arithmetic and no stores: `vmcnt`
//     retires in issue order, so a wave that both loads and stores waits for its own older stores whenever it waits
//     for a plane; the compute waves never wait on `vmcnt` at all (NW = 0 builds the earlier shared form),
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
constexpr int TY = VFEM_DMA_TY, TZ = 64;    // TY waves = TY element rows per block (8: 87.5 % of the lanes emit; 12: 91.7 %)
constexpr int ROW_D = 196;                  // doubles per staged node row (98 pieces)
constexpr int U_PIECES = (TY + 1) * 98;
constexpr int U_INSTR = (U_PIECES + 63) / 64;          // 64 pieces of 16 B per instruction (TY = 8: 14, TY = 12: 20)
constexpr int E_INSTR = (TY * 33 + 63) / 64;           // TY rows x 33 pieces (32 + 1 for the alignment shift) (5 / 7)
static_assert(U_INSTR <= 2 * TY && U_INSTR + E_INSTR <= 3 * TY && E_INSTR <= TY, "every wave issues 2 or 3 DMA instructions per plane");
static_assert(U_INSTR + E_INSTR >= 2 * TY, "every wave issues at least 2 DMA instructions per plane (wait_plane counts on it)");
constexpr int SLOT_BYTES = U_INSTR * 1024 + E_INSTR * 1024;   // 18432
#ifndef VFEM_DMA_RING
#define VFEM_DMA_RING 4
#endif
constexpr int RING = VFEM_DMA_RING;     // planes staged per block; plane ii + RING - 1 is requested while plane ii is consumed
constexpr int PD = RING - 1;
#ifndef VFEM_DMA_WAVES
#define VFEM_DMA_WAVES 2
#endif
constexpr int NW = VFEM_DMA_WAVES;      // waves that only issue the LDS-DMA (0: every compute wave issues its share and waits on it)
constexpr int NQ = U_INSTR + E_INSTR;   // DMA instructions per plane
constexpr int NI = NW > 0 ? (NQ + NW - 1) / NW : 0;      // per DMA wave
static_assert(NW == 0 || NI * PD <= 63, "the planes in flight of one DMA wave must fit the 6-bit vmcnt");
constexpr int SS_DOUBLES = 3 * TY * TZ;     // one scatter buffer (per component the sum owed to the next row in y)
constexpr size_t LDS_BYTES = (size_t) RING * SLOT_BYTES + 2 * SS_DOUBLES * sizeof(double);
static_assert(LDS_BYTES <= 160 * 1024, "ring + scatter buffers must fit the 160 KB of LDS of a CU");
}  // namespace dma

struct DmArgs2 { double v[36]; };

int g_dma_chunks = 0;        // vfem_debug_set(7, n): number of x-chunks of the marching blocks (0 = default)

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                     (__attribute__((address_space(3))) void *) l, 16, 0, 0);
}

template <int CNT>   // CNT = LDS-DMA instructions this wave issues per plane (2 or 3)
__device__ __forceinline__ void wait_plane(bool has_stores, bool steady) {
    // retire the DMA of the oldest plane in flight (vmcnt retires in issue order); in the steady state PD - 1 newer planes
    // (CNT instructions each) and the two output stores (dwordx4 + dwordx2) of each of the last PD - 1 phases were issued after it
    using namespace dma;
    if (!steady) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    if (has_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * (CNT + 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * CNT) : "memory");
}

template <int EXP>
__global__ void __launch_bounds__(64 * (dma::TY + dma::NW)) k_apply_dma(Dims d, DmArgs2 dm, const double *__restrict__ E,
                                                      const double *__restrict__ u, double *__restrict__ out,
                                                      int planes_per_chunk, const char *u_last, const char *e_last,
                                                      int plane_lo, int plane_hi) {
    using namespace dma;
    // 7 / 8: variants 6 / 0 with non-temporal stores; 9: variant 4 storing to two planes only; 10: variant 4 loading four planes only
    // 11: variant 6 (stores only) with row-contiguous 16-byte stores of meaningless values (timing of the store pattern)
    constexpr int X = (EXP == 8) ? 0 : ((EXP == 7 || EXP == 11) ? 6 : (EXP >= 9 ? 4 : EXP));
    constexpr bool NT = EXP == 7 || EXP == 8;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ring = smem;
    double *sS = reinterpret_cast<double *>(smem + (size_t) RING * SLOT_BYTES);

    const int tz = threadIdx.x, ty = threadIdx.y;      // ty is wave-uniform (64 x 8 block)
    const int wave = ty;
    const int k0 = blockIdx.y * (TZ - 1) - 1;
    const int j0 = blockIdx.z * (TY - 1) - 1;
    const int p0 = plane_lo + blockIdx.x * planes_per_chunk;      // output planes [p0, p1] of [plane_lo, plane_hi]
    int p1 = p0 + planes_per_chunk - 1;
    if (p1 > plane_hi) p1 = plane_hi;
    if (p0 > plane_hi) return;
    const int k0u = k0 < 0 ? 0 : k0;                   // first node column held by a staged row image
    const int cshift = k0u - k0;                       // 0, or 1 in the first z tile

    const int ej = j0 + ty, ek = k0 + tz;
    const bool elem_ok = ej >= 0 && ej < d.ny && ek >= 0 && ek < d.nz;
    const bool out_ok = ty >= 1 && tz >= 1 && ej < d.NY && ek < d.NZ;
    const long long plane = (long long) d.NY * d.NZ;
    const long long elayer = (long long) d.ny * d.nz;

    // ---- DMA descriptors of this wave -----------------------------------------------------------
    // addresses in units of doubles from address 0, so that parities are those of the absolute 16-byte grid
    const long long ubase8 = (long long) (reinterpret_cast<uintptr_t>(u) >> 3);
    const long long ebase8 = (long long) (reinterpret_cast<uintptr_t>(E) >> 3);
    const int ppar = (int) ((3 * plane) & 1), epar = (int) (elayer & 1);       // parity added per plane / element layer
    const int i_start = p0 > 0 ? p0 - 1 : 0;
    const int i_end = p1 + 1 < d.NX - 1 ? p1 + 1 : d.NX - 1;

    // ---- DMA waves (ty >= TY): request the planes, retire them in order, keep in step with the barriers of the compute waves ----
    // Stores and loads of one wave retire through the same in-order vmcnt, so a wave that does both waits for the (slow)
    // completion of its older stores whenever it waits for a plane; waves that only load do not.
    if (NW > 0 && ty >= TY) {
        auto dma_loop = [&](auto Wc) {
            constexpr int W = decltype(Wc)::value;
            constexpr int Q0 = W * NI;
            constexpr int N = (NQ - Q0) < NI ? (NQ - Q0) : NI;     // instructions of this wave per plane
            long long go[N > 0 ? N : 1];        // double offset of this lane's piece from the plane / layer start
            int par[N > 0 ? N : 1];             // parity of the row start at plane 0
#pragma unroll
            for (int t = 0; t < N; ++t) {
                const int q = Q0 + t;
                if (q < U_INSTR) {              // pieces [64 q, 64 q + 64) of the (TY+1) x 98 piece image of the node plane
                    const int P = 64 * q + tz;
                    int r = P / 98, c = P - r * 98;
                    if (r > TY) { r = TY; c = 97; }
                    int jj = j0 + r;
                    jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
                    const long long rs = 3LL * ((long long) jj * d.NZ + k0u);
                    go[t] = rs + 2LL * c;
                    par[t] = (int) ((ubase8 + rs) & 1);
                } else {                        // pieces of the TY x 33 piece image of the element layer
                    const int P = 64 * (q - U_INSTR) + tz;
                    int r = P / 33, c = P - r * 33;
                    if (r > TY - 1) { r = TY - 1; c = 32; }
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
        if (ty == TY) dma_loop(std::integral_constant<int, 0>{});
        else dma_loop(std::integral_constant<int, (NW > 1 ? 1 : 0)>{});
        return;
    }
    // u: instruction j moves pieces [64 j, 64 j + 64) of the (TY+1) x 98 piece image; this wave owns j = wave, wave + TY
    long long ugo[2];          // double offset (from plane start) of this lane's row start + 2 q
    int upar[2];               // parity of (ubase8 + row start) at plane 0
    bool uhas[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j = wave + TY * s;
        uhas[s] = NW == 0 && j < U_INSTR;
        const int P = 64 * j + tz;
        int r = P / 98, q = P - r * 98;
        if (r > TY) { r = TY; q = 97; }
        int jj = j0 + r;
        jj = jj < 0 ? 0 : (jj > d.NY - 1 ? d.NY - 1 : jj);
        const long long rs = 3LL * ((long long) jj * d.NZ + k0u);
        ugo[s] = rs + 2LL * q;
        upar[s] = (int) ((ubase8 + rs) & 1);
    }
    // E: instruction e moves pieces [64 e, 64 e + 64) of the TY x 33 piece image (32 + 1 for the shift);
    // the E instructions go first to the waves that own a single u instruction (from the top), then to the waves below them
    const int eidx = (TY - 1 - wave) < E_INSTR ? (TY - 1 - wave) : -1;
    const bool ehas = NW == 0 && eidx >= 0;
    long long ego = 0;
    int epar0 = 0;
    const int k0e = k0u;                               // element columns start at max(k0, 0) as well
    {
        const int P = 64 * (eidx < 0 ? 0 : eidx) + tz;
        int r = P / 33, q = P - r * 33;
        if (r > TY - 1) { r = TY - 1; q = 32; }
        int jj = j0 + r;
        jj = jj < 0 ? 0 : (jj > d.ny - 1 ? d.ny - 1 : jj);
        const long long rs = (long long) jj * d.nz + k0e;
        ego = rs + 2LL * q;
        epar0 = (int) ((ebase8 + rs) & 1);
    }
    const int cnt = (uhas[1] ? 2 : 1) + (ehas ? 1 : 0);          // TY = 8: 2,2,2,3,3,3,2,2
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
        erpar = (int) ((ebase8 + (long long) jj * d.nz + k0e) & 1);
    }

    auto issue_plane = [&](int i, int sl) {            // node plane i + element layer min(i, nx-1) -> ring slot sl
        if (X == 6) return;
        unsigned char *slot = ring + (size_t) sl * SLOT_BYTES;
        const int ip = i & ppar;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (uhas[s]) {
                // aligned piece: one double below the row start when (row start + plane offset) is odd
                const long long off = 3LL * i * plane + ugo[s] - ((upar[s] + ip) & 1);
                const char *g = reinterpret_cast<const char *>(u + off);
                g = g > u_last ? u_last : g;
                glds16(g, slot + 1024 * (wave + TY * s));
            }
        }
        if (ehas) {
            const int il = i < d.nx ? i : d.nx - 1;
            const long long off = (long long) il * elayer + ego - ((epar0 + (il & epar)) & 1);
            const char *g = reinterpret_cast<const char *>(E + off);
            g = g > e_last ? e_last : g;
            glds16(g, slot + U_INSTR * 1024 + 1024 * eidx);
        }
    };

    // steady-state issue: per instruction slot two running pointers (planes of even / odd step parity), each advanced
    // by two plane strides per use; the alignment shift alternates with the plane only when a plane holds an odd number
    // of doubles, so it is folded into the two pointers once.  Only the last plane of the grid needs the end clamp.
    const char *up_run[2][2];
    const char *ep_run[2];
    auto init_running = [&](int i_first) {            // i_first = plane issued by the first loop phase (step parity 0)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = i_first + k;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                up_run[k][s2] = reinterpret_cast<const char *>(u + (3LL * i * plane + ugo[s2] - ((upar[s2] + (i & ppar)) & 1)));
            ep_run[k] = reinterpret_cast<const char *>(E + ((long long) i * elayer + ego - ((epar0 + (i & epar)) & 1)));
        }
    };
    auto issue_running = [&](int i, int k, int sl) {   // k = step parity (compile-time after unrolling), sl = ring slot
        if (X == 6) return;
        unsigned char *slot = ring + (size_t) sl * SLOT_BYTES;
        const bool last = (i >= d.NX - 1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (uhas[s2]) {
                const char *g = up_run[k][s2];
                if (EXP == 10) g -= 24LL * plane * (i & ~3);
                if (last) g = g > u_last ? u_last : g;
                glds16(g, slot + 1024 * (wave + TY * s2));
                up_run[k][s2] += 48LL * plane;         // two planes of 24 * plane bytes
            }
        }
        if (ehas) {
            const char *g = ep_run[k];
            if (EXP == 10) g -= 8LL * elayer * (i & ~3);
            if (i >= d.nx - 1) {                       // last layers: clamp the layer index and the address
                const long long off = (long long) (d.nx - 1) * elayer + ego - ((epar0 + ((d.nx - 1) & epar)) & 1);
                g = reinterpret_cast<const char *>(E + off);
                g = g > e_last ? e_last : g;
            }
            glds16(g, slot + U_INSTR * 1024 + 1024 * eidx);
            ep_run[k] += 16LL * elayer;
        }
    };

    // ---- per-thread read offsets into a slot (doubles) ----------------------------------------
    int c0 = tz - cshift, c1 = tz + 1 - cshift;
    c0 = c0 < 0 ? 0 : c0; c1 = c1 < 0 ? 0 : c1;
    const int o00 = ty * ROW_D + 3 * c0, o01 = ty * ROW_D + 3 * c1;
    const int o10 = (ty + 1) * ROW_D + 3 * c0, o11 = (ty + 1) * ROW_D + 3 * c1;
    int ce = tz - cshift; ce = ce < 0 ? 0 : ce;
    const int oE0 = (U_INSTR * 1024) / 8 + ty * 66 + ce;       // element rows hold 33 pieces = 66 doubles

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
            // the two terms owed to the z-neighbour move one lane up (DPP), only the sum owed to the y-neighbour (next wave)
            // goes through LDS: 3 instead of 9 doubles written and read per thread and plane
            wa[c] += lane_below(p + q);
            sX[(c * TY + ty) * TZ + tz] = (r - t) + lane_below(r + t);
        }
    };
    auto emit_plane = [&](int i, const double wa[3], int buf) {
        if (X == 5) return;
        if (EXP == 11 ? !(ty >= 1 && ej < d.NY) : !out_ok) return;
        const double *sX = sS + buf * SS_DOUBLES;
        const long long n = (long long) (EXP == 9 ? (i & 1) : i) * plane + (long long) ej * d.NZ + ek;
        double w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            w[c] = (X == 1 || X >= 3) ? wa[c] : wa[c] + sX[(c * TY + ty - 1) * TZ + tz];
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

    // ---- prologue: planes i_start .. i_start+RING-1 issued, all retired once ---------------------------
    // ring slot of plane i is (i - i_start) mod RING
#pragma unroll
    for (int r = 0; r < RING; ++r)
        if (i_start + r <= i_end) issue_plane(i_start + r, r);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    init_running(i_start + RING);
    const bool has_stores = X != 5 && ty >= 1 && ej < d.NY;

    int buf = 0, phase = 0, cur = 1 % RING;          // cur = ring slot of the plane consumed by the next phase
    auto run_phase = [&](int ii, int k) {
        // the slot of plane ii + PD held plane ii - 1: every wave passed a barrier after reading it
        if (ii + PD <= i_end) issue_running(ii + PD, k, cur == 0 ? RING - 1 : cur - 1);
        const double *su = reinterpret_cast<const double *>(ring + (size_t) cur * SLOT_BYTES);
        cur = cur + 1 == RING ? 0 : cur + 1;
        double wa[3];
        const double Enext = su[eoff2[k]];             // modulus of layer ii, used in the next phase
        process(Eprev, su, qoff[k], buf, wa);
        Eprev = Enext;
        if (ii + 1 <= i_end) {
            // counted wait: after plane ii+1's DMA this wave issued the stores of phases ii-2 and ii-1 and the DMA of
            // planes ii+2 and ii+3; outside that steady state (start / end of the chunk) drain everything
            const bool steady = (phase >= PD - 1) && (ii + PD <= i_end) && (ii - PD >= p0);
            if (NW == 0) { if (cnt == 2) wait_plane<2>(has_stores, steady); else wait_plane<3>(has_stores, steady); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ii - 1 >= p0) emit_plane(ii - 1, wa, buf);
        buf ^= 1;
        ++phase;
    };
    for (int ii = i_start + 1; ii <= i_end; ii += 2) {
        run_phase(ii, 0);
        if (ii + 1 <= i_end) run_phase(ii + 1, 1);
    }
    if (p1 == d.NX - 1) {
        double wa[3];
        scatter_face(carry, wa, buf);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        emit_plane(d.NX - 1, wa, buf);
    }
}

bool launch_apply_dma(const Dims &d, const double *Dm_host, const double *E, const double *E_alloc_end, const double *u,
                      double *out, hipStream_t s, int plane_lo, int plane_hi) {
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
    dim3 blk(TZ, TY + NW, 1), grd((np + ppc - 1) / ppc, (d.NZ + TZ - 2) / (TZ - 1), (d.NY + TY - 2) / (TY - 1));
    static bool attr = false;
    if (!attr) {
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_apply_dma<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_BYTES));
        attr = true;
    }
    // last admissible (aligned) piece: the one holding the last byte of each array
    auto last_piece = [](const char *end) { return reinterpret_cast<const char *>((reinterpret_cast<uintptr_t>(end) - 1) & ~(uintptr_t) 15); };
    extern int g_apply_skeleton;
#define VFEM_DMA_LAUNCH(X) k_apply_dma<X><<<grd, blk, LDS_BYTES, s>>>(d, dm, E, u, out, ppc, last_piece(u_end), last_piece(e_end), plane_lo, plane_hi)
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
    VFEM_HIP(hipGetLastError());
    return true;
}

}  // namespace vfem
