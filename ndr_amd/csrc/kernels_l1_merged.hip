// Level-1 colour sweep / apply / residual with the node row evaluated per mirror class (l1_merged_core.h).
//
// Reference semantics: m_smoothNode MultigridSolver.hh:193-265 on the level-1 operator Ke = sum_f E_f cK0[f]
// (MultigridSolver.hh:639-657), applyK TensorProductSimulator.hh:905-952 for the residual.  Same operator as
// k_gs_color_mf1_sym / k_apply_gather<1> (kernels_mg.hip), different association of the sums: results agree to rounding.
//
// One node is the work of THREE waves (lane = node): the neighbours of the x-plane below (threadIdx.y = 0), of the node's own
// plane and the diagonal block (1), of the plane above (2).  Each wave holds its nine neighbours (54 registers) and the fine
// moduli it needs (32 / 64 / 32 doubles) in registers, walks the eight mirror classes with compile-time register indices and
// scalar-loaded coefficients (l1m::build_table, 6 KB: resident in the scalar cache), and the partial sums meet in LDS in a
// fixed order.  All loads are buffer loads: a neighbour or element outside the grid is given an out-of-range offset (or a
// zero-length plane) and reads as 0, so there is no clamping, no select on the loaded values and no divergent control flow.
// Per node ~3 150 multiply-adds + additions in ~80 vector loads against 5 184 + 576 in 256 loads of the per-element form.
#include "vfem_internal.h"
#include "device_utils.h"
#include "l1_merged_core.h"
#include "coef_rows.h"


namespace vfem {

namespace {

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

constexpr unsigned OOB = 0x7ffffff0u;                  // beyond any plane (launchers refuse planes of 2^31 bytes or more)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const double *base, long long plane_doubles, int plane, int planes) {
    const bool ok = plane >= 0 && plane < planes;
    const double *p = base + (long long) (ok ? plane : 0) * plane_doubles;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, ok ? (int) (plane_doubles * 8) : 0, 0x00020000);
}
__device__ __forceinline__ double mk(unsigned lo, unsigned hi) { return __longlong_as_double(((unsigned long long) hi << 32) | lo); }

// the 3 x 3 nodes (j + o_y, k + o_z) of one x-plane: un[o_y + 1][3 (o_z + 1) + c]
__device__ __forceinline__ void load_nodes(__amdgpu_buffer_rsrc_t r, const Dims &d, int j, int k, double (&un)[3][9]) {
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
        for (int oz = -1; oz <= 1; ++oz) {
            const int jj = j + oy, kk = k + oz;
            const bool ok = jj >= 0 && jj < d.NY && kk >= 0 && kk < d.NZ;
            const unsigned off = ok ? (unsigned) (jj * d.NZ + kk) * 24u : OOB;
#ifdef L1M_NOLOAD
            const u4_t v = {off, 0x3ff00000u, off, 0x3ff00000u}; const u2_t w = {off, 0x3ff00000u}; (void) r;
#else
            const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
            const u2_t w = __builtin_amdgcn_raw_buffer_load_b64(r, off + 16, 0, 0);
#endif
            un[oy + 1][3 * (oz + 1) + 0] = mk(v.x, v.y);
            un[oy + 1][3 * (oz + 1) + 1] = mk(v.z, v.w);
            un[oy + 1][3 * (oz + 1) + 2] = mk(w.x, w.y);
        }
}
// The moduli of a node group in LDS: sE[fine x-plane p_x][4 p_y + p_z][lane] (32 KB per workgroup, every lane reads back only what it
// wrote: no bank conflicts, no ordering beyond the one workgroup barrier).  The wave of the plane below loads the two fine planes below
// the node, the wave of the plane above the two above; the wave of the node's own plane -- whose sums run over all four -- loads
// none.  Windows of 2 x 2 moduli are read when a mirror class is reached, so a wave holds 4 (8) moduli in registers instead of 32 (64):
// 128 registers per wave, four waves per SIMD instead of two (the kernel is bound by the latency of its loads, not by bandwidth or
// arithmetic: profiles/r03_l1_merged.txt).
typedef double d2_t __attribute__((ext_vector_type(2)));
struct ModuliLds {
    double (*sE)[16][64];
    int lane;
    template <int G>
    __device__ __forceinline__ void window(int px, double (&a)[2][2]) const {
        constexpr int gy = (G >> 1) & 1, gz = G & 1;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dz = 0; dz < 2; ++dz) a[dy][dz] = sE[px][4 * l1m::class_index(dy, gy) + l1m::class_index(dz, gz)][lane];
    }
};
// the lane's 4 x 4 moduli of one fine x-plane, memory -> LDS
__device__ __forceinline__ void stage_moduli(__amdgpu_buffer_rsrc_t r, const unsigned (&off)[4][2], double (*plane)[64], int lane) {
    u4_t v[4][2];
#pragma unroll
    for (int py = 0; py < 4; ++py)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#ifdef L1M_NOLOAD
            v[py][h] = u4_t{off[py][h], 0x3ff00000u, off[py][h] + 1, 0x3ff00000u}; (void) r;
#else
            v[py][h] = __builtin_amdgcn_raw_buffer_load_b128(r, off[py][h], 0, 0);
#endif
        }
#pragma unroll
    for (int py = 0; py < 4; ++py)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            plane[4 * py + 2 * h][lane] = mk(v[py][h].x, v[py][h].y);
            plane[4 * py + 2 * h + 1][lane] = mk(v[py][h].z, v[py][h].w);
        }
}

template <int SIDE, int G>
__device__ __forceinline__ void side_classes(const ModuliLds &m, const double (&un)[3][9], DevCoef<false> &cf, double S[3]) {
#ifdef L1M_NOCOMP
    if constexpr (G == 0) { for (auto &r : un) for (double v : r) S[1] += v; }
#else
    double a[2][2];
    m.template window<G>(SIDE ? 2 + ((G >> 2) & 1) : 1 - ((G >> 2) & 1), a);
    l1m::side_class<SIDE, G>(a, un, cf, S);
#endif
    if constexpr (G + 1 < 8) side_classes<SIDE, G + 1>(m, un, cf, S);
}
template <int G>
__device__ __forceinline__ void mid_classes(const ModuliLds &m, const double (&un)[3][9], DevCoef<true> &cf, double S[3], double M[6]) {
#ifdef L1M_NOCOMP
    if constexpr (G == 0) { for (auto &r : un) for (double v : r) S[1] += v; M[0] = M[3] = M[5] = 1.0; }
#else
    double a0[2][2], a1[2][2];
    m.template window<G>(1 - ((G >> 2) & 1), a0);
    m.template window<G>(2 + ((G >> 2) & 1), a1);
    l1m::mid_class<G>(a0, a1, un, cf, S, M);
#endif
    if constexpr (G + 1 < 8) mid_classes<G + 1>(m, un, cf, S, M);
}

// One group of 64 nodes (lane = node (i, j, k); i uniform over the workgroup): three waves (role = 0: plane below, 1: own plane and the
// relaxation, 2: plane above).  MODE 0: relax in place; 1: out = A u; 2: out = b - A u, 0 at fixed components.  Two workgroup barriers.
template <int MODE>
__device__ __forceinline__ void node_group(const Dims &d, const double *__restrict__ tab, const double *__restrict__ E, const double *u,
                                           const double *__restrict__ b, const uint8_t *__restrict__ mask, double *out, int i, int j, int k,
                                           bool live, int role, int lane, int forward, double (&part)[2][3][64], double (*sE)[16][64]) {
    double un[3][9];
    load_nodes(plane_rsrc(u, 3LL * d.NY * d.NZ, i + role - 1, d.NX), d, j, k, un);
    // the wave of the node's own plane finishes the node: its right-hand side and mask are requested here, with the node values, not
    // behind the second barrier where nothing of this workgroup is left to hide them
    const long long n = ((long long) i * d.NY + j) * d.NZ + k;
    double bn[3] = {0.0, 0.0, 0.0};
    uint8_t mk8 = 0;
    if (MODE != 1 && role == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) bn[c] = b[3 * n + c];
        mk8 = mask ? mask[n] : 0;
    }
    if (role != 1) {
        // byte offsets of the lane's 4 x 2 pieces (16 B: two moduli) in a fine x-plane; pieces outside the grid read as 0
        unsigned eoff[4][2];
        const int nzf = 2 * d.nz;
#pragma unroll
        for (int py = 0; py < 4; ++py)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool ok = (py < 2 ? j >= 1 : j <= d.ny - 1) && (h == 0 ? k >= 1 : k <= d.nz - 1);
                eoff[py][h] = ok ? (unsigned) ((2 * j - 2 + py) * nzf + 2 * k - 2 + 2 * h) * 8u : OOB;
            }
        const long long eplane = 4LL * d.ny * d.nz;
        const int p0 = role == 0 ? 0 : 2;                      // fine planes 2 i - 2 + p; a side outside the grid has no planes
        const bool ok = role == 0 ? i > 0 : i < d.NX - 1;
        stage_moduli(plane_rsrc(E, eplane, ok ? 2 * i - 2 + p0 : -1, 2 * d.nx), eoff, sE[p0], lane);
        stage_moduli(plane_rsrc(E, eplane, ok ? 2 * i - 1 + p0 : -1, 2 * d.nx), eoff, sE[p0 + 1], lane);
    }
    __syncthreads();
    const ModuliLds m{sE, lane};
    double S[3] = {0.0, 0.0, 0.0}, M6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    // (a coefficient request in flight must not cross a branch: each role primes its own pipeline inside its branch)
    if (role == 1) {
        DevCoef<true> cf(tab);
#ifndef L1M_NOCOMP
        cf.prime();
#endif
        mid_classes<0>(m, un, cf, S, M6);
    } else if (role == 0) {
        DevCoef<false> cf(tab);
#ifndef L1M_NOCOMP
        cf.prime();
#endif
        side_classes<0, 0>(m, un, cf, S);
    } else {
        DevCoef<false> cf(tab);
#ifndef L1M_NOCOMP
        cf.prime();
#endif
        side_classes<1, 0>(m, un, cf, S);
    }
    if (role != 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) part[role >> 1][c][lane] = S[c];
    }
    __syncthreads();
    if (role != 1 || !live) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) S[c] = (part[0][c][lane] + S[c]) + part[1][c][lane];
    const double uc[3] = {un[1][3], un[1][4], un[1][5]};
    const double M[9] = {M6[0], M6[1], M6[2], M6[1], M6[3], M6[4], M6[2], M6[4], M6[5]};
    if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * n + c] = S[c] + (M[3 * c] * uc[0] + M[3 * c + 1] * uc[1] + M[3 * c + 2] * uc[2]);
        return;
    }
    double bms[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bms[c] = bn[c] - (S[c] + (M[3 * c] * uc[0] + M[3 * c + 1] * uc[1] + M[3 * c + 2] * uc[2]));
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * n + c] = ((mk8 >> c) & 1) ? 0.0 : bms[c];
        return;
    }
    double ud[3];
    gs_solve(bms, M, mk8, forward != 0, ud);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3 * n + c] = uc[c] + ud[c];
}

#ifndef L1M_WAVES
#define L1M_WAVES 3
#endif
template <int MODE>
__global__ void __launch_bounds__(192, L1M_WAVES) k_l1_merged(Dims d, const double *__restrict__ tab, const double *__restrict__ E,
                                                   const double *u, const double *__restrict__ b, const uint8_t *__restrict__ mask,
                                                   double *out, int cx, int cy, int cz, int forward, int ka, int kskip) {
    __shared__ double part[2][3][64];
    __shared__ double sE[4][16][64];     // (35 KB with the exchange area: four workgroups per CU.  Exactly 32 KB -- exchange area inside it, a
                                         // third barrier -- and five workgroups measured no faster: 2.91 against 2.88 ms per sweep at 257^3)
    const int lane = threadIdx.x, role = __builtin_amdgcn_readfirstlane(threadIdx.y);
    int i, j, k;
    bool live;
    if (MODE == 0) {       // lanes packed over the colour's nodes of an x-plane, row after row (rows have 2^k + 1 nodes)
        // (ka, kskip: the colour's nodes of a row without those of colour index ka .. ka + kskip - 1 -- what the row kernel below leaves over)
        const int cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1 - kskip;
        const int q = blockIdx.x * 64 + lane;
        live = q < cnty * cntz;
        const int qq = live ? q : cnty * cntz - 1, jq = qq / cntz, r = qq - jq * cntz;
        k = 2 * (r < ka ? r : r + kskip) + cz; j = 2 * jq + cy; i = 2 * blockIdx.z + cx;
    } else {
        const int q = blockIdx.x * 64 + lane;
        live = q < d.NY * d.NZ;
        const int qq = live ? q : d.NY * d.NZ - 1;
        j = qq / d.NZ; k = qq - j * d.NZ; i = blockIdx.z;
    }
    node_group<MODE>(d, tab, E, u, b, mask, out, i, j, k, live, role, lane, forward, part, sE);
}

// The two z colours of a row in one launch, in place: a workgroup owns the node row (i, j) and walks it in segments of 64 nodes of the
// colour relaxed first ("A", parity FZ) followed by the 64 nodes of the other colour below them ("B": node k_A - 1), each of which
// has both A neighbours of the row final by then; nodes of the pass touch no other row (other rows and planes belong to other
// colours), so workgroups are independent and the result is that of two launches, bit for bit.  What it saves is HBM traffic: a
// colour launch streams ALL fine moduli (their 4^3 neighbourhoods tile the fine grid: 1.07 GB at 257^3 of the 2.06 GB a launch
// moves); here the B phase finds the moduli and node rows of its segment in L2, where the A phase of the same workgroup left them.
// nseg whole segments per row; what they leave over of the row goes to k_l1_merged (A nodes before, B nodes after).
template <int FZ>
__global__ void __launch_bounds__(192, L1M_WAVES) k_l1_pair_rows(Dims d, const double *__restrict__ tab, const double *__restrict__ E, double *u,
                                                                 const double *__restrict__ b, const uint8_t *__restrict__ mask, int cx, int cy,
                                                                 int nseg) {
    __shared__ double part[2][3][64];
    __shared__ double sE[4][16][64];
    const int lane = threadIdx.x, role = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int i = 2 * blockIdx.z + cx, j = 2 * blockIdx.y + cy;
#pragma unroll 1
    for (int s = 0; s < 2 * nseg; ++s) {           // even s: the A nodes of segment s / 2, odd s: its B nodes
        // FZ = 0: the A nodes are the even ones from k = 2 on (k = 0 is left over, relaxed before the launch), so that the B nodes
        // 1, 3, ... fill their 64 lanes and a row of 2^m + 1 nodes leaves nothing over at its end
        const int k = (FZ == 0 ? 2 : 1) + 128 * (s >> 1) + 2 * lane - (s & 1);
        const bool live = k >= 0 && k < d.NZ;
        node_group<0>(d, tab, E, u, b, mask, u, i, j, live ? k : FZ, live, role, lane, FZ == 0, part, sE);
        // what was stored is read by the next phase's loads of all three waves (same CU: the vector cache is shared and write-through)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

}  // namespace

void build_l1_merged_table(const double *cK0_0, double *tab /* L1M_TABLE_DOUBLES */) { l1m::build_table(cK0_0, tab); }

// planes are addressed with 32-bit byte offsets
bool l1_merged_usable(const Dims &d) {
    return d.nx >= 1 && d.ny >= 1 && d.nz >= 1 && 24LL * d.NY * d.NZ < 0x7f000000LL && 32LL * d.ny * d.nz < 0x7f000000LL;
}

void launch_l1_merged_sweep(const Dims &d, const double *tab, const double *E, double *u, const double *b, const uint8_t *mask,
                            int forward, int xparity, int first, int count, hipStream_t s, int pair) {
    for (int ci = first; ci < first + count; ++ci) {
        const int lni = forward ? ci : 7 - ci;
        const int cx = ((lni >> 2) & 1) ^ (xparity & 1), cy = (lni >> 1) & 1, cz = lni & 1;
        if (cx > d.NX - 1 || cy > d.NY - 1 || cz > d.NZ - 1) continue;
        const int cntx = (d.NX - 1 - cx) / 2 + 1, cnty = (d.NY - 1 - cy) / 2 + 1, cntz = (d.NZ - 1 - cz) / 2 + 1;
        // the row kernel walks the A nodes of colour index ka .. ka + 64 nseg - 1 (ka = 1 for even A nodes: see there) and the B nodes
        // 0 .. 64 nseg - 1
        const int fz = cz, ka = fz == 0 ? 1 : 0, nseg = (cntz - ka) / 64;
        if (pair && ci % 2 == 0 && ci + 1 < first + count && nseg >= 1) {
            // colours ci and ci + 1 differ in c_z only (c_z = fz first): whole segments by the row kernel, the nodes it leaves over
            // colour by colour: first colour before it, second colour after it (a row of 2^m + 1 nodes: one launch of one node per row)
            const int walked = 64 * nseg, cntb = (d.NZ - 1 - (1 - fz)) / 2 + 1;
            if (cntz > walked)
                k_l1_merged<0><<<dim3((cnty * (cntz - walked) + 63) / 64, 1, cntx), dim3(64, 3, 1), 0, s>>>(d, tab, E, u, b, mask, u, cx, cy, fz, forward, ka, walked);
            if (fz == 0) k_l1_pair_rows<0><<<dim3(1, cnty, cntx), dim3(64, 3, 1), 0, s>>>(d, tab, E, u, b, mask, cx, cy, nseg);
            else         k_l1_pair_rows<1><<<dim3(1, cnty, cntx), dim3(64, 3, 1), 0, s>>>(d, tab, E, u, b, mask, cx, cy, nseg);
            if (cntb > walked)
                k_l1_merged<0><<<dim3((cnty * (cntb - walked) + 63) / 64, 1, cntx), dim3(64, 3, 1), 0, s>>>(d, tab, E, u, b, mask, u, cx, cy, 1 - fz, forward, 0, walked);
            ++ci;
            continue;
        }
        k_l1_merged<0><<<dim3((cnty * cntz + 63) / 64, 1, cntx), dim3(64, 3, 1), 0, s>>>(d, tab, E, u, b, mask, u, cx, cy, cz, forward, 0, 0);
    }
    VFEM_HIP(hipGetLastError());
}

void launch_l1_merged_apply(const Dims &d, const double *tab, const double *E, const double *u, const double *b, const uint8_t *mask,
                            int res, double *out, hipStream_t s) {
    const dim3 grd((d.NY * d.NZ + 63) / 64, 1, d.NX), blk(64, 3, 1);
    if (res) k_l1_merged<2><<<grd, blk, 0, s>>>(d, tab, E, u, b, mask, out, 0, 0, 0, 0, 0, 0);
    else     k_l1_merged<1><<<grd, blk, 0, s>>>(d, tab, E, u, b, mask, out, 0, 0, 0, 0, 0, 0);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
