// Resident form of the reference element matrix for the level-0 Gauss-Seidel sweeps: the 36 distinct magnitudes of K0 in 72
// SGPRs, entry and sign selected at compile time (see the comment above build_gs_coef in kernels_mg.hip).
#pragma once
#include "device_utils.h"

namespace vfem {

struct KSel { int idx; bool neg; };
__host__ __device__ constexpr int kbit(int n, int d) { return (n >> (2 - d)) & 1; }
__host__ __device__ constexpr KSel ksel(int n, int a, int m, int b) {
    if (a == b) return KSel{a * 8 + (kbit(n, 0) == kbit(m, 0) ? 4 : 0) + (kbit(n, 1) == kbit(m, 1) ? 2 : 0) + (kbit(n, 2) == kbit(m, 2) ? 1 : 0), false};
    const int lo = a < b ? a : b, hi = a < b ? b : a, t = 3 - a - b;
    const bool t1 = kbit(n, lo) == kbit(m, hi);          // tau1 = s(n_lo) s(m_hi) = +1 iff the bits agree
    const bool t2 = kbit(n, hi) == kbit(m, lo);
    const int idx = 24 + (lo + hi - 1) * 4 + (kbit(n, t) == kbit(m, t) ? 2 : 0) + (t1 == t2 ? 1 : 0);
    return KSel{idx, !(a < b ? t1 : t2)};
}
struct GsCoef { d8_t c[4]; d4_t t; };

// the 36 resident coefficients into SGPRs (one wave-uniform load per wave)
template <bool RES>
__device__ __forceinline__ void gs_load_coef(const double *__restrict__ tab, GsCoef &ck) {
    if constexpr (RES) {
        asm volatile("s_load_dwordx16 %0, %5, 0x0\n\ts_load_dwordx16 %1, %5, 0x40\n\ts_load_dwordx16 %2, %5, 0x80\n\t"
                     "s_load_dwordx16 %3, %5, 0xc0\n\ts_load_dwordx8 %4, %5, 0x100\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(ck.c[0]), "=&s"(ck.c[1]), "=&s"(ck.c[2]), "=&s"(ck.c[3]), "=&s"(ck.t) : "s"(tab));
    }
}


// K0[(n,a),(m,b)] from the resident table, compile-time selection
template <int N, int A, int M, int B>
__device__ __forceinline__ double gs_coef_at(const GsCoef &ck) {
    constexpr KSel k = ksel(N, A, M, B);
    constexpr int i = k.idx;
    const double v = i < 32 ? ck.c[i < 32 ? i / 8 : 0][i < 32 ? i % 8 : 0] : ck.t[i >= 32 ? i - 32 : 0];
    return k.neg ? -v : v;
}


// The marching sweep (kernels_gs_march.hip) uses the coefficients in two parts, each of which needs 22 of the 36 magnitudes: the far
// planes pair a node with neighbours whose x bit DIFFERS, the node's own plane with neighbours whose x bit AGREES.  Holding all 72
// SGPRs across the kernel left too few for everything else (the compiler parked coefficients in VGPR lanes and fetched them back
// with ~110 v_readlane per colour), so each part loads its own compact table of 24 doubles just before its multiply-adds.
//   compact index: same component a: 4 a + (y-agree ? 2 : 0) + (z-agree ? 1 : 0);  pairs (x,y), (x,z): 12 + 4 pair + (idx & 3);
//   pair (y,z): 20 + (t1 == t2)   [its third axis is x, whose agreement is what selects the part]
__host__ __device__ constexpr int gs_part_index(int idx) {
    if (idx < 24) return (idx / 8) * 4 + (idx % 8 & 3);
    const int p = (idx - 24) / 4;
    return p < 2 ? 12 + 4 * p + (idx & 3) : 20 + (idx & 1);
}
// part 0: x bits differ, part 1: x bits agree
__host__ __device__ constexpr bool gs_in_part(int idx, int part) {
    if (idx < 24) return ((idx % 8 >> 2) & 1) == part;
    const int p = (idx - 24) / 4;
    return p < 2 ? true : (((idx >> 1) & 1) == part);
}
inline void build_gs_coef_parts(const double *coef36, double *far24, double *mid24) {
    for (int q = 0; q < 24; ++q) far24[q] = mid24[q] = 0.0;
    for (int idx = 0; idx < 36; ++idx) {
        if (gs_in_part(idx, 0)) far24[gs_part_index(idx)] = coef36[idx];
        if (gs_in_part(idx, 1)) mid24[gs_part_index(idx)] = coef36[idx];
    }
}
struct GsCoef24 { d8_t c[3]; };
// requested and awaited inside one asm statement (device_utils.h: sload12); the wait also covers the LDS reads in flight
__device__ __forceinline__ void gs_load_coef24(const double *__restrict__ tab, GsCoef24 &ck) {
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx16 %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(ck.c[0]), "=&s"(ck.c[1]), "=&s"(ck.c[2]) : "s"(tab) : "memory");
}
template <int PART, int N, int A, int M, int B>
__device__ __forceinline__ double gs_coef24_at(const GsCoef24 &ck) {
    constexpr KSel k = ksel(N, A, M, B);
    static_assert(gs_in_part(k.idx, PART), "coefficient does not belong to this part");
    constexpr int i = gs_part_index(k.idx);
    const double v = ck.c[i / 8][i % 8];
    return k.neg ? -v : v;
}

}  // namespace vfem
