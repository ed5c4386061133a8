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

}  // namespace vfem
