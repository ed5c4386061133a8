// Level-1 (virtual Galerkin) node row, evaluated per MIRROR CLASS instead of per incident element.
//
// Reference: the level-1 element matrices are Ke = sum_f E_f cK0[f], cK0[f] = I_f^T K0 I_f (MultigridSolver.hh:639-657), and a
// node row is the sum over the 8 incident coarse elements (MultigridSolver.hh:199-220).  With the mirror symmetry
//     cK0[f][(n,a),(m,b)] = s_a(f) s_b(f) cK0[0][(n^f,a),(m^f,b)]                      (kernels_mg.hip, k_gs_color_mf1_sym)
// write g = li ^ f for the child f of the element in which the node has local index li.  Then for the neighbour at offset
// o (w = |o| as a bit mask, x = bit 2)
//     block(o)[a][b] = sum_g  s_a(g) s_b(g) cK0[0][(g,a),(g^w,b)]  *  W_ab(g, o),
//     W_ab(g, o)     = sum over the elements d (d_t = side of the node along axis t) that contain the neighbour of
//                      sigma_a(d) sigma_b(d) E[fine element at offset (d_t ? g_t : -1 - g_t) from the node],   sigma_t = d_t ? +1 : -1
// so the 64 fine moduli around a node enter through sums and differences over the sides that SHARE a neighbour (faces: 4
// elements, edges: 2, the node itself: 8), and every coefficient of cK0[0] is used once per class instead of once per element:
// 8 x (26 x 9 + 63 + 9) = 2 500 multiply-adds and ~450 additions per node instead of 4 608 + 576 (tools/l1_merged_check.py
// checks the identity in numpy; tests/test_l1_merged_core.py checks THIS code on the host against the direct double sum).
//
// The work of a node is cut by the x offset of the neighbours: two SIDE parts (o_x = -1 / +1: 9 neighbours, the 32 moduli of
// that side) and the MID part (o_x = 0: 8 neighbours and the diagonal block, all 64 moduli, merged over x first).
#pragma once

#ifndef __HIPCC__
#define __host__
#define __device__
#define __forceinline__ inline
#endif

namespace vfem {
namespace l1m {

constexpr int TAB_ROW = 12;                            // 9 used
constexpr int TAB_DOUBLES = 8 * 8 * TAB_ROW;           // [g][w][3a+b], signs s_a(g) s_b(g) folded in

// cK0_0: the 24 x 24 matrix of child 0 (row-major)
inline void build_table(const double *cK0_0, double *tab) {
    for (int g = 0; g < 8; ++g)
        for (int w = 0; w < 8; ++w) {
            double *t = tab + (g * 8 + w) * TAB_ROW;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    const bool neg = (((g >> (2 - a)) ^ (g >> (2 - b))) & 1) != 0;
                    const double v = cK0_0[(3 * g + a) * 24 + 3 * (g ^ w) + b];
                    t[3 * a + b] = neg ? -v : v;
                }
            for (int q = 9; q < TAB_ROW; ++q) t[q] = 0.0;
        }
}

// index (fine offset + 2) of the class-g element on side d of the node along one axis
__host__ __device__ constexpr int class_index(int d, int g) { return d ? 2 + g : 1 - g; }
// the 2 x 2 moduli of class G in one fine x-plane given as plane[p_y][p_z]
template <int G>
__host__ __device__ __forceinline__ void class_window(const double (&plane)[4][4], double (&a)[2][2]) {
    constexpr int gy = (G >> 1) & 1, gz = G & 1;
    a[0][0] = plane[class_index(0, gy)][class_index(0, gz)]; a[0][1] = plane[class_index(0, gy)][class_index(1, gz)];
    a[1][0] = plane[class_index(1, gy)][class_index(0, gz)]; a[1][1] = plane[class_index(1, gy)][class_index(1, gz)];
}

// T[a][b] (+)= W_ab n[b]  with  W_aa = wd, W_xy = wxy, W_xz = wxz, W_yz = wyz  (callers pass compile-time signs as negations)
template <bool FIRST>
__host__ __device__ __forceinline__ void add_neighbour(double T[9], const double *n, double wd, double wxy, double wxz, double wyz) {
    if (FIRST) {
        T[0] = wd * n[0];  T[1] = wxy * n[1]; T[2] = wxz * n[2];
        T[3] = wxy * n[0]; T[4] = wd * n[1];  T[5] = wyz * n[2];
        T[6] = wxz * n[0]; T[7] = wyz * n[1]; T[8] = wd * n[2];
    } else {
        T[0] += wd * n[0];  T[1] += wxy * n[1]; T[2] += wxz * n[2];
        T[3] += wxy * n[0]; T[4] += wd * n[1];  T[5] += wyz * n[2];
        T[6] += wxz * n[0]; T[7] += wyz * n[1]; T[8] += wd * n[2];
    }
}
__host__ __device__ __forceinline__ void fold(double S[3], const double T[9], const double c[9]) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
    for (int a = 0; a < 3; ++a) {                      // three chained multiply-adds per row (a sum of products costs a multiply and an add more)
        S[a] = __builtin_fma(c[3 * a], T[3 * a], S[a]);
        S[a] = __builtin_fma(c[3 * a + 1], T[3 * a + 1], S[a]);
        S[a] = __builtin_fma(c[3 * a + 2], T[3 * a + 2], S[a]);
    }
}

// ---- SIDE part of mirror class G: neighbours (o_x = SIDE ? +1 : -1, o_y, o_z) -----------------------------------------------
// a[d_y][d_z]: the four moduli of the class on that side, fine offsets (d ? g : -1 - g) in y and z, in the fine x-plane at offset
//              (SIDE ? g_x : -1 - g_x)  -- see class_window()
// un[o_y + 1][3 (o_z + 1) + b]: the nine neighbours of that x-plane
template <int SIDE, int G, class Coef>
__host__ __device__ __forceinline__ void side_class(const double (&a)[2][2], const double (&un)[3][9], Coef &coef, double S[3]) {
    constexpr double sx = SIDE ? 1.0 : -1.0;
    double T[9], c[9];
    // w = 111: corners
    coef.template get<G, 7>(c);
    add_neighbour<true>(T, &un[0][0], a[0][0], sx * -1 * a[0][0], sx * -1 * a[0][0], a[0][0]);           // (o_y, o_z) = (-1, -1)
    add_neighbour<false>(T, &un[0][6], a[0][1], sx * -1 * a[0][1], sx * a[0][1], -a[0][1]);              // (-1, +1)
    add_neighbour<false>(T, &un[2][0], a[1][0], sx * a[1][0], sx * -1 * a[1][0], -a[1][0]);              // (+1, -1)
    add_neighbour<false>(T, &un[2][6], a[1][1], sx * a[1][1], sx * a[1][1], a[1][1]);                    // (+1, +1)
    fold(S, T, c);
    // w = 110: o_z = 0, merged over the z side
    const double Sz[2] = {a[0][0] + a[0][1], a[1][0] + a[1][1]}, Dz[2] = {a[0][1] - a[0][0], a[1][1] - a[1][0]};
    coef.template get<G, 6>(c);
    add_neighbour<true>(T, &un[0][3], Sz[0], sx * -1 * Sz[0], sx * Dz[0], -Dz[0]);
    add_neighbour<false>(T, &un[2][3], Sz[1], sx * Sz[1], sx * Dz[1], Dz[1]);
    fold(S, T, c);
    // w = 101: o_y = 0, merged over the y side
    const double Sy[2] = {a[0][0] + a[1][0], a[0][1] + a[1][1]}, Dy[2] = {a[1][0] - a[0][0], a[1][1] - a[0][1]};
    coef.template get<G, 5>(c);
    add_neighbour<true>(T, &un[1][0], Sy[0], sx * Dy[0], sx * -1 * Sy[0], -Dy[0]);
    add_neighbour<false>(T, &un[1][6], Sy[1], sx * Dy[1], sx * Sy[1], Dy[1]);
    fold(S, T, c);
    // w = 100: the face neighbour, merged over both
    coef.template get<G, 4>(c);
    add_neighbour<true>(T, &un[1][3], Sz[0] + Sz[1], sx * (Sz[1] - Sz[0]), sx * (Dz[0] + Dz[1]), Dz[1] - Dz[0]);
    fold(S, T, c);
}

// ---- MID part of mirror class G: neighbours (0, o_y, o_z) and the diagonal block; coefficient rows in the order w = 3, 2, 0, 1 ------
// a0 / a1 [d_y][d_z]: the class's moduli in the fine x-planes below (offset -1 - g_x) and above (offset g_x) the node; un as above
// for the node's own x-plane (un[1][3..5] = the node itself, not used here)
// PARTS: bit 0 the four neighbours (0, +-1, +-1), bit 1 (0, +-1, 0), bit 2 the diagonal block, bit 3 (0, 0, +-1) -- all of them
// (mid_class) or shared between two waves (the level-0 marching sweep: {0, 2} and {1, 2, 3}; the diagonal block then comes out of
// the same instructions in both, bit for bit)
template <int G, int PARTS, class Coef>
__host__ __device__ __forceinline__ void mid_class_parts(const double (&a0)[2][2], const double (&a1)[2][2], const double (&un)[3][9], Coef &coef,
                                                         double S[3], double M6[6]) {
    const double Sx[2][2] = {{a1[0][0] + a0[0][0], a1[0][1] + a0[0][1]}, {a1[1][0] + a0[1][0], a1[1][1] + a0[1][1]}};
    const double Dx[2][2] = {{a1[0][0] - a0[0][0], a1[0][1] - a0[0][1]}, {a1[1][0] - a0[1][0], a1[1][1] - a0[1][1]}};
    double T[9], c[9];
    if constexpr (PARTS & 1) {                 // w = 011
        coef.template get<G, 3>(c);
        add_neighbour<true>(T, &un[0][0], Sx[0][0], -Dx[0][0], -Dx[0][0], Sx[0][0]);
        add_neighbour<false>(T, &un[0][6], Sx[0][1], -Dx[0][1], Dx[0][1], -Sx[0][1]);
        add_neighbour<false>(T, &un[2][0], Sx[1][0], Dx[1][0], -Dx[1][0], -Sx[1][0]);
        add_neighbour<false>(T, &un[2][6], Sx[1][1], Dx[1][1], Dx[1][1], Sx[1][1]);
        fold(S, T, c);
    }
    if constexpr (PARTS & 6) {
        // merged over x and z
        const double P[2] = {Sx[0][0] + Sx[0][1], Sx[1][0] + Sx[1][1]}, Q[2] = {Dx[0][0] + Dx[0][1], Dx[1][0] + Dx[1][1]};
        const double R[2] = {Dx[0][1] - Dx[0][0], Dx[1][1] - Dx[1][0]}, U[2] = {Sx[0][1] - Sx[0][0], Sx[1][1] - Sx[1][0]};
        if constexpr (PARTS & 2) {             // w = 010
            coef.template get<G, 2>(c);
            add_neighbour<true>(T, &un[0][3], P[0], -Q[0], R[0], -U[0]);
            add_neighbour<false>(T, &un[2][3], P[1], Q[1], R[1], U[1]);
            fold(S, T, c);
        }
        if constexpr (PARTS & 4) {
            // w = 000: the diagonal block, merged over all eight elements (symmetric: M6 = {xx, xy, xz, yy, yz, zz}); before w = 001 so
            // that P, Q, R, U end here
            coef.template get<G, 0>(c);
            const double wd = P[0] + P[1], wxy = Q[1] - Q[0], wxz = R[0] + R[1], wyz = U[1] - U[0];
            M6[0] = __builtin_fma(c[0], wd, M6[0]);  M6[1] = __builtin_fma(c[1], wxy, M6[1]); M6[2] = __builtin_fma(c[2], wxz, M6[2]);
            M6[3] = __builtin_fma(c[4], wd, M6[3]);  M6[4] = __builtin_fma(c[5], wyz, M6[4]); M6[5] = __builtin_fma(c[8], wd, M6[5]);
        }
    }
    if constexpr (PARTS & 8) {                 // w = 001: merged over x and y
        const double P2[2] = {Sx[0][0] + Sx[1][0], Sx[0][1] + Sx[1][1]}, Q2[2] = {Dx[1][0] - Dx[0][0], Dx[1][1] - Dx[0][1]};
        const double R2[2] = {Dx[0][0] + Dx[1][0], Dx[0][1] + Dx[1][1]}, U2[2] = {Sx[1][0] - Sx[0][0], Sx[1][1] - Sx[0][1]};
        coef.template get<G, 1>(c);
        add_neighbour<true>(T, &un[1][0], P2[0], Q2[0], -R2[0], -U2[0]);
        add_neighbour<false>(T, &un[1][6], P2[1], Q2[1], R2[1], U2[1]);
        fold(S, T, c);
    }
}
template <int G, class Coef>
__host__ __device__ __forceinline__ void mid_class(const double (&a0)[2][2], const double (&a1)[2][2], const double (&un)[3][9], Coef &coef,
                                                   double S[3], double M6[6]) {
    mid_class_parts<G, 15>(a0, a1, un, coef, S, M6);
}

}  // namespace l1m
}  // namespace vfem
