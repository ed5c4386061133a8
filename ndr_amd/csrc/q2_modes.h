// Reflection-mode structure of the 27-node (degree-2) reference element, shared by the host packing code (generic.hip) and
// the pencil kernel (kernels_q2.hip).
//
// Per axis the three nodal values (at 0, 1/2, 1) are replaced by  s = v0 + v2,  m = v1,  a = v2 - v0  (slots 0, 1, 2 keep
// their places).  Under the reflection of that axis s and m are even, a is odd; the displacement component along the axis
// changes sign as well.  For a box element with an isotropic tensor K0 commutes with the three reflections, so in mode space
// it only couples (mode, component) pairs of equal combined parity P = parity(mode) xor e_component: eight diagonal blocks of
// sizes 12,12,12,9,12,9,9,6 (855 entries instead of 6561).   y = K0 u = T^T ( Kt (T u) ),  Kt = T^-T K0 T^-1.
#pragma once

namespace vfem {

struct Q2Classes {
    int n[8];           // block size of combined parity class P (bit 2 = x axis odd, bit 1 = y, bit 0 = z)
    int idx[8][12];     // member dofs (3 * mode + component, ascending), mode = 9 tx + 3 ty + tz with t = 0 (s), 1 (m), 2 (a)
    int rowbase[8];     // first row of the class in the packed coefficient table (one row = 12 doubles)
};

constexpr Q2Classes make_q2_classes() {
    Q2Classes q{};
    int rows = 0;
    for (int P = 0; P < 8; ++P) {
        q.n[P] = 0;
        q.rowbase[P] = rows;
        for (int j = 0; j < 12; ++j) q.idx[P][j] = 0;
        for (int dof = 0; dof < 81; ++dof) {
            const int mode = dof / 3, c = dof % 3;
            const int tx = mode / 9, ty = (mode / 3) % 3, tz = mode % 3;
            const int par = ((tx == 2) << 2) | ((ty == 2) << 1) | (tz == 2);
            if ((par ^ (4 >> c)) == P) q.idx[P][q.n[P]++] = dof;
        }
        rows += q.n[P];
    }
    return q;
}

static constexpr Q2Classes Q2C = make_q2_classes();
constexpr int Q2_TABLE_ROWS = 81;          // sum of the block sizes
constexpr int Q2_TABLE_DOUBLES = 81 * 12;

}  // namespace vfem
