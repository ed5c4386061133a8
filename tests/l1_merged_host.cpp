// Host harness for ndr_amd/csrc/l1_merged_core.h (tests/test_l1_merged_core.py): the merged evaluation of a level-1 node row
// against the direct double sum over the 8 incident elements x 8 children with mirror-image child matrices.
#include <cmath>
#include <cstdio>
#include <random>

#include "../ndr_amd/csrc/l1_merged_core.h"

using namespace vfem::l1m;

struct HostCoef {
    const double *tab;
    template <int G, int W> void get(double c[9]) const { for (int q = 0; q < 9; ++q) c[q] = tab[(G * 8 + W) * TAB_ROW + q]; }
};

template <int G>
static void all_classes(const double (&E)[4][4][4], const double (&u)[3][3][9], HostCoef &cf, double S[3], double M[6]) {
    double Es0[2][4][4], Es1[2][4][4];
    for (int gx = 0; gx < 2; ++gx)
        for (int y = 0; y < 4; ++y)
            for (int z = 0; z < 4; ++z) { Es0[gx][y][z] = E[1 - gx][y][z]; Es1[gx][y][z] = E[2 + gx][y][z]; }
    constexpr int gx = (G >> 2) & 1;
    double a[2][2], a0[2][2], a1[2][2];
    class_window<G>(Es0[gx], a);
    side_class<0, G>(a, u[0], cf, S);
    class_window<G>(E[1 - gx], a0);
    class_window<G>(E[2 + gx], a1);
    if (G & 1) mid_class<G>(a0, a1, u[1], cf, S, M);
    else {                                    // the two shares of the marching sweep's waves: together the whole, the diagonal block once
        double Mb[6] = {0, 0, 0, 0, 0, 0};
        mid_class_parts<G, 5>(a0, a1, u[1], cf, S, Mb);
        mid_class_parts<G, 14>(a0, a1, u[1], cf, S, M);
    }
    class_window<G>(Es1[gx], a);
    side_class<1, G>(a, u[2], cf, S);
    if constexpr (G + 1 < 8) all_classes<G + 1>(E, u, cf, S, M);
}

int main() {
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    double worst = 0.0;
    for (int trial = 0; trial < 20; ++trial) {
        double K[24][24];
        for (int i = 0; i < 24; ++i) for (int j = 0; j <= i; ++j) K[i][j] = K[j][i] = U(rng);
        double E[4][4][4], u[3][3][9];
        for (auto &p : E) for (auto &r : p) for (auto &v : r) v = 0.5 + 0.5 * U(rng);
        if (trial % 3 == 1) for (int y = 0; y < 4; ++y) for (int z = 0; z < 4; ++z) E[0][y][z] = E[1][y][z] = 0.0;   // a grid face
        for (auto &p : u) for (auto &r : p) for (auto &v : r) v = U(rng);
        double tab[TAB_DOUBLES];
        build_table(&K[0][0], tab);
        HostCoef cf{tab};
        double S[3] = {0, 0, 0}, M6[6] = {0};
        all_classes<0>(E, u, cf, S, M6);
        const double M[9] = {M6[0], M6[1], M6[2], M6[1], M6[3], M6[4], M6[2], M6[4], M6[5]};
        // direct
        double S0[3] = {0, 0, 0}, M0[9] = {0};
        auto bit = [](int v, int a) { return (v >> (2 - a)) & 1; };
        for (int d = 0; d < 8; ++d) {
            const int dd[3] = {bit(d, 0), bit(d, 1), bit(d, 2)};
            const int li = 7 - d;
            for (int f = 0; f < 8; ++f) {
                const double Ef = E[2 * dd[0] + bit(f, 0)][2 * dd[1] + bit(f, 1)][2 * dd[2] + bit(f, 2)];
                for (int m = 0; m < 8; ++m) {
                    const int o[3] = {dd[0] - 1 + bit(m, 0), dd[1] - 1 + bit(m, 1), dd[2] - 1 + bit(m, 2)};
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) {
                            const double sg = (bit(f, a) ^ bit(f, b)) ? -1.0 : 1.0;
                            const double c = Ef * sg * K[3 * (li ^ f) + a][3 * (m ^ f) + b];
                            if (m == li) M0[3 * a + b] += c;
                            else S0[a] += c * u[o[0] + 1][o[1] + 1][3 * (o[2] + 1) + b];
                        }
                }
            }
        }
        for (int a = 0; a < 3; ++a) worst = std::fmax(worst, std::fabs(S[a] - S0[a]));
        for (int q = 0; q < 9; ++q) worst = std::fmax(worst, std::fabs(M[q] - M0[q]));
    }
    std::printf("max_abs_error %.3e\n", worst);
    return worst < 1e-11 ? 0 : 1;
}
