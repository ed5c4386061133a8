// Degree-2 (27-node hexahedron) matrix-free stiffness apply and sensitivity: TensorProductSimulator<2,2,2>
// (reference: VoxelFEM/TensorProductSimulator.hh:905-952 applyK, :730-751 complianceGradient, instantiated with
// Degrees = 2,2,2; the reference leaves it unbound, VoxelFEM.cc:226-229).
//
// Node grid (2nx+1) x (2ny+1) x (2nz+1), element (i,j,k) owns nodes (2i+a, 2j+b, 2k+c), local index 9a+3b+c.
// Gather form, one launch per node class (parity of the node coordinate per axis): an even coordinate lies on an
// element boundary (two incident elements along that axis), an odd one is a mid node (one element).  Inside a class
// the local index of the node in each incident element is the same for every lane, so the K0 rows are wave-uniform
// (scalar loads).  First correct version: dense 81x81 reference matrix, fp64-FMA-bound (6561 FMA per voxel).
#include "vfem_internal.h"
#include "device_utils.h"
#include "q2_modes.h"

namespace vfem {

struct DimsQ2 { int nx, ny, nz, NX, NY, NZ; };   // elements / nodes per dim (N = 2n + 1)

__global__ void __launch_bounds__(256) k_apply_q2(DimsQ2 d, const double *__restrict__ K0, const double *__restrict__ E,
                                                  const double *__restrict__ u, double *__restrict__ out, int px, int py, int pz) {
    // node (i, j, k) = (2 a + px, 2 b + py, 2 c + pz)
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y, a = blockIdx.z;
    const int i = 2 * a + px, j = 2 * b + py, k = 2 * c + pz;
    if (i >= d.NX || j >= d.NY || k >= d.NZ) return;
    double S0 = 0.0, S1 = 0.0, S2 = 0.0;
    const int nex = px ? 1 : 2, ney = py ? 1 : 2, nez = pz ? 1 : 2;
    for (int sx = 0; sx < nex; ++sx) {
        // mid node: element a, local 1; boundary node: elements a-1 (local 2) and a (local 0)
        const int ex = px ? a : a - 1 + sx, lx = px ? 1 : (sx ? 0 : 2);
        if (ex < 0 || ex >= d.nx) continue;
        for (int sy = 0; sy < ney; ++sy) {
            const int ey = py ? b : b - 1 + sy, ly = py ? 1 : (sy ? 0 : 2);
            if (ey < 0 || ey >= d.ny) continue;
            for (int sz = 0; sz < nez; ++sz) {
                const int ez = pz ? c : c - 1 + sz, lz = pz ? 1 : (sz ? 0 : 2);
                if (ez < 0 || ez >= d.nz) continue;
                const int ln = 9 * lx + 3 * ly + lz;
                const double Ee = E[((long long) ex * d.ny + ey) * d.nz + ez];
                const double *r0 = K0 + (3 * ln) * 81, *r1 = r0 + 81, *r2 = r1 + 81;
                double t0 = 0.0, t1 = 0.0, t2 = 0.0;
                for (int ma = 0; ma < 3; ++ma)
                    for (int mb = 0; mb < 3; ++mb) {
                        const long long rowbase = ((long long) (2 * ex + ma) * d.NY + (2 * ey + mb)) * d.NZ + 2 * ez;
#pragma unroll
                        for (int mc = 0; mc < 3; ++mc) {
                            const int m = 9 * ma + 3 * mb + mc;
                            const double *um = u + 3 * (rowbase + mc);
                            const double u0 = um[0], u1 = um[1], u2 = um[2];
                            t0 = fma(r0[3 * m], u0, fma(r0[3 * m + 1], u1, fma(r0[3 * m + 2], u2, t0)));
                            t1 = fma(r1[3 * m], u0, fma(r1[3 * m + 1], u1, fma(r1[3 * m + 2], u2, t1)));
                            t2 = fma(r2[3 * m], u0, fma(r2[3 * m + 1], u1, fma(r2[3 * m + 2], u2, t2)));
                        }
                    }
                S0 = fma(Ee, t0, S0); S1 = fma(Ee, t1, S1); S2 = fma(Ee, t2, S2);
            }
        }
    }
    const long long n = ((long long) i * d.NY + j) * d.NZ + k;
    out[3 * n] = S0; out[3 * n + 1] = S1; out[3 * n + 2] = S2;
}

void launch_apply_q2(int nx, int ny, int nz, const double *K0, const double *E, const double *u, double *out, hipStream_t s) {
    DimsQ2 d{nx, ny, nz, 2 * nx + 1, 2 * ny + 1, 2 * nz + 1};
    for (int cls = 0; cls < 8; ++cls) {
        const int px = (cls >> 2) & 1, py = (cls >> 1) & 1, pz = cls & 1;
        const int cx = (d.NX - 1 - px) / 2 + 1, cy = (d.NY - 1 - py) / 2 + 1, cz = (d.NZ - 1 - pz) / 2 + 1;
        dim3 blk(64, 4, 1), grd((cz + 63) / 64, (cy + 3) / 4, cx);
        k_apply_q2<<<grd, blk, 0, s>>>(d, K0, E, u, out, px, py, pz);
    }
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------
// Pencil kernel: the production degree-2 apply.
//
// A wave owns one z-pencil of elements (ex, ey fixed), 64 elements per chunk, one element per lane.  The nine node rows
// of the pencil are staged through LDS with dense loads; the element's 81 values live in registers and are transformed
// in place to reflection modes (q2_modes.h), multiplied by the eight diagonal blocks of the mode-space reference matrix
// (855 instead of 6561 multiply-adds, coefficients by scalar loads from a 7.8 KB table), scaled by the modulus and
// transformed back.  Contributions to the node plane shared with the next element in z move one lane up by shuffle
// (a per-wave LDS slot carries them across chunks); the nine rows are then added to `out` with dense stores.
// Rows are shared between neighbouring pencils in x and y, so the pencils are launched in four colours (ex, ey parity)
// in a fixed order: a row is stored by the first pencil that touches it and read-modify-written by the later ones --
// no atomics, no zero fill, a fixed summation order.
// Traffic per voxel: 9 u rows + 9 out-row writes + 5 out-row reads (2 nodes each) = 1104 B against 393 B algorithmic.
// ------------------------------------------------------------------------------------------------------
constexpr int Q2_BUF = 448;

__global__ void __launch_bounds__(256) k_apply_q2_pencil(DimsQ2 d, const double *__restrict__ tab, const double *__restrict__ E,
                                                         const double *__restrict__ u, double *__restrict__ out, int cx, int cy) {
    __shared__ double lds[4][Q2_BUF];
    __shared__ double ldsc[4][32];
    __shared__ double ldso[4][9 * 384];                       // partial sums already in `out` (rows an earlier colour has written)
    const int lane = threadIdx.x, wy = threadIdx.y;
    const int ex = 2 * blockIdx.z + cx, ey = 2 * (blockIdx.y * 4 + wy) + cy;
    if (ex >= d.nx || ey >= d.ny) return;                    // wave-uniform; no block-level barrier below
    double *buf = lds[wy], *cbuf = ldsc[wy], *obuf = ldso[wy];
    const int nchunk = (d.nz + 63) / 64;
    const long long rowlen = 3LL * d.NZ;

    // does an earlier launch (colour order (0,0),(0,1),(1,0),(1,1)) already hold a partial sum for row (rx, ry)?
    unsigned rmw_mask = 0;
    static_for<9>([&](auto rc) {
        constexpr int rx = decltype(rc)::value / 3, ry = decltype(rc)::value % 3;
        bool earlier = false;
        for (int sx = 0; sx < 2; ++sx)
            for (int sy = 0; sy < 2; ++sy) {
                if (sx == 0 && sy == 0) continue;
                const int dx = sx ? (rx == 0 ? -1 : (rx == 2 ? 1 : 0)) : 0, dy = sy ? (ry == 0 ? -1 : (ry == 2 ? 1 : 0)) : 0;
                if ((sx && dx == 0) || (sy && dy == 0)) continue;
                const int ox = ex + dx, oy = ey + dy;
                if (ox < 0 || ox >= d.nx || oy < 0 || oy >= d.ny) continue;
                const int ocx = cx ^ (dx != 0), ocy = cy ^ (dy != 0);
                if (2 * ocx + ocy < 2 * cx + cy) earlier = true;
            }
        if (earlier) rmw_mask |= 1u << (3 * rx + ry);
    });
    if (lane < 27) cbuf[lane] = 0.0;

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int ez = chunk * 64 + lane;
        const long long seg = 3LL * 128 * chunk;             // first double of the chunk inside a node row
        int qc[7];
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) {
            long long q = seg + lane + 64 * s7;
            q = q > rowlen - 1 ? rowlen - 1 : q;
            qc[s7] = (int) (q - seg);
        }
        double v[81];
        // every global read of the chunk is issued up front (the element registers are still empty, so the loads have room):
        // nine u rows, and the rows of `out` that already hold the partial sums of an earlier colour (parked in LDS)
        double pre[9][7];
        static_for<9>([&](auto rc) {
            constexpr int r9 = decltype(rc)::value;
            const long long ro = 3LL * (((long long) (2 * ex + r9 / 3) * d.NY + (2 * ey + r9 % 3)) * d.NZ) + seg;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) (ro & 0xffffffffLL));
            const int hi = __builtin_amdgcn_readfirstlane((int) (ro >> 32));
            const double *rowp = u + (((long long) hi << 32) | (long long) lo);
#pragma unroll
            for (int s7 = 0; s7 < 7; ++s7) pre[r9][s7] = rowp[qc[s7]];
        });
        static_for<9>([&](auto rc) {
            constexpr int g = decltype(rc)::value;
            if ((rmw_mask >> g) & 1) {
                const long long ro = 3LL * (((long long) (2 * ex + g / 3) * d.NY + (2 * ey + g % 3)) * d.NZ) + seg;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) (ro & 0xffffffffLL));
                const int hi = __builtin_amdgcn_readfirstlane((int) (ro >> 32));
                const double *rowp = out + (((long long) hi << 32) | (long long) lo);
                double t[6];
#pragma unroll
                for (int s6 = 0; s6 < 6; ++s6) t[s6] = rowp[qc[s6]];
#pragma unroll
                for (int s6 = 0; s6 < 6; ++s6) obuf[g * 384 + lane + 64 * s6] = t[s6];
            }
        });
        static_for<9>([&](auto rc) {
            constexpr int r9 = decltype(rc)::value;
#pragma unroll
            for (int s7 = 0; s7 < 7; ++s7) buf[lane + 64 * s7] = pre[r9][s7];
            __builtin_amdgcn_wave_barrier();
            static_for<9>([&](auto qq) { constexpr int q = decltype(qq)::value; v[3 * (3 * r9 + q / 3) + q % 3] = buf[6 * lane + q]; });
            __builtin_amdgcn_wave_barrier();
        });
        const bool elem_ok = ez < d.nz;
        const int ezc = elem_ok ? ez : d.nz - 1;
        const double Ev = E[((long long) ex * d.ny + ey) * d.nz + ezc];
        const double Ee = elem_ok ? Ev : 0.0;

        // forward butterflies (z, y, x): slots 0, 1, 2 <- s, m, a
        static_for<27>([&](auto tc) {
            constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;                  // g = 3a+b, nodes 3g+{0,1,2}
            const double v0 = v[3 * (3 * g) + c], v2 = v[3 * (3 * g + 2) + c];
            v[3 * (3 * g) + c] = v0 + v2; v[3 * (3 * g + 2) + c] = v2 - v0;
        });
        static_for<27>([&](auto tc) {
            constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3, a = g / 3, cz = g % 3;   // nodes 9a + 3{0,1,2} + cz
            const double v0 = v[3 * (9 * a + cz) + c], v2 = v[3 * (9 * a + 6 + cz) + c];
            v[3 * (9 * a + cz) + c] = v0 + v2; v[3 * (9 * a + 6 + cz) + c] = v2 - v0;
        });
        static_for<27>([&](auto tc) {
            constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;                  // nodes {0,9,18} + g
            const double v0 = v[3 * g + c], v2 = v[3 * (18 + g) + c];
            v[3 * g + c] = v0 + v2; v[3 * (18 + g) + c] = v2 - v0;
        });
        // block-diagonal mode-space matrix, in place
        static_for<8>([&](auto pc) {
            constexpr int P = decltype(pc)::value, n = Q2C.n[P];
            double z[12];
            static_for<n>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                d8_t c0;
                d4_t c1;
                sload12(tab, (Q2C.rowbase[P] + i) * 96, c0, c1);
                double acc = 0.0;
                static_for<n>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    acc = fma(j < 8 ? c0[j < 8 ? j : 0] : c1[j < 8 ? 0 : j - 8], v[Q2C.idx[P][j]], acc);
                });
                z[i] = acc;
                asm volatile("" : "+v"(z[i]));           // retire before the next coefficient row is requested
            });
            static_for<n>([&](auto ic) { constexpr int i = decltype(ic)::value; v[Q2C.idx[P][i]] = Ee * z[i]; });
        });
        // transposed butterflies (x, y, z): y0 = zs - za, y2 = zs + za
        static_for<27>([&](auto tc) {
            constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;
            const double zs = v[3 * g + c], za = v[3 * (18 + g) + c];
            v[3 * g + c] = zs - za; v[3 * (18 + g) + c] = zs + za;
        });
        static_for<27>([&](auto tc) {
            constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3, a = g / 3, cz = g % 3;
            const double zs = v[3 * (9 * a + cz) + c], za = v[3 * (9 * a + 6 + cz) + c];
            v[3 * (9 * a + cz) + c] = zs - za; v[3 * (9 * a + 6 + cz) + c] = zs + za;
        });
        static_for<27>([&](auto tc) {
            constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;
            const double zs = v[3 * (3 * g) + c], za = v[3 * (3 * g + 2) + c];
            v[3 * (3 * g) + c] = zs - za; v[3 * (3 * g + 2) + c] = zs + za;
        });
        // node plane shared with the next element in z: one lane up; lane 0 takes what lane 63 left in the previous chunk
        static_for<27>([&](auto tc) {
            constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;                  // row g = 3 rx + ry
            const double top = v[3 * (3 * g + 2) + c];
            const double up = __shfl_up(top, 1);
            const double prev = cbuf[t];
            v[3 * (3 * g) + c] += lane == 0 ? prev : up;
            if (lane == 63) cbuf[t] = top;
        });
        // the last element of the pencil also owns the final node plane (z = 2 nz)
        if (ez == d.nz - 1) {
            static_for<9>([&](auto rc) {
                constexpr int g = decltype(rc)::value;
                double *np = out + 3LL * ((((long long) (2 * ex + g / 3) * d.NY + (2 * ey + g % 3)) * d.NZ) + 2 * d.nz);
                const bool rmw = (rmw_mask >> g) & 1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double add = v[3 * (3 * g + 2) + c];
                    np[c] = rmw ? np[c] + add : add;
                }
            });
        }
        // rows out: bottom and middle node of every lane, dense stores; read-modify-write where an earlier colour has written
        static_for<9>([&](auto rc) {
            constexpr int g = decltype(rc)::value;
            static_for<6>([&](auto qq) { constexpr int q = decltype(qq)::value; buf[6 * lane + q] = v[3 * (3 * g + q / 3) + q % 3]; });
            __builtin_amdgcn_wave_barrier();
            const long long ro = 3LL * (((long long) (2 * ex + g / 3) * d.NY + (2 * ey + g % 3)) * d.NZ) + seg;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) (ro & 0xffffffffLL));
            const int hi = __builtin_amdgcn_readfirstlane((int) (ro >> 32));
            double *rowp = out + (((long long) hi << 32) | (long long) lo);
            const bool rmw = (rmw_mask >> g) & 1;
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) {
                const int q = lane + 64 * s6;
                if (seg + q < rowlen - 3) {                 // the final node plane is written by the last element above
                    const double add = buf[q];
                    rowp[q] = rmw ? obuf[g * 384 + q] + add : add;
                }
            }
            __builtin_amdgcn_wave_barrier();
        });
    }
}

void launch_apply_q2_pencil(int nx, int ny, int nz, const double *tab, const double *E, const double *u, double *out, hipStream_t s) {
    DimsQ2 d{nx, ny, nz, 2 * nx + 1, 2 * ny + 1, 2 * nz + 1};
    for (int cx = 0; cx < 2; ++cx)
        for (int cy = 0; cy < 2; ++cy) {
            if (cx > nx - 1 || cy > ny - 1) continue;
            const int cntx = (nx - 1 - cx) / 2 + 1, cnty = (ny - 1 - cy) / 2 + 1;
            k_apply_q2_pencil<<<dim3(1, (cnty + 3) / 4, cntx), dim3(64, 4, 1), 0, s>>>(d, tab, E, u, out, cx, cy);
        }
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------
// Marching kernel: the pencil kernel turned into an x-march.
//
// A wave owns (ey, a chunk of 63 elements in z) and walks ex through one x-chunk.  The node plane an element shares with its
// successor in x never leaves the wave: its raw values (`keep`) are the a = 0 inputs of the next step and its output
// contributions (`carry`) are added to the next step's a = 0 rows before those are stored, so per step only six u rows are
// read and at most six finished out rows are written.  The four waves of a block own four consecutive element rows in y: a wave
// hands the rows it shares with the wave above to that wave through LDS (one block barrier per step), so only blocks still need
// colours in y (2 launches; the second one read-modify-writes the two rows per plane it shares with the blocks next to it).  Lane 0 recomputes the last element of
// the z-chunk below (its top-plane contributions reach lane 1 by the same one-lane shift as inside a chunk), and an x-chunk
// that does not start at the domain face first runs the element in front of it for its carry only -- no exchange between waves.
// The loads of step ex+1 (six u rows, the rows to be read-modify-written, the modulus) are issued before the arithmetic of step ex.
// Traffic per voxel: 4.5 u rows (shared rows once per block) + 4.5 out-row writes + 0.5 out-row reads = 464 B against 393 B algorithmic.
// ------------------------------------------------------------------------------------------------------
constexpr int Q2M_ZS = 63;                                   // elements a wave completes per step

__device__ __forceinline__ const double *q2_row(const double *base, const DimsQ2 &d, int X, int Y) {
    const long long ro = 3LL * (((long long) X * d.NY + Y) * d.NZ);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) (ro & 0xffffffffLL));
    const int hi = __builtin_amdgcn_readfirstlane((int) (ro >> 32));
    return base + (((long long) hi << 32) | (long long) lo);
}

// in-place element product in reflection-mode space: v <- Ee * K0 v (81 values, node-major 3 components)
constexpr int q2_row_class(int r) { int P = 0; while (P < 7 && r >= Q2C.rowbase[P + 1]) ++P; return P; }

// TABLE 0: one scalar-load round trip per coefficient row; 1: rows read from LDS; 2: scalar loads, the next row requested
// while the current one is multiplied (the wave runs alone on its SIMD, nothing else hides the round trip)
template <int TABLE>
__device__ __forceinline__ void q2_element_product(double (&v)[81], const double *tab, double Ee) {
    constexpr bool LDS_TABLE = TABLE == 1;
    static_for<27>([&](auto tc) {
        constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;
        const double v0 = v[3 * (3 * g) + c], v2 = v[3 * (3 * g + 2) + c];
        v[3 * (3 * g) + c] = v0 + v2; v[3 * (3 * g + 2) + c] = v2 - v0;
    });
    static_for<27>([&](auto tc) {
        constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3, a = g / 3, cz = g % 3;
        const double v0 = v[3 * (9 * a + cz) + c], v2 = v[3 * (9 * a + 6 + cz) + c];
        v[3 * (9 * a + cz) + c] = v0 + v2; v[3 * (9 * a + 6 + cz) + c] = v2 - v0;
    });
    static_for<27>([&](auto tc) {
        constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;
        const double v0 = v[3 * g + c], v2 = v[3 * (18 + g) + c];
        v[3 * g + c] = v0 + v2; v[3 * (18 + g) + c] = v2 - v0;
    });
    if constexpr (TABLE == 2) {
        d8_t A0, B0;
        d4_t A1, B1;
        double z[12];
        sload12_issue(tab, 0, A0, A1, v[0]);
        auto one_row = [&](auto rc, d8_t &c0, d4_t &c1, d8_t &n0, d4_t &n1) {
            constexpr int r = decltype(rc)::value, P = q2_row_class(r), n = Q2C.n[P], i = r - Q2C.rowbase[P];
            sload12_wait(c0, c1);
            double acc = 0.0;
            if constexpr (r + 1 < Q2_TABLE_ROWS) sload12_issue(tab, (r + 1) * 96, n0, n1, acc);
            static_for<n>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                acc = fma(j < 8 ? c0[j < 8 ? j : 0] : c1[j < 8 ? 0 : j - 8], v[Q2C.idx[P][j]], acc);
            });
            z[i] = acc;
            asm volatile("" : "+v"(z[i]));
            if constexpr (i == n - 1)
                static_for<n>([&](auto ic) { constexpr int ii = decltype(ic)::value; v[Q2C.idx[P][ii]] = Ee * z[ii]; });
        };
        static_for<Q2_TABLE_ROWS>([&](auto rc) {
            if constexpr (decltype(rc)::value % 2 == 0) one_row(rc, A0, A1, B0, B1);
            else one_row(rc, B0, B1, A0, A1);
        });
    } else
    static_for<8>([&](auto pc) {
        constexpr int P = decltype(pc)::value, n = Q2C.n[P];
        double z[12];
        static_for<n>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            double acc = 0.0;
            if constexpr (LDS_TABLE) {                       // same address in every lane (broadcast); the compiler pipelines the reads
                const double *row = tab + (Q2C.rowbase[P] + i) * 12;
                static_for<n>([&](auto jc) { constexpr int j = decltype(jc)::value; acc = fma(row[j], v[Q2C.idx[P][j]], acc); });
                z[i] = acc;
            } else {
                d8_t c0;
                d4_t c1;
                sload12(tab, (Q2C.rowbase[P] + i) * 96, c0, c1);
                static_for<n>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    acc = fma(j < 8 ? c0[j < 8 ? j : 0] : c1[j < 8 ? 0 : j - 8], v[Q2C.idx[P][j]], acc);
                });
                z[i] = acc;
                asm volatile("" : "+v"(z[i]));
            }
        });
        static_for<n>([&](auto ic) { constexpr int i = decltype(ic)::value; v[Q2C.idx[P][i]] = Ee * z[i]; });
    });
    static_for<27>([&](auto tc) {
        constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;
        const double zs = v[3 * g + c], za = v[3 * (18 + g) + c];
        v[3 * g + c] = zs - za; v[3 * (18 + g) + c] = zs + za;
    });
    static_for<27>([&](auto tc) {
        constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3, a = g / 3, cz = g % 3;
        const double zs = v[3 * (9 * a + cz) + c], za = v[3 * (9 * a + 6 + cz) + c];
        v[3 * (9 * a + cz) + c] = zs - za; v[3 * (9 * a + 6 + cz) + c] = zs + za;
    });
    static_for<27>([&](auto tc) {
        constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;
        const double zs = v[3 * (3 * g) + c], za = v[3 * (3 * g + 2) + c];
        v[3 * (3 * g) + c] = zs - za; v[3 * (3 * g + 2) + c] = zs + za;
    });
}

constexpr int Q2M_OB = 448;                                   // doubles per parked row segment (7 x 64)

typedef unsigned int q2u4_t __attribute__((ext_vector_type(4)));
typedef unsigned int q2u2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double q2_mkd(unsigned lo, unsigned hi) { return __longlong_as_double(((unsigned long long) hi << 32) | lo); }

// Round 4: rows reach and leave the lanes DIRECTLY.  Rounds 1-3 loaded a row segment in lane-contiguous pieces and turned it into the
// 9 values (3 nodes x 3 components) of every lane's element through LDS (7 writes + 9 reads and two wave barriers per row, twelve
// rows per step), and took finished rows back through LDS the same way.  A lane's nine values are 72 contiguous bytes of the row, the
// next lane's start 48 bytes on: buffer loads of 16 + 16 + 16 + 16 + 8 bytes per lane fetch them as they are (the node shared with the
// next element twice -- from the vector cache), buffer stores of 3 x 16 bytes write the lane's two finished nodes; an element outside
// the grid is an out-of-range offset (reads 0, stores nothing).  ~160 of ~250 LDS instructions per step and the 14 KB transpose buffer go.
template <int EXP>      // 0 production; timing ablations (wrong results): 1 no element product, 2 no row stores, 3 no row loads
__global__ void __launch_bounds__(256) k_apply_q2_march(DimsQ2 d, const double *__restrict__ tab, const double *__restrict__ E,
                                                        const double *__restrict__ u, double *__restrict__ out, int cb, int xsteps) {
    // sums a row already holds when this wave stores it, in the lanes' own layout (element t of lane l at [6 l + t], the last element's
    // final node at [6 l + 6 ..]): [wave][2 par + rx] for the rows ry = 0 (what the wave below left, or, for wave 0 of a block of the
    // second colour, what the block below wrote to `out`), [wave][4 + rx] for the rows ry = 2 of wave 3
    __shared__ double ldso[4][6 * Q2M_OB];
    __shared__ double ldsk[4][27 * 64];                       // `carry`: contributions to the node plane shared with the next step
    const int lane = threadIdx.x, wy = threadIdx.y;
    const int B = 2 * blockIdx.y + cb;                        // block of four consecutive element rows in y
    if (4 * B >= d.ny) return;                                // block-uniform
    const int ey = 4 * B + wy;
    const bool active = ey < d.ny;                            // idle waves only keep the barriers company
    const int eyc = active ? ey : d.ny - 1;
    double *carry = ldsk[wy] + lane;                          // element t of this lane at carry[64 t]
    const int xa = blockIdx.z * xsteps;
    const int xb = xa + xsteps < d.nx ? xa + xsteps : d.nx;
    double *obuf = ldso[wy];
    const int zb = Q2M_ZS * blockIdx.x - 1;                   // element of lane 0 (-1 in the first chunk: a dummy with zero modulus)
    const int ez = zb + lane;
    const bool elem_ok = ez >= 0 && ez < d.nz;
    const int ezc = ez < 0 ? 0 : (ez > d.nz - 1 ? d.nz - 1 : ez);
    const int rowbytes = 24 * d.NZ;                           // (launcher: below 2^31)
    const unsigned eoff = elem_ok ? 48u * (unsigned) ez : 0x7ffffff0u;       // this lane's nine doubles inside a node row
    // who else adds to this wave's shared rows: the waves of a block hand their ry = 2 rows to the wave above through LDS; across
    // blocks the second colour (cb = 1, launched after cb = 0) reads what its neighbours wrote to `out`
    const bool from_below = wy >= 1;                          // rows ry = 0: the wave below (always active when this one is)
    const bool rmw0 = wy == 0 && cb == 1;                     // rows ry = 0 of wave 0: block B - 1 ran in the first launch
    const bool hand_up = active && wy < 3 && ey + 1 < d.ny;   // rows ry = 2 go to the wave above instead of to memory
    const bool rmw2 = wy == 3 && cb == 1 && ey + 1 < d.ny;    // rows ry = 2 of wave 3: block B + 1 ran in the first launch

    double v[81], keep[27], pre[6][9], pre_o[2][9], Enext = 0.0;
    auto row_rsrc = [&](const double *base, int X, int Y) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(q2_row(base, d, X, Y)), 0, rowbytes, 0x00020000);
    };
    // the lane's nine values of a row: nodes 2 ez, 2 ez + 1, 2 ez + 2
    auto load9 = [&](double (&dst)[9], __amdgpu_buffer_rsrc_t r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const q2u4_t w = __builtin_amdgcn_raw_buffer_load_b128(r, eoff + 16 * q, 0, 0);
            dst[2 * q] = q2_mkd(w.x, w.y); dst[2 * q + 1] = q2_mkd(w.z, w.w);
        }
        const q2u2_t w = __builtin_amdgcn_raw_buffer_load_b64(r, eoff + 64, 0, 0);
        dst[8] = q2_mkd(w.x, w.y);
    };
    auto issue_loads = [&](int ex) {
        if (EXP == 3) { Enext = E[((long long) ex * d.ny + eyc) * d.nz + ezc]; return; }
        static_for<6>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            load9(pre[r], row_rsrc(u, 2 * ex + 1 + r / 3, 2 * eyc + r % 3));
        });
        Enext = E[((long long) ex * d.ny + eyc) * d.nz + ezc];
    };
    // hand a row to the wave above: bottom and middle node of every lane, and the final node plane from the last element
    auto deposit_row = [&](double *dst, const double *add9) {
#pragma unroll
        for (int q = 0; q < 6; ++q) dst[6 * lane + q] = add9[q];
        if (ez == d.nz - 1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[6 * lane + 6 + c] = add9[6 + c];
        }
    };
    // store one finished row: bottom and middle node of lanes 1..63 (lane 0 belongs to the chunk below), the last element of the
    // pencil also its top node (the final node plane, z = 2 nz); `add9` = the lane's 9 values of the row, `old` = parked sums
    auto store_row = [&](int X, int ry, const double *add9, bool has_old, const double *old) {
        const bool last = ez == d.nz - 1;
        double w9[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) w9[q] = add9[q];
        if (has_old) {
#pragma unroll
            for (int q = 0; q < 6; ++q) w9[q] += old[6 * lane + q];
            if (last) {
#pragma unroll
                for (int c = 0; c < 3; ++c) w9[6 + c] += old[6 * lane + 6 + c];
            }
        }
        const __amdgpu_buffer_rsrc_t r = row_rsrc(out, X, 2 * eyc + ry);
        const unsigned so = (lane >= 1 && elem_ok) ? eoff : 0x7ffffff0u;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const unsigned long long a = (unsigned long long) __double_as_longlong(w9[2 * q]), bq = (unsigned long long) __double_as_longlong(w9[2 * q + 1]);
            const q2u4_t w = {(unsigned) a, (unsigned) (a >> 32), (unsigned) bq, (unsigned) (bq >> 32)};
            __builtin_amdgcn_raw_buffer_store_b128(w, r, so + 16 * q, 0, 0);
        }
        if (last && lane >= 1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned long long a = (unsigned long long) __double_as_longlong(w9[6 + c]);
                const q2u2_t w = {(unsigned) a, (unsigned) (a >> 32)};
                __builtin_amdgcn_raw_buffer_store_b64(w, r, eoff + 48 + 8 * c, 0, 0);
            }
        }
    };

    const int e0 = xa > 0 ? xa - 1 : xa;
    if (active) {   // raw values of the first node plane
        static_for<3>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            double t9[9];
            load9(t9, row_rsrc(u, 2 * e0, 2 * ey + r));
#pragma unroll
            for (int q = 0; q < 9; ++q) keep[9 * r + q] = t9[q];
        });
#pragma unroll
        for (int t = 0; t < 27; ++t) carry[64 * t] = 0.0;
        issue_loads(e0);
    }

    int par = 0;
    for (int ex = e0; ex < xb; ++ex) {
        const bool store = ex >= xa;                          // block-uniform
        if (active) {
#pragma unroll
            for (int t = 0; t < 27; ++t) v[t] = keep[t];
            static_for<6>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
#pragma unroll
                for (int q = 0; q < 9; ++q) v[27 + 9 * r + q] = pre[r][q];
            });
            const double Ee = elem_ok ? Enext : 0.0;
            if (ex + 1 < xb) issue_loads(ex + 1);
#pragma unroll
            for (int t = 0; t < 27; ++t) keep[t] = v[54 + t];
            if (store && (rmw0 || rmw2)) {                    // sums of the first colour: requested now, parked in LDS after the arithmetic
                static_for<2>([&](auto rc) {
                    constexpr int rx = decltype(rc)::value;
                    load9(pre_o[rx], row_rsrc(out, 2 * ex + rx, 2 * ey + (rmw0 ? 0 : 2)));
                });
            }

            if (EXP != 1) q2_element_product<2>(v, tab, Ee);
            else { v[0] *= Ee; }

            // node plane shared with the next element in z: one lane up (lane 0 is the element of the chunk below, its rows are not stored)
            static_for<27>([&](auto tc) {
                constexpr int t = decltype(tc)::value, g = t / 3, c = t % 3;
                v[3 * (3 * g) + c] += lane_below(v[3 * (3 * g + 2) + c]);
            });
            // node plane shared with the previous element in x: what that step left; this step's last plane waits for the next one
#pragma unroll
            for (int t = 0; t < 27; ++t) { v[t] += carry[64 * t]; carry[64 * t] = v[54 + t]; }
            if (store) {
                if (rmw0 || rmw2) {
                    static_for<2>([&](auto rc) {
                        constexpr int rx = decltype(rc)::value;
                        double *dst = obuf + (rmw0 ? 2 * par + rx : 4 + rx) * Q2M_OB;
#pragma unroll
                        for (int q = 0; q < 6; ++q) dst[6 * lane + q] = pre_o[rx][q];
                        if (ez == d.nz - 1) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) dst[6 * lane + 6 + c] = pre_o[rx][6 + c];
                        }
                    });
                }
                if (hand_up) {
                    deposit_row(ldso[wy + 1] + (2 * par + 0) * Q2M_OB, &v[9 * 2]);
                    deposit_row(ldso[wy + 1] + (2 * par + 1) * Q2M_OB, &v[9 * 5]);
                }
            }
        }
        if (store) {
            __syncthreads();                                  // the rows handed up are in place (double-buffered by step parity)
            if (active && EXP == 2) { if (v[5] == 1.2345) out[0] = v[7]; }
            if (active && EXP != 2) {
                static_for<6>([&](auto rc) {
                    constexpr int g = decltype(rc)::value, rx = g / 3, ry = g % 3;
                    if (ry == 0) store_row(2 * ex + rx, 0, &v[9 * g], from_below || rmw0, obuf + (2 * par + rx) * Q2M_OB);
                    else if (ry == 1) store_row(2 * ex + rx, 1, &v[9 * g], false, obuf);
                    else if (!hand_up) store_row(2 * ex + rx, 2, &v[9 * g], rmw2, obuf + (4 + rx) * Q2M_OB);
                });
            }
            par ^= 1;
        }
    }
    if (xb == d.nx) {                                         // the domain face: the last node plane is complete as it is
        double last[27];
        if (active) {
#pragma unroll
            for (int t = 0; t < 27; ++t) last[t] = carry[64 * t];
            if (rmw0 || rmw2) {
                double t9[9];
                load9(t9, row_rsrc(out, 2 * d.nx, 2 * ey + (rmw0 ? 0 : 2)));
                double *dst = obuf + (rmw0 ? 2 * par : 4) * Q2M_OB;
#pragma unroll
                for (int q = 0; q < 6; ++q) dst[6 * lane + q] = t9[q];
                if (ez == d.nz - 1) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) dst[6 * lane + 6 + c] = t9[6 + c];
                }
            }
            if (hand_up) deposit_row(ldso[wy + 1] + (2 * par) * Q2M_OB, &last[18]);
        }
        __syncthreads();
        if (active) {
            store_row(2 * d.nx, 0, &last[0], from_below || rmw0, obuf + (2 * par) * Q2M_OB);
            store_row(2 * d.nx, 1, &last[9], false, obuf);
            if (!hand_up) store_row(2 * d.nx, 2, &last[18], rmw2, obuf + 4 * Q2M_OB);
        }
    }
}

void launch_apply_q2_march(int nx, int ny, int nz, const double *tab, const double *E, const double *u, double *out, hipStream_t s) {
    DimsQ2 d{nx, ny, nz, 2 * nx + 1, 2 * ny + 1, 2 * nz + 1};
    const int nchunk = (nz + Q2M_ZS - 1) / Q2M_ZS;
    int nxc = nx >= 256 ? 8 : (nx >= 64 ? 4 : (nx >= 16 ? 2 : 1));
    const int xsteps = (nx + nxc - 1) / nxc;
    nxc = (nx + xsteps - 1) / xsteps;
    const int nblk = (ny + 3) / 4;                            // blocks of four element rows; even blocks first, then the odd ones
    for (int cb = 0; cb < 2; ++cb) {
        const int cnt = (nblk - cb + 1) / 2;
        if (cnt <= 0) continue;
        const dim3 grd(nchunk, cnt, nxc), blk(64, 4, 1);
#ifdef VFEM_ABLATION
        switch (vfem::ablate_apply()) {
            case 1: k_apply_q2_march<1><<<grd, blk, 0, s>>>(d, tab, E, u, out, cb, xsteps); break;
            case 2: k_apply_q2_march<2><<<grd, blk, 0, s>>>(d, tab, E, u, out, cb, xsteps); break;
            case 3: k_apply_q2_march<3><<<grd, blk, 0, s>>>(d, tab, E, u, out, cb, xsteps); break;
            default: k_apply_q2_march<0><<<grd, blk, 0, s>>>(d, tab, E, u, out, cb, xsteps);
        }
#else
        k_apply_q2_march<0><<<grd, blk, 0, s>>>(d, tab, E, u, out, cb, xsteps);
#endif
    }
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------
// One colour of the 27-colour block Gauss-Seidel sweep on the finest degree-2 level (MG.hh:193-340 with Degrees = 2,2,2):
// one thread per node of the colour (colour = local node index; per axis nodes start at the local index and advance by one
// element for the mid node, by two for a boundary node).  All nodes of a colour have the same local index in each of their
// incident elements, so the rows of K0 are wave-uniform.
// ------------------------------------------------------------------------------------------------------
struct Q2Color { int start[3], inc[3], cnt[3]; };

template <int PX, int PY, int PZ>      // node parities of the colour (odd = mid node of one element) at compile time: the slot loops unroll
__global__ void __launch_bounds__(256) k_gs_q2_level0(DimsQ2 d, Q2Color col, const double *__restrict__ K0, const double *__restrict__ E,
                                                      double *__restrict__ u, const double *__restrict__ b,
                                                      const uint8_t *__restrict__ mask, int forward) {
    const int c = blockIdx.x * 64 + threadIdx.x, bq = blockIdx.y * 4 + threadIdx.y, a = blockIdx.z;
    if (c >= col.cnt[2] || bq >= col.cnt[1] || a >= col.cnt[0]) return;
    const int i = col.start[0] + a * col.inc[0], j = col.start[1] + bq * col.inc[1], k = col.start[2] + c * col.inc[2];
    constexpr int px = PX, py = PY, pz = PZ;
    constexpr int nex = px ? 1 : 2, ney = py ? 1 : 2, nez = pz ? 1 : 2;
    double S[3] = {0.0, 0.0, 0.0}, M[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) M[q] = 0.0;
#pragma unroll
    for (int sx = 0; sx < nex; ++sx) {
        const int ex = px ? i / 2 : i / 2 - 1 + sx, lx = px ? 1 : (sx ? 0 : 2);
        if (ex < 0 || ex >= d.nx) continue;
#pragma unroll
        for (int sy = 0; sy < ney; ++sy) {
            const int ey = py ? j / 2 : j / 2 - 1 + sy, ly = py ? 1 : (sy ? 0 : 2);
            if (ey < 0 || ey >= d.ny) continue;
#pragma unroll
            for (int sz = 0; sz < nez; ++sz) {
                const int ez = pz ? k / 2 : k / 2 - 1 + sz, lz = pz ? 1 : (sz ? 0 : 2);
                if (ez < 0 || ez >= d.nz) continue;
                const int ln = 9 * lx + 3 * ly + lz;
                const double Ee = E[((long long) ex * d.ny + ey) * d.nz + ez];
                const double *r0 = K0 + (3 * ln) * 81, *r1 = r0 + 81, *r2 = r1 + 81;
                double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
                for (int ma = 0; ma < 3; ++ma)
#pragma unroll
                    for (int mb = 0; mb < 3; ++mb) {
                        const long long rowbase = ((long long) (2 * ex + ma) * d.NY + (2 * ey + mb)) * d.NZ + 2 * ez;
#pragma unroll
                        for (int mc = 0; mc < 3; ++mc) {
                            const int m = 9 * ma + 3 * mb + mc;
                            const double *um = u + 3 * (rowbase + mc);
                            const double u0 = um[0], u1 = um[1], u2 = um[2];
                            t0 = fma(r0[3 * m], u0, fma(r0[3 * m + 1], u1, fma(r0[3 * m + 2], u2, t0)));
                            t1 = fma(r1[3 * m], u0, fma(r1[3 * m + 1], u1, fma(r1[3 * m + 2], u2, t1)));
                            t2 = fma(r2[3 * m], u0, fma(r2[3 * m + 1], u1, fma(r2[3 * m + 2], u2, t2)));
                        }
                    }
                S[0] = fma(Ee, t0, S[0]); S[1] = fma(Ee, t1, S[1]); S[2] = fma(Ee, t2, S[2]);
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    M[cc] = fma(Ee, r0[3 * ln + cc], M[cc]);
                    M[3 + cc] = fma(Ee, r1[3 * ln + cc], M[3 + cc]);
                    M[6 + cc] = fma(Ee, r2[3 * ln + cc], M[6 + cc]);
                }
            }
        }
    }
    const long long n = ((long long) i * d.NY + j) * d.NZ + k;
    double bms[3], ud[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) bms[cc] = b[3 * n + cc] - S[cc];
    gs_solve(bms, M, mask ? mask[n] : (uint8_t) 0, forward != 0, ud);
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) u[3 * n + cc] += ud[cc];
}

void launch_gs_sweep_q2_level0(int nx, int ny, int nz, const double *K0, const double *E, double *u, const double *b,
                               const uint8_t *mask, int forward, hipStream_t s, int first, int count) {
    DimsQ2 d{nx, ny, nz, 2 * nx + 1, 2 * ny + 1, 2 * nz + 1};
    const int NN[3] = {d.NX, d.NY, d.NZ};
    for (int ci = first; ci < (first + count < 27 ? first + count : 27); ++ci) {
        const int lni = forward ? ci : 26 - ci;                      // MG.hh:293-295
        const int l[3] = {lni / 9, (lni / 3) % 3, lni % 3};
        Q2Color col;
        bool empty = false;
        for (int a = 0; a < 3; ++a) {
            col.start[a] = l[a];
            col.inc[a] = (l[a] == 1) ? 2 : 4;                        // MG.hh:301-305: (1 + isBoundary) * degree
            col.cnt[a] = l[a] > NN[a] - 1 ? 0 : (NN[a] - 1 - l[a]) / col.inc[a] + 1;
            empty = empty || col.cnt[a] == 0;
        }
        if (empty) continue;
        const dim3 grd((col.cnt[2] + 63) / 64, (col.cnt[1] + 3) / 4, col.cnt[0]), blk(64, 4, 1);
#define VFEM_Q2GS(X, Y, Z) k_gs_q2_level0<X, Y, Z><<<grd, blk, 0, s>>>(d, col, K0, E, u, b, mask, forward)
        switch (4 * (l[0] & 1) + 2 * (l[1] & 1) + (l[2] & 1)) {
            case 0: VFEM_Q2GS(0, 0, 0); break;
            case 1: VFEM_Q2GS(0, 0, 1); break;
            case 2: VFEM_Q2GS(0, 1, 0); break;
            case 3: VFEM_Q2GS(0, 1, 1); break;
            case 4: VFEM_Q2GS(1, 0, 0); break;
            case 5: VFEM_Q2GS(1, 0, 1); break;
            case 6: VFEM_Q2GS(1, 1, 0); break;
            default: VFEM_Q2GS(1, 1, 1);
        }
#undef VFEM_Q2GS
    }
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------
// The same colour sweep with the sum ordered by NEIGHBOUR NODE instead of by element.  k_gs_q2_level0 reads the 27 nodes of each
// of a node's incident elements (8 x 81 = 648 values for a vertex node, at a lane stride of 96 bytes) although the elements
// overlap in only 5^3 = 125 distinct nodes: the sweep is bound by the load instructions a CU's address unit processes, not by
// its 1944 multiply-adds (4 % of the fp64 peak).  Here a lane walks the distinct neighbour nodes, reads each one's three values
// once, and adds K0[rows of the node in e][columns of the neighbour in e] u to a partial sum per incident element e containing
// both (8 x 3 partial sums, scaled by the elements' moduli at the end -- the structure of k_q2_level1 with the incident
// elements in the role of the children): 375 values instead of 648 for a vertex node, the same 1944 multiply-adds.  The
// coefficients are wave-uniform and arrive by scalar loads in the order of use, groups of nine requested three groups ahead.
// ------------------------------------------------------------------------------------------------------
struct Q2Grp { int ox, oy, oz, e, ln, lm, first, last, nb; };     // neighbour offset, element slot sx*4+sy*2+sz, local indices, flags, neighbour ordinal
__host__ __device__ constexpr int q2a_noff(int P) { return P ? 3 : 5; }
__host__ __device__ constexpr int q2a_off(int P, int idx) { return P ? idx - 1 : idx - 2; }
__host__ __device__ constexpr int q2a_nopt(int P, int o) { return (P == 0 && o == 0) ? 2 : 1; }
__host__ __device__ constexpr int q2a_sel(int P, int o, int t) { return P ? 0 : (o < 0 ? 0 : (o > 0 ? 1 : t)); }          // lower / upper element
__host__ __device__ constexpr int q2a_ln(int P, int o, int t) { return P ? 1 : (q2a_sel(P, o, t) ? 0 : 2); }
__host__ __device__ constexpr int q2a_lm(int P, int o, int t) { return P ? 1 + o : (q2a_sel(P, o, t) ? o : 2 + o); }
__host__ __device__ constexpr int q2_ngroups(int PX, int PY, int PZ) { return (PX ? 3 : 6) * (PY ? 3 : 6) * (PZ ? 3 : 6); }
__host__ __device__ constexpr Q2Grp q2_group(int PX, int PY, int PZ, int g) {
    int n = 0, nb = 0;
    for (int ix = 0; ix < q2a_noff(PX); ++ix)
        for (int iy = 0; iy < q2a_noff(PY); ++iy)
            for (int iz = 0; iz < q2a_noff(PZ); ++iz, ++nb) {
                const int ox = q2a_off(PX, ix), oy = q2a_off(PY, iy), oz = q2a_off(PZ, iz);
                const int cnt = q2a_nopt(PX, ox) * q2a_nopt(PY, oy) * q2a_nopt(PZ, oz);
                if (g < n + cnt) {
                    const int w = g - n, tz = w % q2a_nopt(PZ, oz), ty = (w / q2a_nopt(PZ, oz)) % q2a_nopt(PY, oy),
                              tx = w / (q2a_nopt(PZ, oz) * q2a_nopt(PY, oy));
                    return Q2Grp{ox, oy, oz, 4 * q2a_sel(PX, ox, tx) + 2 * q2a_sel(PY, oy, ty) + q2a_sel(PZ, oz, tz),
                                 9 * q2a_ln(PX, ox, tx) + 3 * q2a_ln(PY, oy, ty) + q2a_ln(PZ, oz, tz),
                                 9 * q2a_lm(PX, ox, tx) + 3 * q2a_lm(PY, oy, ty) + q2a_lm(PZ, oz, tz), w == 0, w == cnt - 1, nb};
                }
                n += cnt;
            }
    return Q2Grp{0, 0, 0, 0, 0, 0, 0, 0, 0};
}
__host__ __device__ constexpr int q2_center_group(int PX, int PY, int PZ) {      // first group of the neighbour (0, 0, 0)
    for (int g = 0; g < q2_ngroups(PX, PY, PZ); ++g) {
        const Q2Grp q = q2_group(PX, PY, PZ, g);
        if (q.ox == 0 && q.oy == 0 && q.oz == 0) return g;
    }
    return 0;
}
// coefficient table of one parity class: [group][3 r][3 c] = K0[(3 ln + r) * 81 + 3 lm + c]; classes follow each other in the
// order 4 PX + 2 PY + PZ (offsets: q2_table_offset)
__host__ __device__ constexpr int q2_table_offset(int cls) {
    int o = 0;
    for (int c = 0; c < cls; ++c) o += 9 * q2_ngroups((c >> 2) & 1, (c >> 1) & 1, c & 1);
    return o;
}
void build_q2_gs_table(const double *K0, std::vector<double> &tab) {
    tab.assign((size_t) q2_table_offset(8), 0.0);
    for (int cls = 0; cls < 8; ++cls) {
        const int PX = (cls >> 2) & 1, PY = (cls >> 1) & 1, PZ = cls & 1;
        for (int g = 0; g < q2_ngroups(PX, PY, PZ); ++g) {
            const Q2Grp q = q2_group(PX, PY, PZ, g);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    tab[(size_t) q2_table_offset(cls) + 9 * g + 3 * r + c] = K0[(size_t) (3 * q.ln + r) * 81 + 3 * q.lm + c];
        }
    }
}

#ifndef VFEM_Q2N_AHEAD
#define VFEM_Q2N_AHEAD 3
#endif
#ifndef VFEM_Q2ROWS_WAVES
#define VFEM_Q2ROWS_WAVES 4            // waves (z-rows of nodes) per block of k_gs_q2_level0_rows: LDS per block = waves x 13.4 KB
#endif
constexpr int Q2N_AHEAD = VFEM_Q2N_AHEAD;
template <int PX, int PY, int PZ>
__global__ void __launch_bounds__(256) k_gs_q2_level0_nodes(DimsQ2 d, Q2Color col, const double *__restrict__ tabc, const double *__restrict__ E,
                                                            double *__restrict__ u, const double *__restrict__ b,
                                                            const uint8_t *__restrict__ mask, int forward) {
    // lanes packed over the launch's nodes of an x-plane, row after row
    const int fq = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x, a = blockIdx.z;
    if (fq >= col.cnt[1] * col.cnt[2]) return;
    const int bq = fq / col.cnt[2], c = fq - bq * col.cnt[2];
    const int i = col.start[0] + a * col.inc[0], j = col.start[1] + bq * col.inc[1], k = col.start[2] + c * col.inc[2];
    constexpr int NG = q2_ngroups(PX, PY, PZ);
    // moduli of the incident elements (slot sx*4 + sy*2 + sz; a mid node has one element along that axis: slot bit 0)
    double Ee[8];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
        const int sx = (sl >> 2) & 1, sy = (sl >> 1) & 1, sz = sl & 1;
        const int ex = PX ? i / 2 : i / 2 - 1 + sx, ey = PY ? j / 2 : j / 2 - 1 + sy, ez = PZ ? k / 2 : k / 2 - 1 + sz;
        const bool used = (!PX || sx == 0) && (!PY || sy == 0) && (!PZ || sz == 0);
        const bool ok = used && ex >= 0 && ex < d.nx && ey >= 0 && ey < d.ny && ez >= 0 && ez < d.nz;
        Ee[sl] = ok ? E[((long long) ex * d.ny + ey) * d.nz + ez] : 0.0;
    }
    // t[e][r]: rows of the node in element e against the neighbours visited so far (unscaled); S = sum_e E_e t[e]
    double t[8][3], cf[Q2N_AHEAD + 1][9], uv[Q2N_AHEAD + 1][3];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e][0] = t[e][1] = t[e][2] = 0.0;
    static_for<NG + Q2N_AHEAD>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < NG) {                                              // request group g ...
#pragma unroll
            for (int q = 0; q < 9; ++q) cf[g % (Q2N_AHEAD + 1)][q] = tabc[9 * g + q];
            constexpr Q2Grp G = q2_group(PX, PY, PZ, g);
            if constexpr (G.first) {                                         // ... and the values of its neighbour node
                // a neighbour outside the grid belongs to no existing element (moduli 0): any finite value will do -- the
                // address is clamped into the grid, no divergent control flow
                const int ni = min(max(i + G.ox, 0), d.NX - 1), nj = min(max(j + G.oy, 0), d.NY - 1), nk = min(max(k + G.oz, 0), d.NZ - 1);
                const double *um = u + 3 * (((long long) ni * d.NY + nj) * d.NZ + nk);
                uv[G.nb % (Q2N_AHEAD + 1)][0] = um[0]; uv[G.nb % (Q2N_AHEAD + 1)][1] = um[1]; uv[G.nb % (Q2N_AHEAD + 1)][2] = um[2];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g >= Q2N_AHEAD) {                                      // use group g - AHEAD
            constexpr int gg = g - Q2N_AHEAD;
            constexpr Q2Grp G = q2_group(PX, PY, PZ, gg);
#pragma unroll
            for (int r = 0; r < 3; ++r)
                t[G.e][r] = fma(cf[gg % (Q2N_AHEAD + 1)][3 * r], uv[G.nb % (Q2N_AHEAD + 1)][0],
                                fma(cf[gg % (Q2N_AHEAD + 1)][3 * r + 1], uv[G.nb % (Q2N_AHEAD + 1)][1],
                                    fma(cf[gg % (Q2N_AHEAD + 1)][3 * r + 2], uv[G.nb % (Q2N_AHEAD + 1)][2], t[G.e][r])));
            // (arithmetic carries no ordering of its own: instruction selection may sink it below every barrier, which keeps all
            // coefficients alive; the empty statement ties the group's multiply-adds to this place)
            asm volatile("" : "+v"(t[G.e][0]), "+v"(t[G.e][1]), "+v"(t[G.e][2]));
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    // the groups of the node with itself (offset 0, 0, 0): diagonal block M = sum_e E_e K0[ln_e][ln_e]
    constexpr int G0 = q2_center_group(PX, PY, PZ);
    double S[3] = {0.0, 0.0, 0.0}, M[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) M[q] = 0.0;
    static_for<q2a_nopt(PX, 0) * q2a_nopt(PY, 0) * q2a_nopt(PZ, 0)>([&](auto wc) {
        constexpr Q2Grp G = q2_group(PX, PY, PZ, G0 + decltype(wc)::value);
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = fma(Ee[G.e], tabc[9 * (G0 + decltype(wc)::value) + q], M[q]);
    });
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int r = 0; r < 3; ++r) S[r] = fma(Ee[e], t[e][r], S[r]);
    const long long n = ((long long) i * d.NY + j) * d.NZ + k;
    double bms[3], ud[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) bms[cc] = b[3 * n + cc] - S[cc];
    gs_solve(bms, M, mask ? mask[n] : (uint8_t) 0, forward != 0, ud);
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) u[3 * n + cc] += ud[cc];
}

// ------------------------------------------------------------------------------------------------------
// The neighbour-node sweep with the neighbour ROWS staged through LDS.  A wave relaxes 64 nodes of one colour in a z-row; they
// are 2 or 4 nodes apart, so every load of k_gs_q2_level0_nodes fetches 8 of 128 bytes per cache line it touches and a vertex
// node costs 263 such instructions.  Here the wave copies each of its (up to 25) neighbour rows once, as one contiguous run
// (the z-extent of its 64 nodes plus the stencil reach: 771 doubles at most, 13 coalesced 8-byte loads per lane), into its
// own LDS buffer and the lanes pick their three to five neighbours of that row from there; the next row's loads are in flight
// while the current row is used.  The image is padded by one double per lane stride (13 or 7 doubles between lanes: two-way
// bank conflicts instead of sixteen-way).  No block-level synchronisation: a wave only reads what it wrote itself, and the
// LDS queue of a wave is in order.
// ------------------------------------------------------------------------------------------------------
template <int PX, int PY, int PZ>
__global__ void __launch_bounds__(256) k_gs_q2_level0_rows(DimsQ2 d, Q2Color col, const double *__restrict__ tabc, const double *__restrict__ E,
                                                           double *__restrict__ u, const double *__restrict__ b,
                                                           const uint8_t *__restrict__ mask, int forward) {
    constexpr int SZ = PZ ? 2 : 4, R = PZ ? 1 : 2, NOZ = 2 * R + 1;           // lane stride in nodes, stencil reach, neighbours per row
    constexpr int SPAN3 = 3 * (63 * SZ + NOZ), NLD = (SPAN3 + 63) / 64;          // doubles of a row segment, loads per lane
    constexpr int IMG = SPAN3 + SPAN3 / (3 * SZ) + 2;                            // padded image: one double of padding per 3 SZ doubles
    constexpr int NROWS = q2a_noff(PX) * q2a_noff(PY), NG = q2_ngroups(PX, PY, PZ);
    __shared__ double img[VFEM_Q2ROWS_WAVES][2][IMG];
    const int lane = threadIdx.x, w = threadIdx.y;
    const int bq = blockIdx.y * VFEM_Q2ROWS_WAVES + w, a = blockIdx.z;
    if (bq >= col.cnt[1] || a >= col.cnt[0]) return;                             // (whole wave)
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < col.cnt[2];
    const int i = col.start[0] + a * col.inc[0], j = col.start[1] + bq * col.inc[1];
    const int k = col.start[2] + (live ? c : col.cnt[2] - 1) * col.inc[2];
    const int kseg = col.start[2] + blockIdx.x * 64 * col.inc[2] - R;          // first node of the staged segment (may be < 0: clamped loads)
    double Ee[8];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
        const int sx = (sl >> 2) & 1, sy = (sl >> 1) & 1, sz = sl & 1;
        const int ex = PX ? i / 2 : i / 2 - 1 + sx, ey = PY ? j / 2 : j / 2 - 1 + sy, ez = PZ ? k / 2 : k / 2 - 1 + sz;
        const bool used = (!PX || sx == 0) && (!PY || sy == 0) && (!PZ || sz == 0);
        const bool ok = used && ex >= 0 && ex < d.nx && ey >= 0 && ey < d.ny && ez >= 0 && ez < d.nz;
        Ee[sl] = ok ? E[((long long) ex * d.ny + ey) * d.nz + ez] : 0.0;
    }
    double stage[NLD];
    // row rho = (ix, iy) of the neighbour enumeration; rows outside the grid are read at the clamped row (their elements have
    // modulus 0, any finite value will do)
    auto load_row = [&](int rho) {
        const int ox = q2a_off(PX, rho / q2a_noff(PY)), oy = q2a_off(PY, rho % q2a_noff(PY));
        const int ni = min(max(i + ox, 0), d.NX - 1), nj = min(max(j + oy, 0), d.NY - 1);
        const double *row = u + 3 * (((long long) ni * d.NY + nj) * d.NZ);
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int q = m * 64 + lane;
            const int gz = min(max(3 * kseg + q, 0), 3 * d.NZ - 1);
            stage[m] = row[gz];
        }
    };
    auto store_row = [&](int buf) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int q = m * 64 + lane;
            if (q < SPAN3) img[w][buf][q + q / (3 * SZ)] = stage[m];
        }
    };
    load_row(0);
    store_row(0);
    double t[8][3], cf[Q2N_AHEAD + 1][9], uv[Q2N_AHEAD + 1][3];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e][0] = t[e][1] = t[e][2] = 0.0;
    static_for<NG + Q2N_AHEAD>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < NG) {                                              // request group g ...
#pragma unroll
            for (int q = 0; q < 9; ++q) cf[g % (Q2N_AHEAD + 1)][q] = tabc[9 * g + q];
            constexpr Q2Grp G = q2_group(PX, PY, PZ, g);
            constexpr int rho = G.nb / NOZ, tz = G.nb % NOZ;
            if constexpr (G.first && tz == 0 && rho + 1 < NROWS) load_row(rho + 1);   // next row: in flight while this one is used
            if constexpr (G.first) {                                         // ... and the values of its neighbour node, from the image
                const double *src = &img[w][rho & 1][3 * (lane * SZ + tz) + lane + tz / SZ];
                uv[G.nb % (Q2N_AHEAD + 1)][0] = src[0]; uv[G.nb % (Q2N_AHEAD + 1)][1] = src[1]; uv[G.nb % (Q2N_AHEAD + 1)][2] = src[2];
            }
            // the last neighbour of the row has been read: the other buffer (row rho - 1, fully consumed) takes the next row
            if constexpr (G.last && tz == NOZ - 1 && rho + 1 < NROWS) store_row((rho + 1) & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g >= Q2N_AHEAD) {                                      // use group g - AHEAD
            constexpr int gg = g - Q2N_AHEAD;
            constexpr Q2Grp G = q2_group(PX, PY, PZ, gg);
#pragma unroll
            for (int r = 0; r < 3; ++r)
                t[G.e][r] = fma(cf[gg % (Q2N_AHEAD + 1)][3 * r], uv[G.nb % (Q2N_AHEAD + 1)][0],
                                fma(cf[gg % (Q2N_AHEAD + 1)][3 * r + 1], uv[G.nb % (Q2N_AHEAD + 1)][1],
                                    fma(cf[gg % (Q2N_AHEAD + 1)][3 * r + 2], uv[G.nb % (Q2N_AHEAD + 1)][2], t[G.e][r])));
            asm volatile("" : "+v"(t[G.e][0]), "+v"(t[G.e][1]), "+v"(t[G.e][2]));
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    constexpr int G0 = q2_center_group(PX, PY, PZ);
    double S[3] = {0.0, 0.0, 0.0}, M[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) M[q] = 0.0;
    static_for<q2a_nopt(PX, 0) * q2a_nopt(PY, 0) * q2a_nopt(PZ, 0)>([&](auto wc) {
        constexpr Q2Grp G = q2_group(PX, PY, PZ, G0 + decltype(wc)::value);
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = fma(Ee[G.e], tabc[9 * (G0 + decltype(wc)::value) + q], M[q]);
    });
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int r = 0; r < 3; ++r) S[r] = fma(Ee[e], t[e][r], S[r]);
    if (!live) return;
    const long long n = ((long long) i * d.NY + j) * d.NZ + k;
    double bms[3], ud[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) bms[cc] = b[3 * n + cc] - S[cc];
    gs_solve(bms, M, mask ? mask[n] : (uint8_t) 0, forward != 0, ud);
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) u[3 * n + cc] += ud[cc];
}

void launch_gs_sweep_q2_level0_nodes(int nx, int ny, int nz, const double *tab, const double *E, double *u, const double *b,
                                     const uint8_t *mask, int forward, hipStream_t s, int first, int count, int rows_in_lds) {
    DimsQ2 d{nx, ny, nz, 2 * nx + 1, 2 * ny + 1, 2 * nz + 1};
    const int NN[3] = {d.NX, d.NY, d.NZ};
    for (int ci = first; ci < (first + count < 27 ? first + count : 27); ++ci) {
        const int lni = forward ? ci : 26 - ci;                      // MG.hh:293-295
        const int l[3] = {lni / 9, (lni / 3) % 3, lni % 3};
        Q2Color col;
        bool empty = false;
        for (int a = 0; a < 3; ++a) {
            col.start[a] = l[a];
            col.inc[a] = (l[a] == 1) ? 2 : 4;                        // MG.hh:301-305
            col.cnt[a] = l[a] > NN[a] - 1 ? 0 : (NN[a] - 1 - l[a]) / col.inc[a] + 1;
            empty = empty || col.cnt[a] == 0;
        }
        if (empty) continue;
        const int cls = 4 * (l[0] & 1) + 2 * (l[1] & 1) + (l[2] & 1);
        const double *tc = tab + q2_table_offset(cls);
        // rows kernel: whole waves of 64 nodes per z-row; a row of 2^k + 1 nodes leaves one node over, and a wave for it would
        // run the full stream with one lane: the left-over columns (up to 16 per row) go to the gather kernel, lanes packed
        // over the rows (the nodes of a colour are independent: any order)
        Q2Color colr = col, coll = col;
        const int over = col.cnt[2] % 64;
        const bool split = rows_in_lds && col.cnt[2] > 64 && over >= 1 && over <= 16;
        if (split) {
            colr.cnt[2] = col.cnt[2] - over;
            coll.start[2] = col.start[2] + colr.cnt[2] * col.inc[2];
            coll.cnt[2] = over;
        }
        const Q2Color &coln = split ? coll : col;
        const dim3 grd((coln.cnt[1] * coln.cnt[2] + 255) / 256, 1, coln.cnt[0]), blk(64, 4, 1);
        const dim3 grdr((colr.cnt[2] + 63) / 64, (colr.cnt[1] + VFEM_Q2ROWS_WAVES - 1) / VFEM_Q2ROWS_WAVES, colr.cnt[0]), blkr(64, VFEM_Q2ROWS_WAVES, 1);
#define VFEM_Q2GSN(X, Y, Z) do { if (rows_in_lds) k_gs_q2_level0_rows<X, Y, Z><<<grdr, blkr, 0, s>>>(d, colr, tc, E, u, b, mask, forward); \
                                if (!rows_in_lds || split) k_gs_q2_level0_nodes<X, Y, Z><<<grd, blk, 0, s>>>(d, coln, tc, E, u, b, mask, forward); } while (0)
        switch (cls) {
            case 0: VFEM_Q2GSN(0, 0, 0); break;
            case 1: VFEM_Q2GSN(0, 0, 1); break;
            case 2: VFEM_Q2GSN(0, 1, 0); break;
            case 3: VFEM_Q2GSN(0, 1, 1); break;
            case 4: VFEM_Q2GSN(1, 0, 0); break;
            case 5: VFEM_Q2GSN(1, 0, 1); break;
            case 6: VFEM_Q2GSN(1, 1, 0); break;
            default: VFEM_Q2GSN(1, 1, 1);
        }
#undef VFEM_Q2GSN
    }
    VFEM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------
// Level 1 of the degree-2 hierarchy without stored element matrices.  The Galerkin matrix of a level-1 element is
// Ke = sum_f E_f cK0[f] over its eight children (buildPESCoarse, MG.hh:604-669, with cK0[f] = I_f^T K0 I_f constant); stored it
// is 81 x 81 doubles = 52 KB per element -- 110 GB at 256^3 fine elements, read once per sweep.  Here a thread owns a node as
// on level 0 and evaluates   S = sum_e sum_f E_f (cK0[f] u_e)[rows of the node]   directly: the rows of cK0[f] are the same
// for every node of a launch (one colour, or one parity class of the apply), so they arrive by scalar loads; per incident
// element a lane reads the element's 27 node values once and keeps 8 x 3 partial sums, one per child.  1944 multiply-adds
// per incident element instead of 243 loaded matrix entries.
// MODE 0: relax the nodes of the colour (MG.hh:254-264); 1: out = K u; 2: out = zeroDirichlet(b - K u); 3: out = zeroDirichlet(K u)
// ------------------------------------------------------------------------------------------------------
struct Q2L1 {
    DimsQ2 d;                 // the level-1 grid
    Q2Color col;              // nodes visited: start / stride / count per axis
    const double *tab;        // [27 ln][27 m][8 f][3 r][3 c] = cK0[f][(3 ln + r) * 81 + 3 m + c]; child index bit a = upper half along axis a (bit 0: x)
    const double *Ef;         // moduli of the level-0 elements, stored array (x slowest)
    int fny, fnz;             // level-0 elements per y / z
    int fx0;                  // stored level-0 layer of the first child of local level-1 layer 0
};

// The 1944 coefficients of one incident element are consumed in 216 groups of nine (node m of the element, child f), each read
// once from the table by scalar loads.  Left to itself the compiler requests all of them up front and parks the overflow in
// VGPR lanes (1.9 k v_writelane + 1.9 k v_readlane per element, more issue slots than the multiply-adds).  The groups are
// therefore requested Q2L1_AHEAD groups before their use and scheduling barriers keep requests and uses in that order: at most
// (Q2L1_AHEAD + 1) x 18 SGPRs of coefficients are live.
#ifndef VFEM_Q2L1_AHEAD
#define VFEM_Q2L1_AHEAD 3
#endif
constexpr int Q2L1_AHEAD = VFEM_Q2L1_AHEAD;

// contribution of one incident element (ex, ey, ez), in which the node has local index ln (uniform over the wave), to S and M
__device__ __forceinline__ void q2l1_element(const Q2L1 &a, const double *__restrict__ u, int ex, int ey, int ez, int ln, double S[3], double M[9]) {
    const DimsQ2 &d = a.d;
    double t[8][3];
#pragma unroll
    for (int f = 0; f < 8; ++f) t[f][0] = t[f][1] = t[f][2] = 0.0;
    const double *tab = a.tab + (size_t) ln * (27 * 72);
    const long long nbase = ((long long) (2 * ex) * d.NY + 2 * ey) * d.NZ + 2 * ez;
    double cf[Q2L1_AHEAD + 1][9], uv[2][3];
    static_for<216 + Q2L1_AHEAD>([&](auto gc) {                             // compile-time group index: every array index below folds
        constexpr int g = decltype(gc)::value;
        if constexpr (g < 216) {                                             // request group g
#pragma unroll
            for (int q = 0; q < 9; ++q) cf[g % (Q2L1_AHEAD + 1)][q] = tab[9 * g + q];
            if constexpr (g % 8 == 0) {                                      // ... and the values of its node
                constexpr int m = g / 8;
                const double *um = u + 3 * (nbase + ((long long) (m / 9) * d.NY + (m / 3) % 3) * d.NZ + m % 3);
                uv[m & 1][0] = um[0]; uv[m & 1][1] = um[1]; uv[m & 1][2] = um[2];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g >= Q2L1_AHEAD) {                                     // use group g - AHEAD
            constexpr int gg = g - Q2L1_AHEAD, m = gg / 8, f = gg % 8;
#pragma unroll
            for (int r = 0; r < 3; ++r)
                t[f][r] = fma(cf[gg % (Q2L1_AHEAD + 1)][3 * r], uv[m & 1][0],
                              fma(cf[gg % (Q2L1_AHEAD + 1)][3 * r + 1], uv[m & 1][1],
                                  fma(cf[gg % (Q2L1_AHEAD + 1)][3 * r + 2], uv[m & 1][2], t[f][r])));
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    const double *dg = tab + (size_t) ln * 72;                               // the groups of the node itself: diagonal block
    double E8[8];                                                            // (read here: eight register pairs fewer live in the loop above)
#pragma unroll
    for (int f = 0; f < 8; ++f)
        E8[f] = a.Ef[((long long) (a.fx0 + 2 * ex + (f & 1)) * a.fny + (2 * ey + ((f >> 1) & 1))) * a.fnz + 2 * ez + ((f >> 2) & 1)];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            S[r] = fma(E8[f], t[f][r], S[r]);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) M[3 * r + cc] = fma(E8[f], dg[9 * f + 3 * r + cc], M[3 * r + cc]);
        }
}

template <int MODE>      // 1: out = K u; 2: out = zeroDirichlet(b - K u); 3: out = zeroDirichlet(K u) for the nodes of one parity class
__global__ void __launch_bounds__(256) k_q2_level1(Q2L1 a, const double *__restrict__ u, const double *__restrict__ b,
                                                   const uint8_t *__restrict__ mask, double *__restrict__ out) {
    const DimsQ2 &d = a.d;
    // lanes packed over the launch's nodes of an x-plane, row after row (rows of 2^k + 1 nodes leave a wave per row with one lane)
    const int fq = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x, ai = blockIdx.z;
    if (fq >= a.col.cnt[1] * a.col.cnt[2]) return;
    const int bq = fq / a.col.cnt[2], c = fq - bq * a.col.cnt[2];
    const int i = a.col.start[0] + ai * a.col.inc[0], j = a.col.start[1] + bq * a.col.inc[1], k = a.col.start[2] + c * a.col.inc[2];
    // parity of the launch's nodes per axis (uniform): odd = mid node of one element, even = on an element boundary
    const int px = a.col.start[0] & 1, py = a.col.start[1] & 1, pz = a.col.start[2] & 1;
    const int nex = px ? 1 : 2, ney = py ? 1 : 2, nez = pz ? 1 : 2;
    double S[3] = {0.0, 0.0, 0.0}, M[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) M[q] = 0.0;
    for (int sx = 0; sx < nex; ++sx) {
        const int ex = px ? i / 2 : i / 2 - 1 + sx, lx = px ? 1 : (sx ? 0 : 2);
        for (int sy = 0; sy < ney; ++sy) {
            const int ey = py ? j / 2 : j / 2 - 1 + sy, ly = py ? 1 : (sy ? 0 : 2);
            for (int sz = 0; sz < nez; ++sz) {
                const int ez = pz ? k / 2 : k / 2 - 1 + sz, lz = pz ? 1 : (sz ? 0 : 2);
                if (ex < 0 || ex >= d.nx || ey < 0 || ey >= d.ny || ez < 0 || ez >= d.nz) continue;
                q2l1_element(a, u, ex, ey, ez, 9 * lx + 3 * ly + lz, S, M);
            }
        }
    }
    const long long n = ((long long) i * d.NY + j) * d.NZ + k;
    const uint8_t mk = (MODE >= 2 && mask) ? mask[n] : (uint8_t) 0;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        double v = MODE == 2 ? b[3 * n + cc] - S[cc] : S[cc];
        if ((mk >> cc) & 1) v = 0.0;
        out[3 * n + cc] = v;
    }
}

// One colour of the sweep (MG.hh:254-264).  A block holds 64 nodes of a z-row and one wave per incident element of such a node
// (1, 2, 4 or 8 by the colour): every colour touches every element once, so all 27 launches have about the same number of waves
// (elements / 64) whatever the colour, and a node's chain of dependent multiply-adds is one element long.  Partial sums meet
// in LDS and are added in the order of the element loops above.
__global__ void __launch_bounds__(512) k_q2_level1_gs(Q2L1 a, double *__restrict__ u, const double *__restrict__ b,
                                                      const uint8_t *__restrict__ mask, int forward) {
    extern __shared__ double q2l1_part[];               // [waves][12][64] (nothing for a colour with one incident element)
    double (*part)[12][64] = reinterpret_cast<double (*)[12][64]>(q2l1_part);
    const DimsQ2 &d = a.d;
    // the block's 64 nodes are consecutive in the (row, z) order of the colour's nodes of an x-plane: whole blocks whatever the row length
    const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y), fq = blockIdx.x * 64 + lane, ai = blockIdx.z;
    const bool live = fq < a.col.cnt[1] * a.col.cnt[2];
    const int bq = live ? fq / a.col.cnt[2] : 0, c = live ? fq - bq * a.col.cnt[2] : 0;
    const int i = a.col.start[0] + ai * a.col.inc[0], j = a.col.start[1] + bq * a.col.inc[1], k = a.col.start[2] + c * a.col.inc[2];
    const int px = a.col.start[0] & 1, py = a.col.start[1] & 1, pz = a.col.start[2] & 1;
    const int ney = py ? 1 : 2, nez = pz ? 1 : 2;
    const int sx = w / (ney * nez), sy = (w / nez) % ney, sz = w % nez;      // this wave's incident element (uniform)
    const int ex = px ? i / 2 : i / 2 - 1 + sx, lx = px ? 1 : (sx ? 0 : 2);
    const int ey = py ? j / 2 : j / 2 - 1 + sy, ly = py ? 1 : (sy ? 0 : 2);
    const int ez = pz ? k / 2 : k / 2 - 1 + sz, lz = pz ? 1 : (sz ? 0 : 2);
    double S[3] = {0.0, 0.0, 0.0}, M[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) M[q] = 0.0;
    if (live && ex >= 0 && ex < d.nx && ey >= 0 && ey < d.ny && ez >= 0 && ez < d.nz) q2l1_element(a, u, ex, ey, ez, 9 * lx + 3 * ly + lz, S, M);
    if (blockDim.y > 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) part[w][q][lane] = S[q];
#pragma unroll
        for (int q = 0; q < 9; ++q) part[w][3 + q][lane] = M[q];
        __syncthreads();
        if (w != 0) return;
        for (int o = 1; o < (int) blockDim.y; ++o) {
#pragma unroll
            for (int q = 0; q < 3; ++q) S[q] += part[o][q][lane];
#pragma unroll
            for (int q = 0; q < 9; ++q) M[q] += part[o][3 + q][lane];
        }
    }
    if (!live) return;
    const long long n = ((long long) i * d.NY + j) * d.NZ + k;
    double bms[3], ud[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) bms[cc] = b[3 * n + cc] - S[cc];
    gs_solve(bms, M, mask ? mask[n] : (uint8_t) 0, forward != 0, ud);
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) u[3 * n + cc] += ud[cc];
}

static Q2L1 q2l1_args(int nx, int ny, int nz, const double *tab, const double *Ef, int fx0) {
    Q2L1 a{};
    a.d = DimsQ2{nx, ny, nz, 2 * nx + 1, 2 * ny + 1, 2 * nz + 1};
    a.tab = tab; a.Ef = Ef; a.fny = 2 * ny; a.fnz = 2 * nz; a.fx0 = fx0;
    return a;
}
// colours [first, first + count) of the sweep in visiting order (as launch_gs_sweep_q2_level0)
void launch_gs_sweep_q2_level1(int nx, int ny, int nz, const double *tab, const double *Ef, int fx0, double *u, const double *b,
                               const uint8_t *mask, int forward, hipStream_t s, int first, int count) {
    Q2L1 a = q2l1_args(nx, ny, nz, tab, Ef, fx0);
    const int NN[3] = {a.d.NX, a.d.NY, a.d.NZ};
    for (int ci = first; ci < (first + count < 27 ? first + count : 27); ++ci) {
        const int lni = forward ? ci : 26 - ci;                      // MG.hh:293-295
        const int l[3] = {lni / 9, (lni / 3) % 3, lni % 3};
        bool empty = false;
        for (int ax = 0; ax < 3; ++ax) {
            a.col.start[ax] = l[ax];
            a.col.inc[ax] = (l[ax] == 1) ? 2 : 4;                    // MG.hh:301-305
            a.col.cnt[ax] = l[ax] > NN[ax] - 1 ? 0 : (NN[ax] - 1 - l[ax]) / a.col.inc[ax] + 1;
            empty = empty || a.col.cnt[ax] == 0;
        }
        if (empty) continue;
        const int waves = (l[0] == 1 ? 1 : 2) * (l[1] == 1 ? 1 : 2) * (l[2] == 1 ? 1 : 2);
        const dim3 grd((a.col.cnt[1] * a.col.cnt[2] + 63) / 64, 1, a.col.cnt[0]), blk(64, waves, 1);
        k_q2_level1_gs<<<grd, blk, waves > 1 ? (size_t) waves * 12 * 64 * sizeof(double) : 0, s>>>(a, u, b, mask, forward);
    }
    VFEM_HIP(hipGetLastError());
}
// mode 0: out = K u; 1: out = zeroDirichlet(b - K u); 2: out = zeroDirichlet(K u) -- one launch per node-parity class
void launch_apply_q2_level1(int nx, int ny, int nz, const double *tab, const double *Ef, int fx0, const double *u, const double *b,
                            const uint8_t *mask, int mode, double *out, hipStream_t s) {
    Q2L1 a = q2l1_args(nx, ny, nz, tab, Ef, fx0);
    const int NN[3] = {a.d.NX, a.d.NY, a.d.NZ};
    for (int pc = 0; pc < 8; ++pc) {
        const int par[3] = {(pc >> 2) & 1, (pc >> 1) & 1, pc & 1};
        for (int ax = 0; ax < 3; ++ax) {
            a.col.start[ax] = par[ax]; a.col.inc[ax] = 2;
            a.col.cnt[ax] = (NN[ax] - 1 - par[ax]) / 2 + 1;
        }
        const dim3 grd((a.col.cnt[1] * a.col.cnt[2] + 255) / 256, 1, a.col.cnt[0]), blk(64, 4, 1);
        if (mode == 0) k_q2_level1<1><<<grd, blk, 0, s>>>(a, u, b, mask, out);
        else if (mode == 1) k_q2_level1<2><<<grd, blk, 0, s>>>(a, u, b, mask, out);
        else k_q2_level1<3><<<grd, blk, 0, s>>>(a, u, b, mask, out);
    }
    VFEM_HIP(hipGetLastError());
}

// r = zeroDirichlet(b - Ku) (mode 1) or zeroDirichlet(Ku) (mode 2) in place on Ku
__global__ void __launch_bounds__(256) k_q2_residual_fix(long long nn, const double *__restrict__ b, const uint8_t *__restrict__ mask, int mode,
                                                         double *__restrict__ out) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * nn) return;
    double v = out[i];
    if (mode == 1) v = b[i] - v;
    if (mask && ((mask[i / 3] >> (i % 3)) & 1)) v = 0.0;
    out[i] = v;
}
void launch_q2_residual_fix(long long nn, const double *b, const uint8_t *mask, int mode, double *out, hipStream_t s) {
    k_q2_residual_fix<<<dim3((unsigned) ((3 * nn + 255) / 256)), dim3(256), 0, s>>>(nn, b, mask, mode, out);
    VFEM_HIP(hipGetLastError());
}

// g_e = -1/2 gamma rho^(gamma-1) (E0 - Emin) u_e^T K0 u_e, one wave per element (81 dofs over 64 lanes)
__global__ void __launch_bounds__(256) k_gradient_q2(DimsQ2 d, const double *__restrict__ K0, const double *__restrict__ rho,
                                                     double E0, double Emin, double gamma, const double *__restrict__ u,
                                                     double *__restrict__ g) {
    __shared__ double ue[4][81];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long) blockIdx.x * 4 + w;
    const long long ne = (long long) d.nx * d.ny * d.nz;
    if (e >= ne) return;
    const int ez = (int) (e % d.nz), ey = (int) ((e / d.nz) % d.ny), ex = (int) (e / ((long long) d.nz * d.ny));
    for (int q = lane; q < 81; q += 64) {
        const int m = q / 3, cc = q % 3, ma = m / 9, mb = (m / 3) % 3, mc = m % 3;
        ue[w][q] = u[3 * (((long long) (2 * ex + ma) * d.NY + (2 * ey + mb)) * d.NZ + (2 * ez + mc)) + cc];
    }
    __builtin_amdgcn_wave_barrier();
    double acc = 0.0;
    for (int r = lane; r < 81; r += 64) {
        double t = 0.0;
        for (int cidx = 0; cidx < 81; ++cidx) t = fma(K0[r * 81 + cidx], ue[w][cidx], t);
        acc = fma(ue[w][r], t, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) {
        const double r0 = rho[e];
        g[e] = -0.5 * gamma * pow(r0, gamma - 1.0) * (E0 - Emin) * acc;
    }
}

void launch_gradient_q2(int nx, int ny, int nz, const double *K0, const double *rho, double E0, double Emin, double gamma,
                        const double *u, double *g, hipStream_t s) {
    DimsQ2 d{nx, ny, nz, 2 * nx + 1, 2 * ny + 1, 2 * nz + 1};
    const long long ne = (long long) nx * ny * nz;
    k_gradient_q2<<<dim3((unsigned) ((ne + 3) / 4)), dim3(256), 0, s>>>(d, K0, rho, E0, Emin, gamma, u, g);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
