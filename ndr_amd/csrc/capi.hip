// Host side of libvfem: handle types, reference-element setup, multigrid hierarchy, V-cycle / FMG /
// PCG drivers and the extern "C" boundary declared in include/vfem.h.
//
// Control flow follows the reference (paths relative to the reference checkout):
//   MultigridSolver ctor (hierarchy, Dirichlet coarsening)   VoxelFEM/MultigridSolver.hh:22-90
//   vcycle / fullMultigrid / solve / applyPreconditionerInv   VoxelFEM/MultigridSolver.hh:447-553
//   preconditionedConjugateGradient                           VoxelFEM/MultigridSolver.hh:679-732
#include "vfem_internal.h"
#include "gs_coef.h"


#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

namespace vfem {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

// ------------------------------------------------------------------------------------------
// timer registry (BENCHMARK_* of MeshFEM GlobalBenchmark.hh / Timer.hh): host wall time + call count
// per named section; sections enclosing only asynchronous launches measure enqueue time unless the
// caller synchronises (the PCG loop does, once per iteration).
// ------------------------------------------------------------------------------------------
struct TimerEntry { double seconds = 0.0; long long calls = 0; };
static std::map<std::string, TimerEntry> g_timers;
static std::mutex g_timer_mu;
struct ScopedTimer {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit ScopedTimer(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~ScopedTimer() {
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(g_timer_mu);
        auto &e = g_timers[name];
        e.seconds += dt; e.calls += 1;
    }
};

}  // namespace vfem

using namespace vfem;

#define VFEM_TRY try {
#define VFEM_CATCH                                                                              \
    } catch (const std::exception &e) { vfem::set_error(e.what()); return 1; }                  \
      catch (...) { vfem::set_error("unknown error"); return 1; }                               \
    return 0;

static inline hipStream_t S(void *s) { return (hipStream_t) s; }

// ------------------------------------------------------------------------------------------
// reference element: closed-form Q1 stiffness for an axis-aligned box voxel with isotropic C.
// Same quantity as Element_T::Stiffness (TPS.hh:127-140), which integrates it by 2-point Gauss
// quadrature (exact for these integrands); derived here from the 1-D integrals
//   Mm[a][b] = int N_a N_b,  Dd[a][b] = int N_a' N_b',  Gg[a][b] = int N_a' N_b   on [0,1].
// ------------------------------------------------------------------------------------------
void vfem_sim::update_k0() {
    static const double Mm[2][2] = {{1.0 / 3, 1.0 / 6}, {1.0 / 6, 1.0 / 3}};
    static const double Dd[2][2] = {{1.0, -1.0}, {-1.0, 1.0}};
    static const double Gg[2][2] = {{-0.5, -0.5}, {0.5, 0.5}};
    const double vol = h[0] * h[1] * h[2];
    auto I = [&](int n, int m, int p, int q) {   // int d_p N_n d_q N_m over the reference cube, physical gradients
        double v = 1.0 / (h[p] * h[q]);
        for (int dd = 0; dd < 3; ++dd) {
            const int a = (n >> (2 - dd)) & 1, b = (m >> (2 - dd)) & 1;
            if (dd == p && dd == q) v *= Dd[a][b];
            else if (dd == p)       v *= Gg[a][b];
            else if (dd == q)       v *= Gg[b][a];
            else                    v *= Mm[a][b];
        }
        return v;
    };
    for (int n = 0; n < 8; ++n)
        for (int a = 0; a < 3; ++a)
            for (int m = 0; m < 8; ++m)
                for (int b = 0; b < 3; ++b) {
                    double v = lambda * I(n, m, a, b) + mu * I(n, m, b, a);
                    if (a == b) v += mu * (I(n, m, 0, 0) + I(n, m, 1, 1) + I(n, m, 2, 2));
                    K0[(3 * n + a) * 24 + 3 * m + b] = vol * v;
                }
    // mode-space form: Dmode = T K0 T^T / 64 with T = H (x) H (x) H, H = [[1,1],[-1,1]] per axis.
    // For a box voxel with an orthotropic/isotropic tensor only 45 entries survive:
    // 21 diagonal ones (the three rigid translations are null) and 12 symmetric couplings.
    double T[8][8];
    for (int p = 0; p < 8; ++p)
        for (int n = 0; n < 8; ++n) {
            double v = 1.0;
            for (int dd = 0; dd < 3; ++dd) {
                const int pb = (p >> (2 - dd)) & 1, nb = (n >> (2 - dd)) & 1;
                if (pb && !nb) v = -v;
            }
            T[p][n] = v;
        }
    std::vector<double> TK(576), Dfull(576);
    for (int p = 0; p < 8; ++p)
        for (int a = 0; a < 3; ++a)
            for (int col = 0; col < 24; ++col) {
                double v = 0.0;
                for (int n = 0; n < 8; ++n) v += T[p][n] * K0[(3 * n + a) * 24 + col];
                TK[(3 * p + a) * 24 + col] = v;
            }
    double maxabs = 0.0;
    for (int row = 0; row < 24; ++row)
        for (int q = 0; q < 8; ++q)
            for (int b = 0; b < 3; ++b) {
                double v = 0.0;
                for (int m = 0; m < 8; ++m) v += TK[row * 24 + 3 * m + b] * T[q][m];
                Dfull[row * 24 + 3 * q + b] = v / 64.0;
                maxabs = std::max(maxabs, std::fabs(v / 64.0));
            }
    // pack: Dm[0..23] diagonal (3p+a); Dm[24..35] couplings (order fixed in kernels_apply.hip)
    for (int q = 0; q < 64; ++q) Dm[q] = 0.0;
    std::vector<char> used(576, 0);
    for (int r = 0; r < 24; ++r) { Dm[r] = Dfull[r * 24 + r]; used[r * 24 + r] = 1; }
    // coupling list: for each component pair (a<b), third axis t, parity pt of the third axis:
    //   lambda-type: u_a mode (bit a [+ pt*bit t]) <-> u_b mode (bit b [+ pt*bit t])
    //   mu-type:     u_a mode (bit b [+ pt*bit t]) <-> u_b mode (bit a [+ pt*bit t])
    int idx = 24;
    auto bit = [](int axis) { return 1 << (2 - axis); };
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b) {
            const int t = 3 - a - b;
            for (int pt = 0; pt < 2; ++pt)
                for (int type = 0; type < 2; ++type) {
                    const int pa = (type == 0 ? bit(a) : bit(b)) | (pt ? bit(t) : 0);
                    const int pb = (type == 0 ? bit(b) : bit(a)) | (pt ? bit(t) : 0);
                    const int r = 3 * pa + a, c = 3 * pb + b;
                    Dm[idx++] = Dfull[r * 24 + c];
                    used[r * 24 + c] = 1; used[c * 24 + r] = 1;
                }
        }
    fast_ok = true;
    for (int q = 0; q < 576; ++q)
        if (!used[q] && std::fabs(Dfull[q]) > 1e-13 * maxabs) fast_ok = false;
    dK0.alloc(576);
    VFEM_HIP(hipMemcpy(dK0.p, K0, sizeof(K0), hipMemcpyHostToDevice));
    double tab[GS_TABLE_DOUBLES + 36 + 48 + 96];
    vfem::build_gs_table(K0, tab);
    gs_resident_ok = vfem::build_gs_coef(K0, tab + GS_TABLE_DOUBLES);
    vfem::build_gs_coef_parts(tab + GS_TABLE_DOUBLES, tab + GS_TABLE_DOUBLES + 36, tab + GS_TABLE_DOUBLES + 60);
    tune.gs_resident = gs_resident_ok ? 1 : 0;
    {   // K0 by neighbour kind for the node-per-lane marching sweep (class 0 of l1m::build_table applied to K0 itself); it relies on
        // K0[(n^f,a),(m^f,b)] = s_a(f) s_b(f) K0[(n,a),(m,b)] (box voxel, isotropic / orthotropic tensor), checked here
        double full[L1M_TABLE_DOUBLES];
        vfem::build_l1_merged_table(K0, full);
        std::memcpy(tab + GS_TABLE_DOUBLES + 84, full, 96 * sizeof(double));
        double scale = 0.0, err = 0.0;
        for (int q = 0; q < 576; ++q) scale = std::max(scale, std::fabs(K0[q]));
        for (int f = 1; f < 8; ++f)
            for (int n = 0; n < 8; ++n)
                for (int a = 0; a < 3; ++a)
                    for (int m = 0; m < 8; ++m)
                        for (int b = 0; b < 3; ++b) {
                            const double sg = (((f >> (2 - a)) ^ (f >> (2 - b))) & 1) ? -1.0 : 1.0;
                            err = std::max(err, std::fabs(K0[(3 * n + a) * 24 + 3 * m + b] - sg * K0[(3 * (n ^ f) + a) * 24 + 3 * (m ^ f) + b]));
                        }
        k0_mirror_ok = err <= 1e-13 * scale;
    }
    dGsTab.alloc(GS_TABLE_DOUBLES + 36 + 48 + 96);
    VFEM_HIP(hipMemcpy(dGsTab.p, tab, sizeof(tab), hipMemcpyHostToDevice));
}

// ------------------------------------------------------------------------------------------
// multigrid internals
// ------------------------------------------------------------------------------------------
static void coarsen_dirichlet(const Dims &f, const std::vector<uint8_t> &fm, const Dims &c, std::vector<uint8_t> &cm) {
    // MG.hh:57-84: a fine Dirichlet node lying on a coarse element vertex/edge/face constrains all coarse
    // nodes of that entity; a fine Dirichlet node strictly inside a coarse element is an error.
    cm.assign((size_t) c.nn, 0);
    for (int i = 0; i < f.NX; ++i)
        for (int j = 0; j < f.NY; ++j)
            for (int k = 0; k < f.NZ; ++k) {
                const uint8_t m = fm[((size_t) i * f.NY + j) * f.NZ + k];
                if (!m) continue;
                const int g[3] = {i, j, k}, nce[3] = {c.nx, c.ny, c.nz};
                int e[3], lo[3], hi[3];
                bool any = false;
                for (int dd = 0; dd < 3; ++dd) {
                    e[dd] = std::min(g[dd] / 2, nce[dd] - 1);
                    const int loc = g[dd] - 2 * e[dd];
                    if (loc == 0)      { lo[dd] = hi[dd] = e[dd]; any = true; }
                    else if (loc == 2) { lo[dd] = hi[dd] = e[dd] + 1; any = true; }
                    else               { lo[dd] = e[dd]; hi[dd] = e[dd] + 1; }
                }
                if (!any) throw Error("Dirichlet constraints on internal nodes are not supported");
                for (int a = lo[0]; a <= hi[0]; ++a)
                    for (int b = lo[1]; b <= hi[1]; ++b)
                        for (int cc = lo[2]; cc <= hi[2]; ++cc) cm[((size_t) a * c.NY + b) * c.NZ + cc] |= m;
            }
}

static const double *level_K(const vfem_mg *mg, int l) {
    return l == 0 ? mg->fine->dK0.p : mg->cK0.p;
}
// fine-moduli pointer seen by the matrix-free kernels of level l (0 or 1): element (ex,ey,ez) of the level's node grid
// must land on the right entry of the fine array, which may hold extra x-layers (slab decomposition)
static const double *level_E(const vfem_mg *mg, int l) {
    const vfem_sim *sim = mg->fine;
    if (l == 0) return sim->Ep();
    return sim->E.p + 2 * mg->lv[1].ex_lo * (long long) sim->d.ny * sim->d.nz;
}

// level 1 with its operator stored as a stencil (VFEM_OPT_L1_STORED): the kernels of the deeper levels apply
static bool level_uses_stencil(const vfem_mg *mg, int l) {
    const MgLevel &L = mg->lv[l];
    return L.kind == OP_STENCIL || (l == 1 && L.kind == OP_MF1 && mg->fine->tune.l1_stored == 1 && L.S.p);
}
static bool level_uses_half_stencil(const vfem_mg *mg, int l) {
    const MgLevel &L = mg->lv[l];
    return l == 1 && L.kind == OP_MF1 && mg->fine->tune.l1_stored == 2 && L.Sh.p;
}

// level 1 evaluated per mirror class (VFEM_OPT_L1_MERGED): needs the mirror symmetry of the coarsened matrices
static bool level_uses_merged_rows(const vfem_mg *mg, int l) {
    const MgLevel &L = mg->lv[l];
    return l == 1 && L.kind == OP_MF1 && mg->mf1_sym && mg->fine->tune.l1_merged && mg->fine->tune.gs_variant == 0 && mg->l1mtab.p &&
           l1_merged_usable(L.d);
}

static void mg_apply(vfem_mg *mg, int l, const double *u, const double *b, int res, double *out, hipStream_t s) {
    MgLevel &L = mg->lv[l];
    if (level_uses_half_stencil(mg, l)) launch_apply_stencil_half(L.d, L.Sh.p, u, b, L.maskp, res, out, s);
    else if (level_uses_stencil(mg, l)) launch_apply_stencil(L.d, L.S.p, u, b, L.maskp, res, out, s);
    else if (L.kind == OP_MF0 && mg->fine->fast_ok) {
        const vfem_sim *sim = mg->fine;
        const Tuning &t = sim->tune;
        if (t.apply_impl == 0 &&
            launch_apply_dma(L.d, sim->Dm, level_E(mg, 0), sim->E.p + sim->n_store(), u, out, s, 0, -1, t.dma_chunks, t.dma_strip,
                             res ? b : nullptr, res ? L.maskp : nullptr, t.dma_lx)) return;
        launch_apply_fast(L.d, sim->Dm, level_E(mg, 0), u, b, L.maskp, res, out, s, t.apply_pd);
    }
    else if (level_uses_merged_rows(mg, l)) launch_l1_merged_apply(L.d, mg->l1mtab.p, level_E(mg, l), u, b, L.maskp, res, out, s);
    else launch_apply_gather(L.d, L.kind, level_K(mg, l), level_E(mg, l), u, b, L.maskp, res, out, s);
}

static void mg_smooth(vfem_mg *mg, int l, double *u, const double *b, int forward, hipStream_t s, int first = 0, int count = 8) {
    MgLevel &L = mg->lv[l];
    if (level_uses_half_stencil(mg, l)) launch_gs_sweep_stencil_half(L.d, L.Sh.p, u, b, L.maskp, forward, L.xparity, first, count, s);
    else if (level_uses_stencil(mg, l)) launch_gs_sweep_stencil(L.d, L.S.p, u, b, L.maskp, forward, L.xparity, first, count, s, L.Sn.p, mg->fine->tune.stencil_split);
    else if (level_uses_merged_rows(mg, l)) launch_l1_merged_sweep(L.d, mg->l1mtab.p, level_E(mg, l), u, b, L.maskp, forward, L.xparity, first, count, s,
                                                                   mg->fine->tune.l1_merged == 2);
    else launch_gs_sweep_mf(L.d, L.kind, level_K(mg, l), l == 0 ? mg->fine->dGsTab.p : mg->mf1diag.p, level_E(mg, l), u, b, L.maskp,
                            forward, L.xparity, first, count, s, mg->fine->tune, mg->mf1_sym,
                            (l == 1 && mg->fine->tune.l1_diag) ? L.Mdiag.p : nullptr);
}

// gs_march: 0 row kernels, 1 marching kernel on grids where it wins (measured per sweep, marching / rows, profiles/r04_gs_march_chunks.txt:
// 512^3 6.27 / 9.76 ms, 256^3 0.98 / 1.43, 160^3 0.26 / 0.44, 128^3 0.20 / 0.23, 96^3 0.08 / 0.105; slabs 64 x 512^2 0.90 / 1.42,
// 32 x 256^2 0.18 / 0.23), 2 marching kernel always
static bool gs_march_wanted(const MgLevel &L, const Tuning &t) {
    return t.gs_march == 2 || (t.gs_march == 1 && L.d.nn >= 800000);
}

// the marching sweep sums a node row per neighbour kind, which needs K0 with the mirror symmetry of a box voxel and an isotropic tensor:
// the neighbour-kind table of K0, or null (then the row kernels do the sweep)
static const double *march_table(const vfem_sim *sim) {
    return sim->k0_mirror_ok ? sim->dGsTab.p + GS_TABLE_DOUBLES + 84 : nullptr;
}

// level 0: solve data of the marching sweeps, recomputed when the moduli (or the material / Dirichlet mask: both bump the version) changed
static void gs_solve_data(vfem_mg *mg, hipStream_t s) {
    MgLevel &L = mg->lv[0];
    const vfem_sim *sim = mg->fine;
    if (L.gs_sd.p && L.gs_sd_version == sim->operator_version) return;
    L.gs_sd.reserve((size_t) L.d.nn * 3);
    launch_gs_solve_data(L.d, sim->dK0.p, level_E(mg, 0), L.maskp, L.gs_sd.p, s);
    L.gs_sd_version = sim->operator_version;
}

// n consecutive sweeps of level l in one direction.  Level 0 runs them as marching half sweeps (kernels_gs_march.hip) when
// it can: those are out of place, so the planes of either parity alternate between u and the level's scratch vector; an even
// number of sweeps ends in u, an odd one is followed by a copy of the planes left in the scratch vector.
static void mg_smooth_n(vfem_mg *mg, int l, double *u, const double *b, int forward, int n, hipStream_t s) {
    MgLevel &L = mg->lv[l];
    const vfem_sim *sim = mg->fine;
    const Tuning &t = sim->tune;
    if (!(l == 0 && L.kind == OP_MF0 && gs_march_wanted(L, t) && t.gs_variant == 0 && t.gs_resident && sim->dGsTab.p && sim->k0_mirror_ok && n > 0)) {
        for (int i = 0; i < n; ++i) mg_smooth(mg, l, u, b, forward, s);
        return;
    }
    L.tmp.reserve((size_t) L.d.nn * 3);
    gs_solve_data(mg, s);
    double *cur[2] = {u, u};                         // where the planes of local parity 0 / 1 currently live
    for (int i = 0; i < n; ++i)
        for (int half = 0; half < 2; ++half) {
            const int cx = forward ? half : 1 - half;                    // colour groups 0-3 / 4-7 of MG.hh:292-310, reversed for a backward sweep
            const int cxl = cx ^ (L.xparity & 1);
            if (cxl > L.d.NX - 1) continue;
            double *dst = cur[cxl] == u ? L.tmp.p : u;
            if (!launch_gs_march_mf0(L.d, march_table(sim), level_E(mg, 0), cur[cxl], cur[1 - cxl], dst, b, L.gs_sd.p,
                                     cxl, forward, t.gs_march_chunks, s, 0, -1)) {
                // buffers the kernel cannot take: finish in place with the row kernels
                for (int par = 0; par < 2; ++par)
                    if (cur[par] != u) { launch_copy_planes(L.d, par, cur[par], u, s); cur[par] = u; }
                mg_smooth(mg, l, u, b, forward, s, 4 * half, 4);
                continue;
            }
            cur[cxl] = dst;
        }
    for (int par = 0; par < 2; ++par)
        if (cur[par] != u) launch_copy_planes(L.d, par, cur[par], u, s);
}

// one colour group (half sweep `half` of the sweep order) of level 0 by the marching kernel, result back in u; false: not available
static bool mg_smooth_half(vfem_mg *mg, int l, double *u, const double *b, int forward, int half, hipStream_t s, int plane_lo = 0, int plane_hi = -1) {
    MgLevel &L = mg->lv[l];
    const vfem_sim *sim = mg->fine;
    const Tuning &t = sim->tune;
    if (!(l == 0 && L.kind == OP_MF0 && gs_march_wanted(L, t) && t.gs_variant == 0 && t.gs_resident && sim->dGsTab.p && sim->k0_mirror_ok)) return false;
    const int cx = forward ? half : 1 - half;
    const int cxl = cx ^ (L.xparity & 1);
    if (cxl > L.d.NX - 1) return true;
    L.tmp.reserve((size_t) L.d.nn * 3);
    gs_solve_data(mg, s);
    if (!launch_gs_march_mf0(L.d, march_table(sim), level_E(mg, 0), u, u, L.tmp.p, b, L.gs_sd.p,
                             cxl, forward, t.gs_march_chunks, s, plane_lo, plane_hi)) return false;
    launch_copy_planes(L.d, cxl, L.tmp.p, u, s, plane_lo, plane_hi);
    return true;
}

static void coarsest_solve(vfem_mg *mg, const double *b, double *x, hipStream_t s) {
    const long long n = 3 * mg->lv[mg->L].d.nn;
    launch_gemv_sym(n, mg->Ainv.p, b, x, s);
}

static void update_operators(vfem_mg *mg, hipStream_t s) {
    vfem_sim *sim = mg->fine;
    // the reference rebuilds the coarse operators at the start of every solve (MG.hh:690-691) because its densities may have
    // changed; here the simulator counts the changes, so a solve on unchanged moduli (a second right-hand side, the objective's
    // constructor solve followed by setVars with the same design) keeps Galerkin matrices, stencils and the dense inverse
    if (mg->operators_valid && mg->operators_version == sim->operator_version) return;
    ScopedTimer tm("updateElementStiffnessMatrices");
    const int L = mg->slab ? mg->L - 1 : mg->L;      // the last level of a slab hierarchy only serves the grid transfers
    // Galerkin element matrices for levels >= 2 (level 1 stays virtual: sum_f E_f cK0[f])
    // (element arrays cover lv.da = node grid + extra x-layers; array origins halve exactly from level to level)
    // (a replicated coarse hierarchy of a slab decomposition may be handed the element matrices of its first active level,
    // vfem_mg_import_level_ke: it then never looks at its simulator's moduli)
    for (int l = mg->external_ke_level > 0 ? mg->external_ke_level + 1 : 2; l <= L; ++l) {
        MgLevel &lv = mg->lv[l];
        lv.Ke.alloc((size_t) lv.da.ne * 576);
        if (l == 2) launch_coarsen_ke(lv.da, 3, mg->c2K0.p, sim->E.p, nullptr, lv.Ke.p, s);
        else        launch_coarsen_ke(lv.da, 2, nullptr, nullptr, mg->lv[l - 1].Ke.p, lv.Ke.p, s);
    }
    if (mg->L >= 1 && mg->first_active <= 1 && mg->mf1_sym && sim->tune.l1_diag) {   // level 1: diagonal blocks of the virtual operator
        MgLevel &l1 = mg->lv[1];
        l1.Mdiag.alloc((size_t) l1.d.nn * 9);
        launch_mf1_diag(l1.d, mg->mf1diag.p, level_E(mg, 1), l1.Mdiag.p, s);
    }
    if (mg->L >= 1 && mg->first_active <= 1 && !mg->slab) {                           // level 1 stored as a stencil (option, off by default)
        MgLevel &l1 = mg->lv[1];
        if (sim->tune.l1_stored == 1 && l1.kind == OP_MF1 && L >= 2) {
            l1.S.alloc((size_t) stencil_storage_doubles(l1.d));
            launch_stencil_from_mf(l1.d, OP_MF1, level_K(mg, 1), level_E(mg, 1), l1.S.p, s);
            l1.Sn.release();
            l1.Sh.release();
        } else if (sim->tune.l1_stored == 2 && l1.kind == OP_MF1 && L >= 2) {
            l1.Sh.alloc((size_t) stencil_half_storage_doubles(l1.d));
            launch_stencil_half_from_mf1(l1.d, level_K(mg, 1), level_E(mg, 1), l1.Sh.p, s);
            l1.S.release();
        } else if (l1.kind == OP_MF1) { l1.S.release(); l1.Sh.release(); }
    }
    for (int l = std::max(2, mg->first_active); l <= L; ++l) {
        MgLevel &lv = mg->lv[l];
        lv.S.alloc((size_t) stencil_storage_doubles(lv.d));
        launch_stencil_from_ke(lv.d, lv.Ke.p + lv.ex_lo * (long long) lv.d.ny * lv.d.nz * 576, lv.S.p, s);
        if (lv.d.nn <= WAVE_SWEEP_MAX_NODES) {
            lv.Sn.alloc((size_t) stencil_storage_doubles(lv.d));
            launch_stencil_node_major(lv.d, lv.S.p, lv.Sn.p, s);
        } else lv.Sn.release();
    }
    if (mg->slab) { mg->operators_valid = true; mg->operators_version = sim->operator_version; return; }       // the coarse levels live in the replicated hierarchy
    // coarsest level: dense inverse
    MgLevel &cl = mg->lv[L];
    const long long n = 3 * cl.d.nn;
    if (n > 40000) throw Error("coarsest grid too large for the dense coarsest-level solve (" + std::to_string(n) +
                               " dofs); use more coarsening levels");
    const double *Sc = cl.S.p;
    DevBuf<double> tmpS;
    if (L < 2) {
        tmpS.alloc((size_t) stencil_storage_doubles(cl.d));
        launch_stencil_from_mf(cl.d, L == 0 ? OP_MF0 : OP_MF1, level_K(mg, L), level_E(mg, L), tmpS.p, s);
        Sc = tmpS.p;
    }
    mg->Ainv.alloc((size_t) n * n);
    mg->Ainv.zero(s);
    launch_dense_from_stencil(cl.d, Sc, cl.maskp, mg->Ainv.p, s);
    dense_spd_inverse(n, mg->Ainv.p, mg->dense, s);      // own kernels, fixed summation order (dense_spd.hip)
    // three further n x n work matrices: kept between operator updates while they are small (2 187 dofs: 115 MB), released when the
    // coarsest level is large -- they would otherwise be held for the hierarchy's lifetime at three times the inverse's size
    if ((size_t) n * (size_t) n * sizeof(double) > ((size_t) 256 << 20)) {
        VFEM_HIP(hipStreamSynchronize(s));
        mg->dense.L.release(); mg->dense.X.release(); mg->dense.Tm.release();
    }
    launch_dense_finish_inverse(n, cl.maskp, mg->Ainv.p, s);
    VFEM_HIP(hipStreamSynchronize(s));   // tmpS lifetime
    mg->operators_valid = true;
    mg->operators_version = sim->operator_version;
}

// vcycle, MG.hh:516-553
// dirichlet_zeroed: the level's iterate has zeros at the Dirichlet components already (just zeroed by the restriction of the level
// above, or interpolated with the mask by full_multigrid), which is all the residual system asks for (MG.hh:521-523)
static void vcycle(vfem_mg *mg, int l, int nsmooth, bool residual_system, hipStream_t s, bool dirichlet_zeroed = false) {
    MgLevel &L = mg->lv[l];
    if (l == mg->L) { coarsest_solve(mg, L.b.p, L.x.p, s); return; }
    MgLevel &C = mg->lv[l + 1];
    if (!(dirichlet_zeroed && residual_system))
        launch_enforce_dirichlet(L.d.nn, L.maskp, l == 0 ? mg->fine->dvals.p : nullptr, L.x.p, residual_system ? 1 : 0, s);
    mg_smooth_n(mg, l, L.x.p, L.b.p, 1, nsmooth, s);
    mg_apply(mg, l, L.x.p, L.b.p, 1, L.r.p, s);                       // computeResidual (Dirichlet zeroed)
    launch_restrict(C.d, L.d.NX, C.xshift, L.r.p, C.b.p, s, C.x.p);  // ... and the zero initial guess of the coarse level
    vcycle(mg, l + 1, nsmooth, true, s, true);
    launch_prolong(C.d, L.d.NX, C.xshift, C.x.p, L.x.p, 1, s);
    mg_smooth_n(mg, l, L.x.p, L.b.p, mg->symmetric_gs ? 0 : 1, nsmooth, s);
}

// fullMultigrid, MG.hh:486-508
static void full_multigrid(vfem_mg *mg, int l, int nsmooth, bool residual_system, hipStream_t s) {
    MgLevel &L = mg->lv[l];
    if (l == mg->L) { coarsest_solve(mg, L.b.p, L.x.p, s); return; }
    MgLevel &C = mg->lv[l + 1];
    launch_restrict(C.d, L.d.NX, C.xshift, L.b.p, C.b.p, s);
    full_multigrid(mg, l + 1, nsmooth, residual_system, s);
    launch_prolong(C.d, L.d.NX, C.xshift, C.x.p, L.x.p, 0, s, residual_system ? L.maskp : nullptr);
    vcycle(mg, l, nsmooth, residual_system, s, residual_system);
}

// MG::solve on the level-0 work vectors (x[0], b[0] already set), MG.hh:457-471
static void mg_cycles(vfem_mg *mg, int num_steps, int nsmooth, bool zero_dirichlet, bool fmg, hipStream_t s) {
    if (fmg) {
        full_multigrid(mg, 0, nsmooth, zero_dirichlet, s);
        for (int i = 1; i < num_steps; ++i) vcycle(mg, 0, nsmooth, zero_dirichlet, s);
    } else {
        for (int i = 0; i < num_steps; ++i) vcycle(mg, 0, nsmooth, zero_dirichlet, s);
    }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char *vfem_last_error(void) { return vfem::g_err.c_str(); }
int vfem_version(void) { return 100; }
#ifdef VFEM_ABLATION
// timing ablations with WRONG results (tools/ only): exists in `make ablation` builds of the library, never in the shipped one
int vfem_debug_set(int key, int value) {
    if (key == 1) vfem::g_ablate_apply = value;
    else if (key == 3) vfem::g_ablate_store = value;
    else if (key == 8) vfem::g_ablate_mlp = value;
    else return 1;
    return 0;
}
#endif

int vfem_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int vfem_set_device(int device) { VFEM_TRY VFEM_HIP(hipSetDevice(device)); VFEM_CATCH }

int vfem_malloc(void **ptr, size_t bytes) { VFEM_TRY VFEM_HIP(hipMalloc(ptr, bytes)); VFEM_CATCH }
int vfem_free(void *ptr) { VFEM_TRY VFEM_HIP(hipFree(ptr)); VFEM_CATCH }
int vfem_copy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    VFEM_TRY
    VFEM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream)));
    VFEM_HIP(hipStreamSynchronize(S(stream)));
    VFEM_CATCH
}
int vfem_copy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    VFEM_TRY
    VFEM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream)));
    VFEM_HIP(hipStreamSynchronize(S(stream)));
    VFEM_CATCH
}
int vfem_copy_d2d(void *dst, const void *src, size_t bytes, void *stream) {
    VFEM_TRY VFEM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(stream))); VFEM_CATCH
}
int vfem_memset(void *dst, int value, size_t bytes, void *stream) {
    VFEM_TRY VFEM_HIP(hipMemsetAsync(dst, value, bytes, S(stream))); VFEM_CATCH
}
int vfem_stream_sync(void *stream) { VFEM_TRY VFEM_HIP(hipStreamSynchronize(S(stream))); VFEM_CATCH }

// ---- simulator ----
static thread_local long long g_next_sim_extra[2] = {0, 0};
int vfem_sim_set_next_element_padding(int64_t extra_lo, int64_t extra_hi) {
    if (extra_lo < 0 || extra_hi < 0) { vfem::set_error("negative padding"); return 1; }
    g_next_sim_extra[0] = extra_lo; g_next_sim_extra[1] = extra_hi;
    return 0;
}
int vfem_sim_create(vfem_sim **out, const double bbmin[3], const double bbmax[3], const int64_t ne[3]) {
    VFEM_TRY
    for (int dd = 0; dd < 3; ++dd)
        if (ne[dd] < 1 || ne[dd] > 4096) throw Error("elements per dimension must be in [1, 4096]");
    std::unique_ptr<vfem_sim> sim(new vfem_sim);
    sim->d = Dims(ne[0], ne[1], ne[2]);
    for (int dd = 0; dd < 3; ++dd) {
        sim->bbmin[dd] = bbmin[dd]; sim->bbmax[dd] = bbmax[dd];
        sim->h[dd] = (bbmax[dd] - bbmin[dd]) / (double) ne[dd];          // TPS.hh:287
        if (!(sim->h[dd] > 0)) throw Error("empty domain bounding box");
    }
    sim->update_k0();
    sim->ex_lo = g_next_sim_extra[0]; sim->ex_hi = g_next_sim_extra[1];
    g_next_sim_extra[0] = g_next_sim_extra[1] = 0;
    sim->rho.alloc((size_t) sim->n_store());   sim->rho.zero(nullptr);
    sim->E.alloc((size_t) sim->n_store());
    launch_simp(sim->n_store(), sim->rho.p, sim->E0, sim->Emin, sim->gamma, sim->E.p, nullptr);
    sim->dmask.alloc((size_t) sim->d.nn); sim->dmask.zero(nullptr);
    sim->dvals.alloc((size_t) sim->d.nn * 3); sim->dvals.zero(nullptr);
    sim->loads.alloc((size_t) sim->d.nn * 3); sim->loads.zero(nullptr);
    sim->hmask.assign((size_t) sim->d.nn, 0);
    sim->hvals.assign((size_t) sim->d.nn * 3, 0.0);
    sim->red.alloc(2048 + 8);
    VFEM_HIP(hipDeviceSynchronize());
    *out = sim.release();
    VFEM_CATCH
}
int vfem_sim_destroy(vfem_sim *sim) { VFEM_TRY delete sim; VFEM_CATCH }
int64_t vfem_sim_num_nodes(const vfem_sim *sim) { return sim->d.nn; }
int64_t vfem_sim_num_elements(const vfem_sim *sim) { return sim->d.ne; }
int64_t vfem_sim_num_stored_elements(const vfem_sim *sim) { return sim->n_store(); }

int vfem_sim_set_isotropic(vfem_sim *sim, double young, double poisson) {
    VFEM_TRY
    sim->lambda = poisson * young / ((1.0 + poisson) * (1.0 - 2.0 * poisson));   // ElasticityTensor.hh:105-106
    sim->mu = young / (2.0 + 2.0 * poisson);
    sim->update_k0();
    ++sim->operator_version;
    VFEM_CATCH
}
int vfem_sim_set_simp(vfem_sim *sim, double E0, double Emin, double gamma) {
    VFEM_TRY
    sim->E0 = E0; sim->Emin = Emin; sim->gamma = gamma;
    ++sim->operator_version;
    launch_simp(sim->n_store(), sim->rho.p, E0, Emin, gamma, sim->E.p, nullptr);
    VFEM_HIP(hipDeviceSynchronize());
    VFEM_CATCH
}
int vfem_sim_set_option(vfem_sim *sim, int key, int value) {
    VFEM_TRY
    Tuning &t = sim->tune;
    switch (key) {
        case VFEM_OPT_APPLY_PLANES:  if (value < 2 || value > 4) throw Error("planes in flight must be 2..4"); t.apply_pd = value; break;
        case VFEM_OPT_GS_VARIANT:    t.gs_variant = value != 0; break;
        case VFEM_OPT_APPLY_IMPL:    t.apply_impl = value != 0; break;
        case VFEM_OPT_DMA_CHUNKS:    if (value < 0) throw Error("negative chunk count"); t.dma_chunks = value; break;
        case VFEM_OPT_DMA_STRIP:     if (value < 0 || value > 2) throw Error("strip mode must be 0..2"); t.dma_strip = value; break;
        case VFEM_OPT_DMA_LX:        if (value < 0 || value > 2) throw Error("line-exclusive tiling mode must be 0..2"); t.dma_lx = value; break;
        case VFEM_OPT_GS_PAIR:       t.gs_pair = value != 0; break;
        case VFEM_OPT_GS_RESIDENT:   t.gs_resident = (value != 0 && sim->gs_resident_ok) ? 1 : 0; break;
        case VFEM_OPT_L1_SPLIT:      if (value != 1 && value != 2 && value != 4 && value != 8) throw Error("level-1 slot split 1, 2, 4 or 8"); t.l1_split = value; break;
        case VFEM_OPT_STENCIL_SPLIT: t.stencil_split = value != 0; break;
        case VFEM_OPT_GS_MARCH:      if (value < 0 || value > 2) throw Error("marching sweep mode 0..2"); t.gs_march = value; break;
        case VFEM_OPT_GS_MARCH_CHUNKS: if (value < 0) throw Error("negative chunk count"); t.gs_march_chunks = value; break;
        case VFEM_OPT_L1_STORED:     if (value < 0 || value > 2) throw Error("level-1 storage mode 0..2"); t.l1_stored = value; ++sim->operator_version; break;
        case VFEM_OPT_L1_MERGED:     if (value < 0 || value > 2) throw Error("level-1 row mode 0..2"); t.l1_merged = value; break;
        case VFEM_OPT_L1_DIAG:       t.l1_diag = value != 0; ++sim->operator_version; break;   // hierarchies (re)build the blocks
        default: throw Error("unknown simulator option " + std::to_string(key));
    }
    VFEM_CATCH
}
int vfem_sim_k0(const vfem_sim *sim, double *K0_host) {
    VFEM_TRY std::memcpy(K0_host, sim->K0, sizeof(sim->K0)); VFEM_CATCH
}
int vfem_sim_set_dirichlet(vfem_sim *sim, const uint8_t *mask_host, const double *values_host) {
    VFEM_TRY
    sim->hmask.assign(mask_host, mask_host + sim->d.nn);
    sim->nonzero_dirichlet = false;
    if (values_host) {
        sim->hvals.assign(values_host, values_host + 3 * sim->d.nn);
        for (long long n = 0; n < sim->d.nn; ++n)
            for (int c = 0; c < 3; ++c)
                if (((sim->hmask[n] >> c) & 1) && sim->hvals[3 * n + c] != 0.0) sim->nonzero_dirichlet = true;
    } else sim->hvals.assign((size_t) sim->d.nn * 3, 0.0);
    ++sim->operator_version;                    // (the level-0 solve data of the marching sweeps carries the mask)
    VFEM_HIP(hipMemcpy(sim->dmask.p, sim->hmask.data(), (size_t) sim->d.nn, hipMemcpyHostToDevice));
    VFEM_HIP(hipMemcpy(sim->dvals.p, sim->hvals.data(), (size_t) sim->d.nn * 3 * sizeof(double), hipMemcpyHostToDevice));
    VFEM_CATCH
}
int vfem_sim_set_loads(vfem_sim *sim, const double *f, void *stream) {
    VFEM_TRY
    VFEM_HIP(hipMemcpyAsync(sim->loads.p, f, (size_t) sim->d.nn * 3 * sizeof(double), hipMemcpyDeviceToDevice, S(stream)));
    VFEM_CATCH
}
int vfem_sim_build_load_vector(const vfem_sim *sim, double *f, void *stream) {
    VFEM_TRY
    VFEM_HIP(hipMemcpyAsync(f, sim->loads.p, (size_t) sim->d.nn * 3 * sizeof(double), hipMemcpyDeviceToDevice, S(stream)));
    VFEM_CATCH
}
int vfem_sim_set_densities(vfem_sim *sim, const double *rho, void *stream) {
    VFEM_TRY
    VFEM_HIP(hipMemcpyAsync(sim->rho.p, rho, (size_t) sim->n_store() * sizeof(double), hipMemcpyDeviceToDevice, S(stream)));
    ++sim->operator_version;
    launch_simp(sim->n_store(), sim->rho.p, sim->E0, sim->Emin, sim->gamma, sim->E.p, S(stream));
    VFEM_CATCH
}
int vfem_sim_set_uniform_density(vfem_sim *sim, double rho, void *stream) {
    VFEM_TRY
    if (rho > 1.0 || rho < 0.0)
        throw Error("Density value (" + std::to_string(rho) + ") has to be in between 0 and 1");   // TPS.hh:457-458
    ++sim->operator_version;
    launch_fill(sim->n_store(), rho, sim->rho.p, S(stream));
    launch_simp(sim->n_store(), sim->rho.p, sim->E0, sim->Emin, sim->gamma, sim->E.p, S(stream));
    VFEM_CATCH
}
int vfem_sim_get_densities(const vfem_sim *sim, double *rho, void *stream) {
    VFEM_TRY
    VFEM_HIP(hipMemcpyAsync(rho, sim->rho.p, (size_t) sim->n_store() * sizeof(double), hipMemcpyDeviceToDevice, S(stream)));
    VFEM_CATCH
}
int vfem_sim_apply_k(const vfem_sim *sim, const double *u, double *out, int variant, void *stream) {
    VFEM_TRY
    ScopedTimer tm("applyK");
    if (variant != 1 && sim->fast_ok) {
        bool done = false;
        if (variant == 0 && sim->tune.apply_impl == 0)
            done = launch_apply_dma(sim->d, sim->Dm, sim->Ep(), sim->E.p + sim->n_store(), u, out, S(stream), 0, -1,
                                    sim->tune.dma_chunks, sim->tune.dma_strip, nullptr, nullptr, sim->tune.dma_lx);
        if (!done) launch_apply_fast(sim->d, sim->Dm, sim->Ep(), u, nullptr, nullptr, 0, out, S(stream), sim->tune.apply_pd);
    }
    else launch_apply_gather(sim->d, OP_MF0, sim->dK0.p, sim->Ep(), u, nullptr, nullptr, 0, out, S(stream));
    VFEM_CATCH
}
int vfem_sim_apply_k_planes(const vfem_sim *sim, const double *u, double *out, int64_t plane_lo, int64_t plane_hi, void *stream) {
    VFEM_TRY
    if (plane_lo < 0 || plane_hi > sim->d.NX - 1) throw Error("plane range outside the node grid");
    if (plane_lo > plane_hi) return 0;
    if (!sim->fast_ok) throw Error("plane-range apply needs the mode-space kernel (box voxels, isotropic tensor)");
    if (!launch_apply_dma(sim->d, sim->Dm, sim->Ep(), sim->E.p + sim->n_store(), u, out, S(stream), (int) plane_lo, (int) plane_hi,
                          sim->tune.dma_chunks, sim->tune.dma_strip, nullptr, nullptr, sim->tune.dma_lx))
        throw Error("plane-range apply needs 8-byte aligned device buffers");
    VFEM_CATCH
}
int vfem_sim_compliance_gradient(const vfem_sim *sim, const double *u, double *g, void *stream) {
    VFEM_TRY
    launch_compliance_gradient(sim->d, sim->dK0.p, sim->rhop(), sim->E0, sim->Emin, sim->gamma, u, g, S(stream));
    VFEM_CATCH
}
int vfem_compliance(const vfem_sim *sim, const double *f, const double *u, double *value_host, void *stream) {
    VFEM_TRY
    // stream-ordered scratch: two evaluations of one simulator on different streams must not share partial sums (the simulator's
    // persistent `red` buffer did, and the handle is const here)
    double *tmp = nullptr;
    VFEM_HIP(hipMallocAsync((void **) &tmp, (2048 + 8) * sizeof(double), S(stream)));
    launch_dot(3 * sim->d.nn, f, u, tmp + 8, tmp, S(stream));
    double v = 0.0;
    VFEM_HIP(hipMemcpyAsync(&v, tmp, sizeof(double), hipMemcpyDeviceToHost, S(stream)));
    VFEM_HIP(hipFreeAsync(tmp, S(stream)));
    VFEM_HIP(hipStreamSynchronize(S(stream)));
    *value_host = 0.5 * v;
    VFEM_CATCH
}

// ---- multigrid ----
static void finish_mg_create(vfem_mg *mg) {
    vfem_sim *fine = mg->fine;
    // coarsened reference matrices cK0[g] = I_g^T K0 I_g (MG.hh:644-648), children g = 4gx+2gy+gz
    std::vector<double> c(8 * 576, 0.0), T(576);
    for (int g = 0; g < 8; ++g) {
        double ph[8][8];
        for (int fn = 0; fn < 8; ++fn)
            for (int cn = 0; cn < 8; ++cn) {
                double v = 1.0;
                for (int dd = 0; dd < 3; ++dd) {
                    const int sh = 2 - dd;
                    const double p = 0.5 * ((fn >> sh) & 1) + 0.5 * ((g >> sh) & 1);
                    v *= ((cn >> sh) & 1) ? p : (1.0 - p);
                }
                ph[fn][cn] = v;
            }
        for (int a = 0; a < 24; ++a)
            for (int j = 0; j < 8; ++j)
                for (int dd = 0; dd < 3; ++dd) {
                    double v = 0.0;
                    for (int i = 0; i < 8; ++i) v += fine->K0[a * 24 + 3 * i + dd] * ph[i][j];
                    T[a * 24 + 3 * j + dd] = v;
                }
        for (int j = 0; j < 8; ++j)
            for (int cc = 0; cc < 3; ++cc)
                for (int b = 0; b < 24; ++b) {
                    double v = 0.0;
                    for (int i = 0; i < 8; ++i) v += ph[i][j] * T[(3 * i + cc) * 24 + b];
                    c[(size_t) g * 576 + (3 * j + cc) * 24 + b] = v;
                }
    }
    mg->cK0.alloc(8 * 576);
    VFEM_HIP(hipMemcpy(mg->cK0.p, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice));
    {   // c2K0[g][f] = I_g^T cK0[f] I_g: level-2 element matrices are a plain weighted sum of these 64 over the fine moduli
        std::vector<double> c2((size_t) 64 * 576, 0.0);
        for (int g = 0; g < 8; ++g) {
            double ph[8][8];
            for (int fn = 0; fn < 8; ++fn)
                for (int cn = 0; cn < 8; ++cn) {
                    double v = 1.0;
                    for (int dd = 0; dd < 3; ++dd) {
                        const int sh = 2 - dd;
                        const double p = 0.5 * ((fn >> sh) & 1) + 0.5 * ((g >> sh) & 1);
                        v *= ((cn >> sh) & 1) ? p : (1.0 - p);
                    }
                    ph[fn][cn] = v;
                }
            for (int f = 0; f < 8; ++f) {
                const double *Kf = c.data() + (size_t) f * 576;
                for (int a = 0; a < 24; ++a)
                    for (int j = 0; j < 8; ++j)
                        for (int dd = 0; dd < 3; ++dd) {
                            double v = 0.0;
                            for (int i = 0; i < 8; ++i) v += Kf[a * 24 + 3 * i + dd] * ph[i][j];
                            T[a * 24 + 3 * j + dd] = v;
                        }
                double *out = c2.data() + (size_t) (g * 8 + f) * 576;
                for (int j = 0; j < 8; ++j)
                    for (int cc = 0; cc < 3; ++cc)
                        for (int b = 0; b < 24; ++b) {
                            double v = 0.0;
                            for (int i = 0; i < 8; ++i) v += ph[i][j] * T[(3 * i + cc) * 24 + b];
                            out[(3 * j + cc) * 24 + b] = v;
                        }
            }
        }
        mg->c2K0.alloc(c2.size());
        VFEM_HIP(hipMemcpy(mg->c2K0.p, c2.data(), c2.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    mg->mf1_sym = coarsened_matrices_are_mirror_images(c.data());
    {
        double dt[96];
        build_mf1_diag_table(c.data(), dt);
        mg->mf1diag.alloc(96);
        VFEM_HIP(hipMemcpy(mg->mf1diag.p, dt, sizeof(dt), hipMemcpyHostToDevice));
        double mt[L1M_TABLE_DOUBLES];
        build_l1_merged_table(c.data(), mt);
        mg->l1mtab.alloc(L1M_TABLE_DOUBLES);
        VFEM_HIP(hipMemcpy(mg->l1mtab.p, mt, sizeof(mt), hipMemcpyHostToDevice));
    }
    for (int l = mg->first_active; l <= mg->L; ++l) {
        MgLevel &lv = mg->lv[(size_t) l];
        lv.x.alloc((size_t) lv.d.nn * 3); lv.b.alloc((size_t) lv.d.nn * 3); lv.r.alloc((size_t) lv.d.nn * 3);
        lv.x.zero(nullptr); lv.b.zero(nullptr); lv.r.zero(nullptr);
    }
    if (mg->first_active == 0 && !mg->slab) {
        const size_t n3 = (size_t) fine->d.nn * 3;
        mg->pd.alloc(n3); mg->pAd.alloc(n3); mg->ps.alloc(n3);
    }
    mg->scal.alloc(16); mg->scratch.alloc(2048);
    mg->scal.zero(nullptr);
    VFEM_HIP(hipDeviceSynchronize());
}

static int mg_create_common(vfem_mg **out, vfem_sim *fine, int L, int first_active) {
    VFEM_TRY
    if (L < 0) throw Error("numCoarseningLevels must be >= 0");
    if (first_active < 0 || first_active > L) throw Error("first active level out of range");
    if (fine->ex_lo || fine->ex_hi) throw Error("simulators with element padding need vfem_mg_create_slab");
    std::unique_ptr<vfem_mg> mg(new vfem_mg);
    mg->fine = fine; mg->L = L; mg->first_active = first_active;
    mg->lv.resize((size_t) L + 1);
    long long ne[3] = {fine->d.nx, fine->d.ny, fine->d.nz};
    for (int l = 0; l <= L; ++l) {
        MgLevel &lv = mg->lv[l];
        if (l > 0) {
            for (int dd = 0; dd < 3; ++dd) {
                if (ne[dd] % 2 == 1)
                    throw Error("Grid size currently must be divisible by 2^numCoarseningLevels (nonuniform coarsening not yet implemented)");
                ne[dd] /= 2;
            }
        }
        lv.d = Dims(ne[0], ne[1], ne[2]);
        lv.da = lv.d;
        lv.fineNX = l > 0 ? mg->lv[l - 1].d.NX : 0;
        lv.kind = (l == 0) ? OP_MF0 : (l == 1 ? OP_MF1 : OP_STENCIL);
        if (l == 0) { lv.hmask = fine->hmask; lv.maskp = fine->dmask.p; }
        else {
            coarsen_dirichlet(mg->lv[l - 1].d, mg->lv[l - 1].hmask, lv.d, lv.hmask);
            lv.mask.alloc((size_t) lv.d.nn);
            VFEM_HIP(hipMemcpy(lv.mask.p, lv.hmask.data(), (size_t) lv.d.nn, hipMemcpyHostToDevice));
            lv.maskp = lv.mask.p;
        }
    }
    finish_mg_create(mg.get());
    *out = mg.release();
    VFEM_CATCH
}

int vfem_mg_create(vfem_mg **out, vfem_sim *fine, int L) { return mg_create_common(out, fine, L, 0); }
int vfem_mg_create_partial(vfem_mg **out, vfem_sim *fine, int L, int first_active_level) {
    return mg_create_common(out, fine, L, first_active_level);
}

int vfem_mg_create_slab(vfem_mg **out, vfem_sim *fine, int n_levels, const vfem_slab_level *lv_in,
                        const uint8_t *const *masks_host) {
    VFEM_TRY
    if (n_levels < 1) throw Error("need at least one level");
    std::unique_ptr<vfem_mg> mg(new vfem_mg);
    mg->fine = fine; mg->L = n_levels - 1; mg->slab = true;
    mg->lv.resize((size_t) n_levels);
    long long ny = fine->d.ny, nz = fine->d.nz;
    for (int l = 0; l < n_levels; ++l) {
        MgLevel &lv = mg->lv[l];
        if (l > 0) {
            if (ny % 2 || nz % 2) throw Error("Grid size currently must be divisible by 2^numCoarseningLevels (nonuniform coarsening not yet implemented)");
            ny /= 2; nz /= 2;
        }
        lv.d = Dims(lv_in[l].nx, ny, nz);
        lv.ex_lo = lv_in[l].elem_extra_lo; lv.ex_hi = lv_in[l].elem_extra_hi;
        lv.da = Dims(lv_in[l].nx + lv.ex_lo + lv.ex_hi, ny, nz);
        lv.xshift = (int) lv_in[l].xshift; lv.xparity = lv_in[l].xparity & 1;
        lv.fineNX = l > 0 ? mg->lv[l - 1].d.NX : 0;
        lv.kind = (l == 0) ? OP_MF0 : (l == 1 ? OP_MF1 : OP_STENCIL);
        if (l > 0 && l < n_levels - 1 && mg->lv[l - 1].da.nx != 2 * lv.da.nx)
            throw Error("slab element arrays must halve exactly between levels");   // (the last level only serves the transfers)
        lv.hmask.assign(masks_host[l], masks_host[l] + lv.d.nn);
        lv.mask.alloc((size_t) lv.d.nn);
        VFEM_HIP(hipMemcpy(lv.mask.p, lv.hmask.data(), (size_t) lv.d.nn, hipMemcpyHostToDevice));
        lv.maskp = lv.mask.p;
    }
    if (fine->d.nx != mg->lv[0].d.nx || fine->ex_lo != mg->lv[0].ex_lo || fine->ex_hi != mg->lv[0].ex_hi)
        throw Error("level 0 of the slab hierarchy does not match the simulator");
    finish_mg_create(mg.get());
    *out = mg.release();
    VFEM_CATCH
}
int vfem_mg_destroy(vfem_mg *mg) {
    VFEM_TRY
    delete mg;
    VFEM_CATCH
}
int vfem_mg_num_levels(const vfem_mg *mg) { return mg->L + 1; }
int vfem_mg_level_dims(const vfem_mg *mg, int level, int64_t ne[3]) {
    VFEM_TRY
    const Dims &d = mg->lv.at((size_t) level).d;
    ne[0] = d.nx; ne[1] = d.ny; ne[2] = d.nz;
    VFEM_CATCH
}
int64_t vfem_mg_level_num_nodes(const vfem_mg *mg, int level) {
    if (level < 0 || level > mg->L) return -1;
    return mg->lv[(size_t) level].d.nn;
}
int vfem_mg_level_dirichlet_mask(const vfem_mg *mg, int level, uint8_t *mask_host) {
    VFEM_TRY
    const MgLevel &lv = mg->lv.at((size_t) level);
    std::memcpy(mask_host, lv.hmask.data(), lv.hmask.size());
    VFEM_CATCH
}
int vfem_mg_set_symmetric_gauss_seidel(vfem_mg *mg, int symmetric) { mg->symmetric_gs = symmetric != 0; return 0; }
const double *vfem_mg_field_ptr(const vfem_mg *mg, int which, int level) {
    if (which == 2) return mg->lv[0].b.p;          // the PCG residual lives in the level-0 right-hand side
    if (level < 0 || level > mg->L) return nullptr;
    return which == 0 ? mg->lv[(size_t) level].x.p : mg->lv[(size_t) level].b.p;
}

static void check_level(const vfem_mg *mg, int level) {
    if (level < 0 || level > mg->L) throw Error("level out of range");
}

int vfem_mg_update_operators(vfem_mg *mg, void *stream) { VFEM_TRY update_operators(mg, S(stream)); VFEM_CATCH }

int vfem_mg_export_level_ke(vfem_mg *mg, int level, int64_t child_first_layer, int64_t count_x, double *ke_out, void *stream) {
    VFEM_TRY
    check_level(mg, level);
    if (level < 2) throw Error("element matrices exist from level 2 on (level 1 is virtual)");
    if (count_x < 0 || child_first_layer < 0) throw Error("negative layer range");
    update_operators(mg, S(stream));
    const MgLevel &lv = mg->lv[(size_t) level];
    const Dims c(count_x, lv.d.ny, lv.d.nz);
    if (level == 2) {         // straight from the fine moduli (64 per element): layers child_first_layer .. of the simulator's array
        const vfem_sim *sim = mg->fine;
        if (child_first_layer + 4 * count_x > sim->d.nx + sim->ex_lo + sim->ex_hi) throw Error("layer range outside the fine element array");
        launch_coarsen_ke(c, 3, mg->c2K0.p, sim->E.p + child_first_layer * (long long) sim->d.ny * sim->d.nz, nullptr, ke_out, S(stream));
    } else {
        const MgLevel &ch = mg->lv[(size_t) level - 1];
        if (!ch.Ke.p) throw Error("the child level holds no element matrices");
        if (child_first_layer + 2 * count_x > ch.da.nx) throw Error("layer range outside the child level's element array");
        launch_coarsen_ke(c, 2, nullptr, nullptr, ch.Ke.p + child_first_layer * (long long) ch.da.ny * ch.da.nz * 576, ke_out, S(stream));
    }
    VFEM_CATCH
}
int vfem_mg_import_level_ke(vfem_mg *mg, int level, const double *ke, void *stream) {
    VFEM_TRY
    check_level(mg, level);
    if (level < 2 || level != mg->first_active) throw Error("element matrices can be supplied for the first active level (>= 2) of a partial hierarchy");
    MgLevel &lv = mg->lv[(size_t) level];
    lv.Ke.alloc((size_t) lv.da.ne * 576);
    VFEM_HIP(hipMemcpyAsync(lv.Ke.p, ke, (size_t) lv.da.ne * 576 * sizeof(double), hipMemcpyDeviceToDevice, S(stream)));
    mg->external_ke_level = level;
    mg->operators_valid = false;              // rebuilt from these matrices at the next update
    VFEM_CATCH
}

int vfem_mg_apply_k(vfem_mg *mg, int level, const double *u, double *out, void *stream) {
    VFEM_TRY
    check_level(mg, level);
    if (level >= 1) update_operators(mg, S(stream));          // no-op when the operators match the current moduli
    mg_apply(mg, level, u, nullptr, 0, out, S(stream));
    VFEM_CATCH
}
int vfem_mg_residual(vfem_mg *mg, int level, const double *u, const double *b, double *r, void *stream) {
    VFEM_TRY
    check_level(mg, level);
    if (level >= 1) update_operators(mg, S(stream));          // no-op when the operators match the current moduli
    mg_apply(mg, level, u, b, 1, r, S(stream));
    VFEM_CATCH
}
int vfem_mg_smooth(vfem_mg *mg, int level, double *u, const double *b, int forward, void *stream) {
    VFEM_TRY
    check_level(mg, level);
    if (level >= 1) update_operators(mg, S(stream));          // no-op when the operators match the current moduli
    mg_smooth_n(mg, level, u, b, forward, 1, S(stream));
    VFEM_CATCH
}
int vfem_mg_smooth_sweeps(vfem_mg *mg, int level, double *u, const double *b, int forward, int sweeps, void *stream) {
    VFEM_TRY
    check_level(mg, level);
    if (sweeps < 0) throw Error("negative sweep count");
    if (level >= 1) update_operators(mg, S(stream));
    mg_smooth_n(mg, level, u, b, forward, sweeps, S(stream));
    VFEM_CATCH
}
int vfem_mg_zero_dirichlet(vfem_mg *mg, int level, double *u, void *stream) {
    VFEM_TRY
    check_level(mg, level);
    launch_zero_dirichlet(mg->lv[(size_t) level].d.nn, mg->lv[(size_t) level].maskp, u, S(stream));
    VFEM_CATCH
}
int vfem_mg_restrict(vfem_mg *mg, int fine_level, const double *fine, double *coarse, void *stream) {
    VFEM_TRY
    check_level(mg, fine_level + 1);
    launch_restrict(mg->lv[(size_t) fine_level + 1].d, mg->lv[(size_t) fine_level].d.NX, mg->lv[(size_t) fine_level + 1].xshift, fine, coarse, S(stream));
    VFEM_CATCH
}
int vfem_mg_interpolate(vfem_mg *mg, int fine_level, const double *coarse, double *fine, int accumulate, void *stream) {
    VFEM_TRY
    check_level(mg, fine_level + 1);
    launch_prolong(mg->lv[(size_t) fine_level + 1].d, mg->lv[(size_t) fine_level].d.NX, mg->lv[(size_t) fine_level + 1].xshift, coarse, fine, accumulate, S(stream));
    VFEM_CATCH
}
int vfem_dense_spd_inverse(int64_t n, double *A, void *stream) {
    VFEM_TRY
    if (n < 1 || n > 40000) throw Error("dense inverse: n must be in [1, 40000]");      // (n = 40 000: 12.8 GB + three work matrices of the padded size = 51 GB)
    DenseWork w;
    dense_spd_inverse(n, A, w, S(stream));
    VFEM_HIP(hipStreamSynchronize(S(stream)));      // the workspace is released on return
    VFEM_CATCH
}
int vfem_mg_coarsest_solve(vfem_mg *mg, const double *b, double *x, void *stream) {
    VFEM_TRY
    update_operators(mg, S(stream));
    coarsest_solve(mg, b, x, S(stream));
    VFEM_CATCH
}

int vfem_mg_smooth_colors(vfem_mg *mg, int level, double *u, const double *b, int forward, int first, int count, void *stream) {
    VFEM_TRY
    check_level(mg, level);
    if (first < 0 || count < 0 || first + count > 8) throw Error("colour range out of [0, 8)");
    if (level >= 1) update_operators(mg, S(stream));          // no-op when the operators match the current moduli
    if (first % 4 == 0 && count == 4 && mg_smooth_half(mg, level, u, b, forward, first / 4, S(stream))) return 0;
    mg_smooth(mg, level, u, b, forward, S(stream), first, count);
    VFEM_CATCH
}
/* one colour group (colours [4 group, 4 group + 4) of the sweep order) restricted to the node planes [plane_lo, plane_hi] of the
 * level's local grid: what a slab rank needs to relax its interface planes first, start the halo exchange, and relax the
 * interior meanwhile.  Only the marching finest-level sweep can do this (out of place, plane by plane); the return value of
 * vfem_mg_can_smooth_planes says whether this level of this hierarchy does. */
int vfem_mg_can_smooth_planes(vfem_mg *mg, int level) {
    if (!mg || level != 0 || level > mg->L) return 0;
    const vfem_sim *sim = mg->fine;
    const MgLevel &L = mg->lv[0];
    return (L.kind == OP_MF0 && gs_march_wanted(L, sim->tune) && sim->tune.gs_variant == 0 && sim->tune.gs_resident && sim->dGsTab.p && sim->k0_mirror_ok) ? 1 : 0;
}
int vfem_mg_smooth_group_planes(vfem_mg *mg, int level, double *u, const double *b, int forward, int group, int64_t plane_lo, int64_t plane_hi,
                                void *stream) {
    VFEM_TRY
    check_level(mg, level);
    if (group < 0 || group > 1) throw Error("colour group must be 0 or 1");
    if (plane_lo < 0 || plane_hi > mg->lv[(size_t) level].d.NX - 1) throw Error("plane range outside the level's node grid");
    if (plane_lo > plane_hi) return 0;
    if (!mg_smooth_half(mg, level, u, b, forward, group, S(stream), (int) plane_lo, (int) plane_hi))
        throw Error("plane-range sweeps need the marching finest-level kernel (vfem_mg_can_smooth_planes)");
    VFEM_CATCH
}
int vfem_mg_cycle_from_level(vfem_mg *mg, int level, double *x, const double *b, int nsmooth, int fmg, void *stream) {
    VFEM_TRY
    check_level(mg, level);
    if (level < mg->first_active) throw Error("level below the first active level of this hierarchy");
    if (mg->slab) throw Error("slab hierarchies are cycled by the distributed driver");
    hipStream_t s = S(stream);
    update_operators(mg, s);
    MgLevel &L = mg->lv[(size_t) level];
    const size_t bytes = (size_t) L.d.nn * 3 * sizeof(double);
    VFEM_HIP(hipMemcpyAsync(L.b.p, b, bytes, hipMemcpyDeviceToDevice, s));
    if (fmg) full_multigrid(mg, level, nsmooth, true, s);
    else {
        VFEM_HIP(hipMemcpyAsync(L.x.p, x, bytes, hipMemcpyDeviceToDevice, s));
        vcycle(mg, level, nsmooth, true, s);
    }
    VFEM_HIP(hipMemcpyAsync(x, L.x.p, bytes, hipMemcpyDeviceToDevice, s));
    VFEM_CATCH
}

int vfem_mg_solve(vfem_mg *mg, double *x, const double *f, int num_steps, int nsmooth, int stiffness_updated,
                  int zero_dirichlet, int fmg, void *stream) {
    VFEM_TRY
    ScopedTimer tm("MG Solver");
    hipStream_t s = S(stream);
    (void) stiffness_updated;                       // the simulator tracks changes of the moduli itself
    update_operators(mg, s);
    if (num_steps == 0) return 0;
    const size_t bytes = (size_t) mg->fine->d.nn * 3 * sizeof(double);
    VFEM_HIP(hipMemcpyAsync(mg->lv[0].x.p, x, bytes, hipMemcpyDeviceToDevice, s));
    VFEM_HIP(hipMemcpyAsync(mg->lv[0].b.p, f, bytes, hipMemcpyDeviceToDevice, s));
    mg_cycles(mg, num_steps, nsmooth, zero_dirichlet != 0, fmg != 0, s);
    VFEM_HIP(hipMemcpyAsync(x, mg->lv[0].x.p, bytes, hipMemcpyDeviceToDevice, s));
    VFEM_CATCH
}

int vfem_mg_pcg(vfem_mg *mg, double *x, const double *b, int max_iter, double tol, int mg_iterations,
                int mg_smoothing, int fmg, vfem_residual_cb residual_cb, void *cb_user, int *iters_out,
                double *relres_out, void *stream) {
    VFEM_TRY
    hipStream_t s = S(stream);
    vfem_sim *sim = mg->fine;
    if (mg->slab || mg->first_active != 0) throw Error("this hierarchy is driven by the distributed solver");
    const long long nn = sim->d.nn, n3 = 3 * nn;
    const size_t bytes = (size_t) n3 * sizeof(double);
    // the residual lives in the level-0 right-hand-side buffer of the hierarchy and the preconditioned residual is read from
    // its level-0 iterate: the cycle never writes b[0], so neither vector has to be copied in or out (2 x 3.2 GB per iteration
    // at 512^3)
    double *r = mg->lv[0].b.p, *d = mg->pd.p, *Ad = mg->pAd.p, *sc = mg->scal.p;
    double *sv = mg_smoothing == 0 ? mg->ps.p : mg->lv[0].x.p;
    const uint8_t *mask = mg->lv[0].maskp;

    launch_enforce_dirichlet(nn, mask, sim->dvals.p, x, 0, s);          // MG.hh:687-688
    update_operators(mg, s);                                            // MG.hh:690-691
    ScopedTimer tm("CG Iterations");
    double host_sc[4];
    launch_dot(n3, b, b, mg->scratch.p, sc + 4, s);                     // ||b||^2
    mg_apply(mg, 0, x, b, 1, r, s);                                     // r = b - K x, Dirichlet zeroed (MG.hh:696)
    launch_dot(n3, r, r, mg->scratch.p, sc + 3, s);
    VFEM_HIP(hipMemcpyAsync(host_sc, sc + 3, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    VFEM_HIP(hipStreamSynchronize(s));
    double rr = host_sc[0];
    const double bb = host_sc[1];
    int it = 0;
    while (it < max_iter && rr > tol * tol * bb) {                      // MG.hh:711 (counter started at 0)
        ++it;
        // s = M^{-1} r  (applyPreconditionerInv, MG.hh:476-479)
        if (mg_smoothing == 0) {
            VFEM_HIP(hipMemcpyAsync(sv, r, bytes, hipMemcpyDeviceToDevice, s));
        } else {
            if (!fmg) mg->lv[0].x.zero(s);          // full multigrid overwrites the iterate of every level (prolongation, MG.hh:500)
            mg_cycles(mg, mg_iterations, mg_smoothing, true, fmg != 0, s);
        }
        // the vector work between the cycle and the apply, three passes fewer than one kernel per line of MG.hh:713-725 (same
        // sums in the same order: iterates and residuals are unchanged bit for bit)
        launch_shift_scalar(sc, s);                                     // rMr_old = rMr
        launch_dot_zero_dirichlet(n3, r, sv, mask, mg->scratch.p, sc + 0, s);   // s = zeroDirichlet(s); rMr = r . s
        launch_pcg_direction(n3, sv, d, sc, it == 1, s);
        mg_apply(mg, 0, d, nullptr, 0, Ad, s);                          // Ad = K d
        launch_dot_zero_dirichlet(n3, d, Ad, mask, mg->scratch.p, sc + 2, s);   // Ad = zeroDirichlet(Ad); d . Ad
        launch_pcg_step_dot(n3, x, r, d, Ad, sc, mg->scratch.p, sc + 3, s);      // x += alpha d, r -= alpha Ad, ||r||^2
        VFEM_HIP(hipMemcpyAsync(host_sc, sc + 3, sizeof(double), hipMemcpyDeviceToHost, s));
        VFEM_HIP(hipStreamSynchronize(s));
        rr = host_sc[0];
        if (!(rr == rr)) throw Error("PCG produced NaN residual");
        if (residual_cb) residual_cb(cb_user, it, std::sqrt(rr));
    }
    if (iters_out) *iters_out = it;
    if (relres_out) *relres_out = bb > 0 ? std::sqrt(rr / bb) : 0.0;
    VFEM_CATCH
}

// ---- slab-decomposed MG-PCG driven from here (round 4) ----------------------------------------------------------------------
// ndr_amd/distributed.py drove the distributed cycle from Python: ~600 ctypes calls per PCG iteration, 5-6.6 ms of host time per
// iteration measured by the rank proxy (profiles/r04_rank_proxy_python_driver.json) against the 2.0 / 12 ms a rank has at 256^3 /
// 512^3 on eight ranks.  Here a rank's whole solve is ONE call; the two things only the host language can do -- refresh ghost planes
// from the neighbours, sum a few doubles over the ranks (torch.distributed: RCCL on the GPU box, gloo in the tests) -- are callbacks.
// Control flow = DistributedMGSolver's (vcycle / full_multigrid / smooth with the parity-aware, boundary-planes-first exchanges),
// itself MG.hh:486-553, 679-732; all work vectors belong to the caller, so a callback can map a pointer back to its own array.
}  // extern "C"
namespace {
struct DistDriver {
    vfem_mg *loc, *rep;
    int T, rank, world, nsmooth;
    bool overlap;
    const vfem_dist_level *g;
    double *xT, *bT;
    vfem_halo_fn halo_fn;
    vfem_allreduce_fn allreduce_fn;
    void *user;
    hipStream_t s;

    void halo(int l, double *f, bool left = true, bool right = true, int phase = 0) {
        if (world == 1 || !(g[l].gl || g[l].gr)) return;
        if (halo_fn(user, l, f, left ? 1 : 0, right ? 1 : 0, phase) != 0) throw Error("halo exchange callback failed");
    }
    void allreduce(double *buf, long long n) {
        if (world > 1 && allreduce_fn(user, buf, (int64_t) n) != 0) throw Error("all-reduce callback failed");
    }
    // sum over the node planes this rank counts (interface planes belong to the lower rank), then over the ranks
    void dot(const double *a, const double *b, double *out) {
        const vfem_dist_level &G = g[0];
        const long long lo = G.first_owned, hi = G.last_owned + (rank == world - 1 ? 1 : 0), per = 3 * G.plane_nodes;
        launch_dot((hi - lo) * per, a + lo * per, b + lo * per, loc->scratch.p, out, s);
        allreduce(out, 1);
    }
    void smooth_colors(int l, double *x, const double *b, int forward, int first) {
        if (!mg_smooth_half(loc, l, x, b, forward, first / 4, s)) mg_smooth(loc, l, x, b, forward, s, first, 4);
    }
    // one sweep of a distributed level (DistributedMGSolver.smooth): a colour group changes the planes of one global x parity, so the
    // neighbours' ghost planes go stale only if the planes they mirror have it; where the level is swept plane by plane, the planes a
    // neighbour waits for are relaxed first and travel while the interior is relaxed
    void smooth(int l, double *x, const double *b, int forward) {
        const vfem_dist_level &G = g[l];
        const bool by_planes = overlap && vfem_mg_can_smooth_planes(loc, l);
        for (int group = 0; group < 2; ++group) {
            const int cx = forward ? group : 1 - group;
            const bool send_left = G.gl && ((G.xoffn + G.first_owned + 1) & 1) == cx;
            const bool send_right = G.gr && ((G.xoffn + G.last_owned - 1) & 1) == cx;
            if (!(send_left || send_right) || world == 1) { smooth_colors(l, x, b, forward, 4 * group); continue; }
            if (!by_planes) {
                smooth_colors(l, x, b, forward, 4 * group);
                halo(l, x, send_left, send_right, 0);
                continue;
            }
            auto sweep = [&](long long lo, long long hi) {
                if (lo > hi) return;
                if (!mg_smooth_half(loc, l, x, b, forward, group, s, (int) lo, (int) hi)) throw Error("plane-range sweep unavailable");
            };
            const long long lo_plane = G.first_owned + 1, hi_plane = G.last_owned - 1;
            long long inner_lo = G.first_owned, inner_hi = G.last_owned;
            if (send_left) { sweep(lo_plane, lo_plane); inner_lo = lo_plane + 1; }
            if (send_right && !(send_left && hi_plane == lo_plane)) { sweep(hi_plane, hi_plane); inner_hi = hi_plane - 1; }
            else if (send_right) inner_hi = hi_plane - 1;
            halo(l, x, send_left, send_right, 1);
            sweep(inner_lo, inner_hi);
            halo(l, x, send_left, send_right, 2);
        }
    }
    // the replicated coarse hierarchy: right-hand side = sum of the ranks' disjoint planes, every rank runs the same cycle and keeps its slab
    void coarse_cycle(bool fmg) {
        const vfem_dist_level &G = g[T];
        MgLevel &R = rep->lv[(size_t) T];
        const long long lo = G.first_owned, hi = G.last_owned + (rank == world - 1 ? 1 : 0), per = 3 * G.plane_nodes;
        const size_t bytes = (size_t) R.d.nn * 3 * sizeof(double);
        VFEM_HIP(hipMemsetAsync(bT, 0, bytes, s));
        VFEM_HIP(hipMemcpyAsync(bT + (G.xoffn + lo) * per, G.b + lo * per, (size_t) ((hi - lo) * per) * sizeof(double), hipMemcpyDeviceToDevice, s));
        allreduce(bT, (long long) R.d.nn * 3);
        VFEM_HIP(hipMemcpyAsync(R.b.p, bT, bytes, hipMemcpyDeviceToDevice, s));
        if (fmg) full_multigrid(rep, T, nsmooth, true, s);
        else { R.x.zero(s); vcycle(rep, T, nsmooth, true, s); }
        VFEM_HIP(hipMemcpyAsync(G.x, R.x.p + G.xoffn * per, (size_t) (G.n_planes * per) * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    void vcycle_d(int l) {
        if (l == T) { coarse_cycle(false); return; }
        const vfem_dist_level &G = g[l], &C = g[l + 1];
        MgLevel &L = loc->lv[(size_t) l], &LC = loc->lv[(size_t) l + 1];
        launch_zero_dirichlet(L.d.nn, L.maskp, G.x, s);                  // residual system
        for (int i = 0; i < nsmooth; ++i) smooth(l, G.x, G.b, 1);
        mg_apply(loc, l, G.x, G.b, 1, G.r, s);
        halo(l, G.r);
        launch_restrict(LC.d, L.d.NX, LC.xshift, G.r, C.b, s, C.x);      // ... and the zero initial guess of the coarse level
        vcycle_d(l + 1);
        launch_prolong(LC.d, L.d.NX, LC.xshift, C.x, G.x, 1, s);
        halo(l, G.x);
        for (int i = 0; i < nsmooth; ++i) smooth(l, G.x, G.b, loc->symmetric_gs ? 0 : 1);
    }
    void fmg_d(int l) {
        if (l == T) { coarse_cycle(true); return; }
        const vfem_dist_level &G = g[l], &C = g[l + 1];
        MgLevel &L = loc->lv[(size_t) l], &LC = loc->lv[(size_t) l + 1];
        halo(l, G.b);
        launch_restrict(LC.d, L.d.NX, LC.xshift, G.b, C.b, s);
        fmg_d(l + 1);
        launch_prolong(LC.d, L.d.NX, LC.xshift, C.x, G.x, 0, s);
        halo(l, G.x);
        vcycle_d(l);
    }
};
}  // namespace
extern "C" {
int vfem_mg_pcg_slab(vfem_mg *local, vfem_mg *replicated, int first_replicated_level, const vfem_dist_level *levels, int rank, int world,
                     double *replicated_x, double *replicated_b, double *x, const double *b, double *work_d, double *work_Ad, double *scalars,
                     int max_iter, double tol, int mg_iterations, int mg_smoothing, int fmg, int overlap_sweeps,
                     vfem_halo_fn halo, vfem_allreduce_fn allreduce, void *cb_user, vfem_residual_cb residual_cb, void *residual_user,
                     int *iters_out, double *relres_out, void *stream) {
    VFEM_TRY
    if (!local || !local->slab) throw Error("vfem_mg_pcg_slab: the local hierarchy must come from vfem_mg_create_slab");
    const int T = first_replicated_level;
    if (T < 1 || T != local->L) throw Error("vfem_mg_pcg_slab: the local hierarchy must end at the first replicated level");
    if (!replicated || replicated->slab || T > replicated->L || T < replicated->first_active) throw Error("vfem_mg_pcg_slab: level not active in the replicated hierarchy");
    if (world > 1 && (!halo || !allreduce)) throw Error("vfem_mg_pcg_slab: callbacks missing");
    for (int l = 0; l <= T; ++l) {
        const MgLevel &L = local->lv[(size_t) l];
        if (levels[l].n_planes != L.d.NX || levels[l].plane_nodes != (int64_t) L.d.NY * L.d.NZ) throw Error("vfem_mg_pcg_slab: level geometry does not match the hierarchy");
        if (!levels[l].x || !levels[l].b || (l < T && !levels[l].r)) throw Error("vfem_mg_pcg_slab: work vector missing");
    }
    DistDriver D{local, replicated, T, rank, world, mg_smoothing, overlap_sweeps != 0, levels, replicated_x, replicated_b, halo, allreduce, cb_user, S(stream)};
    hipStream_t s = D.s;
    const vfem_dist_level &G0 = levels[0];
    const MgLevel &L0 = local->lv[0];
    const long long nn = L0.d.nn, n3 = 3 * nn;
    const size_t bytes = (size_t) n3 * sizeof(double);
    // as in vfem_mg_pcg the residual lives in the level-0 right-hand side of the cycle and the preconditioned residual is its iterate
    double *r = G0.b, *sv = G0.x, *d = work_d, *Ad = work_Ad, *sc = scalars;
    launch_zero_dirichlet(nn, L0.maskp, x, s);                           // (zero Dirichlet values only: DistributedMGSolver.pcg)
    update_operators(local, s);                                          // no-ops when the caller has done it (sharded densities: it must)
    update_operators(replicated, s);
    double host_sc[2];
    D.dot(b, b, sc + 4);
    D.halo(0, x);
    mg_apply(local, 0, x, b, 1, r, s);
    D.dot(r, r, sc + 3);
    VFEM_HIP(hipMemcpyAsync(host_sc, sc + 3, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    VFEM_HIP(hipStreamSynchronize(s));
    double rr = host_sc[0];
    const double bb = host_sc[1];
    int it = 0;
    while (it < max_iter && rr > tol * tol * bb) {
        ++it;
        if (mg_smoothing == 0) {
            VFEM_HIP(hipMemcpyAsync(sv, r, bytes, hipMemcpyDeviceToDevice, s));
        } else if (fmg) {
            D.fmg_d(0);
            for (int i = 1; i < mg_iterations; ++i) D.vcycle_d(0);
        } else {
            VFEM_HIP(hipMemsetAsync(sv, 0, bytes, s));
            for (int i = 0; i < mg_iterations; ++i) D.vcycle_d(0);
        }
        launch_zero_dirichlet(nn, L0.maskp, sv, s);
        launch_shift_scalar(sc, s);                                     // rMr_old = rMr
        D.dot(r, sv, sc + 0);
        launch_pcg_direction(n3, sv, d, sc, it == 1, s);
        D.halo(0, d);
        mg_apply(local, 0, d, nullptr, 0, Ad, s);
        launch_zero_dirichlet(nn, L0.maskp, Ad, s);
        D.dot(d, Ad, sc + 2);
        launch_pcg_step(n3, x, r, d, Ad, sc, s);                         // x += alpha d, r -= alpha Ad
        D.dot(r, r, sc + 3);
        VFEM_HIP(hipMemcpyAsync(host_sc, sc + 3, sizeof(double), hipMemcpyDeviceToHost, s));
        VFEM_HIP(hipStreamSynchronize(s));
        rr = host_sc[0];
        if (!(rr == rr)) throw Error("PCG produced NaN residual");
        if (residual_cb) residual_cb(residual_user, it, std::sqrt(rr));
    }
    if (iters_out) *iters_out = it;
    if (relres_out) *relres_out = bb > 0 ? std::sqrt(rr / bb) : 0.0;
    VFEM_CATCH
}

// ---- design-update path ----
int vfem_box_filter(const int64_t n[3], int radius, const double *in, double *out, int transpose, void *stream) {
    VFEM_TRY
    if (radius < 0) throw Error("negative filter radius");
    launch_box_filter((int) n[0], (int) n[1], (int) n[2], radius, in, out, transpose, S(stream));
    VFEM_CATCH
}
int vfem_projection(int64_t n, double beta, const double *x, double *out, void *stream) {
    VFEM_TRY
    if (!(beta > 0)) throw Error("Beta parameter has to be positive (received beta = " + std::to_string(beta) + ")");
    launch_projection(n, beta, x, nullptr, out, 0, S(stream));
    VFEM_CATCH
}
int vfem_projection_backprop(int64_t n, double beta, const double *g, const double *vars, double *out, void *stream) {
    VFEM_TRY launch_projection(n, beta, vars, g, out, 1, S(stream)); VFEM_CATCH
}
int vfem_oc_candidate(int64_t n, const double *x0, const double *dJ, const double *dc, double lambda, double move, double *out,
                      void *stream) {
    VFEM_TRY launch_oc_candidate(n, x0, dJ, dc, lambda, move, out, S(stream)); VFEM_CATCH
}
int vfem_mean(int64_t n, const double *x, double *mean_host, void *stream) {
    VFEM_TRY
    DevBuf<double> tmp; tmp.alloc(2048 + 8);
    launch_sum(n, x, tmp.p + 8, tmp.p, S(stream));
    double v = 0.0;
    VFEM_HIP(hipMemcpyAsync(&v, tmp.p, sizeof(double), hipMemcpyDeviceToHost, S(stream)));
    VFEM_HIP(hipStreamSynchronize(S(stream)));
    *mean_host = n > 0 ? v / (double) n : 0.0;
    VFEM_CATCH
}

// ---- MLP ----
int vfem_mlp_create(vfem_mlp **out, int es, int nn, int n_layers, int sigmoid) {
    VFEM_TRY
    if (es <= 0 || es % 32 != 0) throw Error("embedding_size must be a positive multiple of 32");
    if (nn <= 0 || nn % 32 != 0 || nn > 512) throw Error("n_neurons must be a multiple of 32, at most 512");
    if (n_layers < 2) throw Error("n_layers must be at least 2");
    std::unique_ptr<vfem_mlp> m(new vfem_mlp);
    m->es = es; m->nn = nn; m->n_layers = n_layers; m->sigmoid = sigmoid;
    *out = m.release();
    VFEM_CATCH
}
int vfem_mlp_destroy(vfem_mlp *mlp) {
    VFEM_TRY
    delete mlp;
    VFEM_CATCH
}
int vfem_mlp_set_option(vfem_mlp *m, int key, int value) {
    VFEM_TRY
    if (key == VFEM_MLP_OPT_BWD_TERMS) {
        if (value != 1 && value != 3) throw Error("VFEM_MLP_OPT_BWD_TERMS: 3 (hi hi + hi lo + lo hi, reference precision) or 1 (hi hi)");
        m->bwd_terms = value;
    } else if (key == VFEM_MLP_OPT_KEEP_FIRST) {
        m->keep_first = value != 0;
        if (!m->keep_first) { m->h0_valid = false; m->h0_hi.release(); m->h0_lo.release(); }
    } else throw Error("unknown MLP option");
    VFEM_CATCH
}
int vfem_mlp_load_weights(vfem_mlp *m, const float *B, const float *W1, const float *Wh, const float *biases,
                          const float *wout, float bout) {
    VFEM_TRY
    const int nh = m->n_layers - 2;
    auto up = [](DevBuf<float> &d, const float *h, size_t n) {
        d.alloc(n);
        if (n) VFEM_HIP(hipMemcpy(d.p, h, n * sizeof(float), hipMemcpyDefault));
    };
    // fp32 copies first (the reference-precision forward uses them); the fp16 operands and the transposed hidden weights are
    // converted from those on the device: no temporary allocations, no device-wide synchronisation per training step
    up(m->B, B, (size_t) m->es * 3);
    up(m->W1f, W1, (size_t) m->nn * 2 * m->es);
    up(m->Whf, Wh, (size_t) nh * m->nn * m->nn);
    m->W1.alloc((size_t) m->nn * 2 * m->es);
    launch_f32_to_f16_frag(m->nn, 2 * m->es, 0, m->W1f.p, m->W1.p, nullptr);
    m->W1h.alloc((size_t) m->nn * 2 * m->es);
    m->W1l.alloc((size_t) m->nn * 2 * m->es);
    m->kc = m->es % 64 == 0 ? 128 : 64;
    launch_split_f32_frag(m->nn, 2 * m->es, m->W1f.p, m->W1h.p, m->W1l.p, nullptr, m->es, 0, m->kc);
    m->Whh.alloc((size_t) nh * m->nn * m->nn);
    m->Whl.alloc((size_t) nh * m->nn * m->nn);
    for (int l = 0; l < nh; ++l)
        launch_split_f32_frag(m->nn, m->nn, m->Whf.p + (size_t) l * m->nn * m->nn, m->Whh.p + (size_t) l * m->nn * m->nn, m->Whl.p + (size_t) l * m->nn * m->nn, nullptr, 0);
    m->Wh.alloc((size_t) nh * m->nn * m->nn);
    m->WhTh.alloc((size_t) nh * m->nn * m->nn);
    m->WhTl.alloc((size_t) nh * m->nn * m->nn);
    if (nh) {
        for (int l = 0; l < nh; ++l) {
            const size_t o = (size_t) l * m->nn * m->nn;
            launch_f32_to_f16_frag(m->nn, m->nn, 0, m->Whf.p + o, m->Wh.p + o, nullptr);
            launch_split_f32_frag(m->nn, m->nn, m->Whf.p + o, m->WhTh.p + o, m->WhTl.p + o, nullptr, 0, 1);
        }
    }
    up(m->bias, biases, (size_t) (nh + 1) * m->nn);
    up(m->wout, wout, (size_t) m->nn);
    // the split operands carry fp16(w) as their high half: a weight of 65 504 or more would become inf (the fp32 reference has no
    // such limit; networks of this kind have |w| < 10)
    m->range_flag.alloc(1);
    m->range_flag.zero(nullptr);
    launch_range_check_f32((long long) m->nn * 2 * m->es, m->W1f.p, 65504.f, m->range_flag.p, nullptr);
    if (nh) launch_range_check_f32((long long) nh * m->nn * m->nn, m->Whf.p, 65504.f, m->range_flag.p, nullptr);
    {
        int bad = 0;
        VFEM_HIP(hipMemcpy(&bad, m->range_flag.p, sizeof(int), hipMemcpyDeviceToHost));
        if (bad) { m->loaded = false; throw Error("MLP weight of magnitude >= 65504 (or not finite): outside the range of the split fp16 operands"); }
    }
    VFEM_HIP(hipStreamSynchronize(nullptr));      // the conversions ran on the null stream; consumers may launch on any stream
    m->bout = bout;
    m->h0_valid = false;
    m->loaded = true;
    VFEM_CATCH
}
}  // extern "C" (reopened below)
#include "mlp_args.h"
// a hidden activation left fp16's range in an earlier reference-precision launch: its high half was inf, the results of that launch
// are not the network's.  Reported by the next entry point (the check costs one 4-byte read-back; launches stay asynchronous)
static void mlp_check_range(vfem_mlp *m, hipStream_t s) {
    if (!m->range_flag.p) return;
    int bad = 0;
    VFEM_HIP(hipMemcpyAsync(&bad, m->range_flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    VFEM_HIP(hipStreamSynchronize(s));
    if (bad) {
        m->range_flag.zero(s);
        throw Error("MLP activation outside fp16's range (>= 65000 or not finite) in the previous reference-precision evaluation: "
                    "its results are invalid; rescale the network or use torch for it");
    }
}
static vfem::MlpArgs mlp_base_args(const vfem_mlp *m) {
    vfem::MlpArgs a{};
    a.range_flag = m->range_flag.p;
    a.ablate = ablate_mlp();
    a.es = m->es; a.nn = m->nn; a.n_hidden = m->n_layers - 2; a.sigmoid = m->sigmoid;
    a.B = m->B.p; a.W1 = m->W1.p; a.Wh = m->Wh.p; a.bias = m->bias.p; a.wout = m->wout.p; a.bout = m->bout;
    return a;
}
extern "C" {
int vfem_mlp_forward(vfem_mlp *m, const float *coords, int64_t nvox, float *o32, double *o64, void *stream) {
    VFEM_TRY
    if (!m->loaded) throw Error("vfem_mlp_load_weights has not been called");
    MlpArgs a = mlp_base_args(m);
    a.coords = coords; a.nvox = nvox; a.out32 = o32; a.out64 = o64;
    launch_mlp_forward(a, S(stream));
    VFEM_CATCH
}
static void mlp_grid_args(MlpArgs &a, const int64_t n[3], const double lo[3], const double hi[3]) {
    a.coords = nullptr;
    a.nvox = 1;
    for (int dd = 0; dd < 3; ++dd) {
        a.gn[dd] = (int) n[dd];
        a.glo[dd] = (float) lo[dd];
        a.gstep[dd] = n[dd] > 1 ? (float) ((hi[dd] - lo[dd]) / (double) (n[dd] - 1)) : 0.f;
        a.nvox *= n[dd];
    }
}
int vfem_mlp_forward_grid(vfem_mlp *m, const int64_t n[3], const double lo[3], const double hi[3], float *o32, double *o64,
                          void *stream) {
    VFEM_TRY
    if (!m->loaded) throw Error("vfem_mlp_load_weights has not been called");
    MlpArgs a = mlp_base_args(m);
    mlp_grid_args(a, n, lo, hi);
    a.out32 = o32; a.out64 = o64;
    launch_mlp_forward(a, S(stream));
    VFEM_CATCH
}
int vfem_mlp_forward_grid_range(vfem_mlp *m, const int64_t n[3], const double lo[3], const double hi[3], int64_t first_voxel,
                                int64_t num_voxels, float *o32, double *o64, void *stream) {
    VFEM_TRY
    if (!m->loaded) throw Error("vfem_mlp_load_weights has not been called");
    MlpArgs a = mlp_base_args(m);
    mlp_grid_args(a, n, lo, hi);
    if (first_voxel < 0 || num_voxels < 0 || first_voxel + num_voxels > a.nvox) throw Error("voxel range outside the grid");
    if (num_voxels == 0) return 0;
    a.v_offset = first_voxel; a.nvox = num_voxels;
    a.out32 = o32; a.out64 = o64;                      // outputs are indexed from the start of the range
    launch_mlp_forward(a, S(stream));
    VFEM_CATCH
}
}  // extern "C"
// Reference-precision forward (the reference evaluates networks.MLP in fp32 end to end, networks.py:178-185): the fused kernel with
// split fp16 operands (kernels_mlp_x3.hip) -- three MFMA products per product, fp32 accumulation, accurate fp32 sin / cos of the
// argument formed as the reference forms it.  Nothing wider than the output scalar per voxel reaches HBM.
static void mlp_forward_f32_impl(vfem_mlp *m, vfem::MlpArgs base, float *o32, double *o64, hipStream_t s,
                                 const int64_t *grid_n = nullptr, const double *grid_lo = nullptr, const double *grid_hi = nullptr) {
    if (!m->loaded) throw Error("vfem_mlp_load_weights has not been called");
    mlp_check_range(m, s);
    base.out32 = o32; base.out64 = o64;
    m->h0_valid = false;
    bool keep = m->keep_first && grid_n && base.nvox > 0 && m->n_layers > 2 && (size_t) base.nvox * m->nn * 4 <= ((size_t) 96 << 30);
    if (keep) {
        // room for the padded rows of the backward pass's last chunk (they must exist and be finite: they meet dz = 0)
        const size_t rows = (size_t) base.nvox + 4096;
        try {
            m->h0_hi.reserve(rows * m->nn);
            m->h0_lo.reserve(rows * m->nn);
        } catch (const Error &) { (void) hipGetLastError(); m->h0_hi.release(); m->h0_lo.release(); keep = false; }
    }
    if (keep) {
        VFEM_HIP(hipMemsetAsync(m->h0_hi.p + (size_t) base.nvox * m->nn, 0, (size_t) 4096 * m->nn * 2, s));
        VFEM_HIP(hipMemsetAsync(m->h0_lo.p + (size_t) base.nvox * m->nn, 0, (size_t) 4096 * m->nn * 2, s));
        base.save_act = m->h0_hi.p; base.save_act_lo = m->h0_lo.p; base.act_rows = 0; base.save_first_only = 1;
    }
    launch_mlp_forward_x3(base, m->W1h.p, m->W1l.p, m->Whh.p, m->Whl.p, s, m->kc);
    if (keep) {
        for (int dd = 0; dd < 3; ++dd) { m->h0_n[dd] = grid_n[dd]; m->h0_lo_c[dd] = grid_lo[dd]; m->h0_hi_c[dd] = grid_hi[dd]; }
        m->h0_first = base.v_offset; m->h0_count = base.nvox;
        m->h0_valid = true;
    }
}
extern "C" {
int vfem_mlp_forward_f32(vfem_mlp *m, const float *coords, int64_t nvox, float *o32, double *o64, void *stream) {
    VFEM_TRY
    MlpArgs a = mlp_base_args(m);
    a.coords = coords; a.nvox = nvox;
    mlp_forward_f32_impl(m, a, o32, o64, S(stream));
    VFEM_CATCH
}
int vfem_mlp_forward_grid_range_f32(vfem_mlp *m, const int64_t n[3], const double lo[3], const double hi[3], int64_t first_voxel,
                                    int64_t num_voxels, float *o32, double *o64, void *stream) {
    VFEM_TRY
    MlpArgs a = mlp_base_args(m);
    mlp_grid_args(a, n, lo, hi);
    if (first_voxel < 0 || num_voxels < 0 || first_voxel + num_voxels > a.nvox) throw Error("voxel range outside the grid");
    if (num_voxels == 0) return 0;
    a.v_offset = first_voxel; a.nvox = num_voxels;
    mlp_forward_f32_impl(m, a, o32, o64, S(stream), n, lo, hi);
    VFEM_CATCH
}
}  // extern "C"
// Gradients of a scalar loss wrt the MLP parameters given dL/d(out) per voxel (what torch.autograd computes for
// networks.MLP in the reference, train_xdg.py:282-329), at the reference's precision and with no library GEMM
// (kernels_mlp_bwd.hip).  Voxels are processed in chunks: reference-precision forward with saved split activations, fused backward
// data pass, the weight gradients as voxel-reduction GEMMs of our own (first layer: Fourier features regenerated in the kernel),
// column sums for the biases and the output layer.
static void mlp_backward_impl(vfem_mlp *m, vfem::MlpArgs base, const float *coords, const float *g_out, float scale,
                              float *dW1, float *dWh, float *dbias, float *dwout, float *dbout, hipStream_t s,
                              const int64_t *grid_n = nullptr, const double *grid_lo = nullptr, const double *grid_hi = nullptr) {
    if (!m->loaded) throw Error("vfem_mlp_load_weights has not been called");
    if (!(scale > 0.f)) throw Error("loss scale must be positive");
    mlp_check_range(m, s);
    const long long V = base.nvox;
    const int nn = m->nn, K1 = 2 * m->es, nh = m->n_layers - 2, nact = nh + 1;
    if (V <= 0) throw Error("empty voxel set");
    // a chunk: at most 2^20 voxels; its voxel slices (one block of the weight-gradient kernel per slice and output tile): enough
    // blocks to fill the chip -- the first layer has 16 output tiles at the run.md sizes, a hidden layer 4 -- of at least 128 voxels each
    auto plan = [](long long n_c, int &s1, int &sh, long long &rows) {
        s1 = 8; sh = 8;
        while (s1 < 32 && n_c >= (long long) 2 * s1 * 128) s1 *= 2;
        while (sh < 128 && n_c >= (long long) 2 * sh * 128) sh *= 2;
        const long long q = 32LL * std::max(s1, sh);
        rows = (n_c + q - 1) / q * q;
    };
    const long long Vc = std::min<long long>(V, 1LL << 20);
    int s1, sh; long long rows_max;
    plan(Vc, s1, sh, rows_max);
    m->acts.alloc((size_t) nact * rows_max * nn);
    m->acts_lo.alloc((size_t) nact * rows_max * nn);
    m->dz.alloc((size_t) nact * rows_max * nn);
    m->dz_lo.alloc((size_t) nact * rows_max * nn);
    m->gs.alloc((size_t) rows_max);
    m->out_chunk.alloc((size_t) rows_max);
    const size_t colblocks = (size_t) ((rows_max + 511) / 512);
    m->partial.alloc(std::max(std::max((size_t) s1 * nn * K1, (size_t) sh * nn * nn), colblocks * (size_t) nn));
    m->partial_b.alloc((size_t) 128 * nn);
    const float inv = 1.f / scale;
    // the first layer's activations as the forward pass of this step left them (VFEM_MLP_OPT_KEEP_FIRST), if they belong to this grid and range
    bool kept = m->h0_valid && grid_n && !coords && nh >= 1 && m->h0_first == base.v_offset && m->h0_count == V;
    if (kept)
        for (int dd = 0; dd < 3; ++dd) kept = kept && m->h0_n[dd] == grid_n[dd] && m->h0_lo_c[dd] == grid_lo[dd] && m->h0_hi_c[dd] == grid_hi[dd];
    for (long long c0 = 0; c0 < V; c0 += Vc) {
        const long long n_c = std::min(Vc, V - c0);
        long long rows;
        plan(n_c, s1, sh, rows);
        const float beta = c0 == 0 ? 0.f : 1.f;
        if (rows != n_c) { m->acts.zero(s); m->acts_lo.zero(s); }      // padded rows must be finite (they meet dz = 0)
        vfem::MlpArgs a = base;
        a.nvox = n_c; a.v_offset = base.v_offset + c0; a.coords = coords ? coords + 3 * c0 : nullptr;
        a.out32 = m->out_chunk.p; a.out64 = nullptr; a.save_act = m->acts.p; a.save_act_lo = m->acts_lo.p; a.act_rows = rows;
        const uint16_t *k_hi = kept ? m->h0_hi.p + (size_t) c0 * nn : nullptr, *k_lo = kept ? m->h0_lo.p + (size_t) c0 * nn : nullptr;
        a.h0_hi = k_hi; a.h0_lo = k_lo;
        launch_mlp_forward_x3(a, m->W1h.p, m->W1l.p, m->Whh.p, m->Whl.p, s, m->kc);
        a.h0_hi = nullptr; a.h0_lo = nullptr;
        vfem::MlpBwdArgs b{};
        b.nn = nn; b.n_hidden = nh; b.sigmoid = m->sigmoid; b.WhTh = m->WhTh.p; b.WhTl = m->WhTl.p; b.wout = m->wout.p; b.g = g_out + c0;
        b.out32 = m->out_chunk.p; b.scale = scale; b.act_hi = m->acts.p; b.act_lo = m->acts_lo.p; b.dz_hi = m->dz.p; b.dz_lo = m->dz_lo.p;
        b.gs = m->gs.p; b.act_rows = rows; b.nvox = n_c; b.act0_hi = k_hi; b.act0_lo = k_lo;
        launch_mlp_backward_x3(b, rows, s);
        a.save_act = nullptr; a.save_act_lo = nullptr;
        vfem::MlpDwArgs w{};
        w.nn = nn; w.rows = rows; w.terms = m->bwd_terms; w.partial = m->partial.p; w.grid = a;
        // first layer: against the Fourier features of the chunk's voxels, regenerated in the kernel
        w.K = K1; w.dz_hi = m->dz.p; w.dz_lo = m->dz_lo.p; w.h_hi = nullptr; w.h_lo = nullptr; w.slices = s1;
        w.colsum_partial = m->partial_b.p;                              // the layer's bias gradient: column sums of its dz, formed by the same kernel
        launch_mlp_dw(w, s);
        launch_reduce_partials(s1, (long long) nn * K1, m->partial.p, inv, beta, dW1, s);
        launch_reduce_partials(s1, nn, m->partial_b.p, inv, beta, dbias, s);
        for (int l = 0; l < nh; ++l) {
            w.K = nn; w.slices = sh;
            w.dz_hi = m->dz.p + (size_t) (l + 1) * rows * nn; w.dz_lo = m->dz_lo.p + (size_t) (l + 1) * rows * nn;
            w.h_hi = (l == 0 && kept) ? k_hi : m->acts.p + (size_t) l * rows * nn;
            w.h_lo = (l == 0 && kept) ? k_lo : m->acts_lo.p + (size_t) l * rows * nn;
            w.h_lo_scaled = (l == 0 && kept) ? 1 : 0;
            launch_mlp_dw(w, s);
            launch_reduce_partials(sh, (long long) nn * nn, m->partial.p, inv, beta, dWh + (size_t) l * nn * nn, s);
            launch_reduce_partials(sh, nn, m->partial_b.p, inv, beta, dbias + (size_t) (l + 1) * nn, s);
        }
        const int cb = (int) ((rows + 511) / 512);
        launch_colsum_split(rows, nn, m->acts.p + (size_t) nh * rows * nn, m->acts_lo.p + (size_t) nh * rows * nn, m->gs.p, m->partial.p, s);
        launch_reduce_partials(cb, nn, m->partial.p, inv, beta, dwout, s);
        launch_sum_f32(rows, m->gs.p, inv, beta, dbout, m->partial.p, s);
    }
    mlp_check_range(m, s);                               // (the pass's own forward)
}
extern "C" {
int vfem_mlp_backward(vfem_mlp *m, const float *coords, int64_t nvox, const float *g_out, float loss_scale, float *dW1,
                      float *dWh, float *dbias, float *dwout, float *dbout, void *stream) {
    VFEM_TRY
    MlpArgs a = mlp_base_args(m);
    a.nvox = nvox;
    mlp_backward_impl(m, a, coords, g_out, loss_scale, dW1, dWh, dbias, dwout, dbout, S(stream));
    VFEM_CATCH
}
int vfem_mlp_backward_grid(vfem_mlp *m, const int64_t n[3], const double lo[3], const double hi[3], const float *g_out,
                           float loss_scale, float *dW1, float *dWh, float *dbias, float *dwout, float *dbout, void *stream) {
    VFEM_TRY
    MlpArgs a = mlp_base_args(m);
    a.nvox = 1;
    for (int dd = 0; dd < 3; ++dd) {
        a.gn[dd] = (int) n[dd];
        a.glo[dd] = (float) lo[dd];
        a.gstep[dd] = n[dd] > 1 ? (float) ((hi[dd] - lo[dd]) / (double) (n[dd] - 1)) : 0.f;
        a.nvox *= n[dd];
    }
    mlp_backward_impl(m, a, nullptr, g_out, loss_scale, dW1, dWh, dbias, dwout, dbout, S(stream), n, lo, hi);
    VFEM_CATCH
}
int vfem_mlp_backward_grid_range(vfem_mlp *m, const int64_t n[3], const double lo[3], const double hi[3], int64_t first_voxel,
                                 int64_t num_voxels, const float *g_out, float loss_scale, float *dW1, float *dWh, float *dbias,
                                 float *dwout, float *dbout, void *stream) {
    VFEM_TRY
    MlpArgs a = mlp_base_args(m);
    mlp_grid_args(a, n, lo, hi);
    if (first_voxel < 0 || num_voxels <= 0 || first_voxel + num_voxels > a.nvox) throw Error("voxel range outside the grid");
    a.v_offset = first_voxel; a.nvox = num_voxels;
    mlp_backward_impl(m, a, nullptr, g_out, loss_scale, dW1, dWh, dbias, dwout, dbout, S(stream), n, lo, hi);
    VFEM_CATCH
}
int vfem_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float lr, float beta1,
                   float beta2, float eps, int step, void *stream) {
    VFEM_TRY
    if (step < 1) throw Error("Adam step count starts at 1");
    launch_adam(n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, S(stream));
    VFEM_CATCH
}

int vfem_timers_reset(void) {
    std::lock_guard<std::mutex> lk(vfem::g_timer_mu);
    vfem::g_timers.clear();
    return 0;
}
int vfem_timers_report(char *buf, size_t len) {
    std::lock_guard<std::mutex> lk(vfem::g_timer_mu);
    std::string out;
    for (auto &kv : vfem::g_timers) {
        char line[256];
        std::snprintf(line, sizeof(line), "%s\t%.6f\t%lld\n", kv.first.c_str(), kv.second.seconds, kv.second.calls);
        out += line;
    }
    if (len == 0) return 0;
    std::strncpy(buf, out.c_str(), len - 1);
    buf[len - 1] = 0;
    return 0;
}

}  // extern "C"
