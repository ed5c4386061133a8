/*
 * voxel_ref.c -- CPU ORACLE (test infrastructure, NOT a product path).
 *
 * Plain-C restatement of the element/node loops of the reference's voxel FEM
 * hot path (Nikronic/ndr, vendored VoxelFEM).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the
 * product (ndr_amd/) never links or calls it.
 *
 * Every function cites the reference code it follows (file:line, paths
 * relative to the reference checkout).  Conventions:
 *   - grids are row-major with the LAST axis fastest (NDVector.hh:284-292)
 *   - nodes per dim = elements per dim + 1 (degree-1 elements only here)
 *   - local node index inside an element follows the same rule
 *     (TensorProductSimulator.hh:291-315): 3-D n = 4i+2j+k, 2-D n = 2i+j
 *   - nodal fields are [numNodes][N] row-major (TensorProductSimulator.hh:227)
 *   - element matrices are KS x KS (KS = N * 2^N), dof = N*localNode + comp,
 *     stored row-major here (they are symmetric, so the reference's
 *     column-major Eigen storage holds the same bytes).
 *
 * Parity status: pinned by the reference's logged compliance values
 * (tests/golden/reference_logs.json) through oracle/vfem_oracle.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int N;            /* 2 or 3 */
    int npe;          /* nodes per element = 2^N */
    int ks;           /* N * npe */
    long ne[3];       /* elements per dim */
    long nn[3];       /* nodes per dim */
    long nstr[3];     /* node flat-index stride per dim */
    long estr[3];     /* element flat-index stride per dim */
    long num_nodes, num_elems;
    long noff[8];     /* flat node offset of local node m from the element's first node */
} grid_t;

static void grid_init(grid_t *g, int N, const long *ne)
{
    g->N = N; g->npe = 1 << N; g->ks = N * g->npe;
    for (int d = 0; d < 3; ++d) { g->ne[d] = 1; g->nn[d] = 1; }
    for (int d = 0; d < N; ++d) { g->ne[d] = ne[d]; g->nn[d] = ne[d] + 1; }
    long s = 1, t = 1;
    for (int d = N - 1; d >= 0; --d) { g->nstr[d] = s; s *= g->nn[d]; g->estr[d] = t; t *= g->ne[d]; }
    g->num_nodes = s; g->num_elems = t;
    /* local node m -> Nd index, last axis fastest (TensorProductSimulator.hh:291-315) */
    for (int m = 0; m < g->npe; ++m) {
        long off = 0;
        for (int d = 0; d < N; ++d) {
            int bit = (m >> (N - 1 - d)) & 1;
            off += bit * g->nstr[d];
        }
        g->noff[m] = off;
    }
}

/* flattenedFirstNodeOfElement1D, TensorProductSimulator.hh:998-1012 */
static inline long first_node_of_element(const grid_t *g, long ei)
{
    long result = 0;
    for (int d = g->N - 1; d >= 0; --d) {
        long ei_d = ei % g->ne[d];
        ei /= g->ne[d];
        result += ei_d * g->nstr[d];
    }
    return result;
}

/* ------------------------------------------------------------------------- */
/* SIMP: E_e = E_min + rho^gamma (E_0 - E_min), TensorProductSimulator.hh:725-727 */
void ref_simp(long n, const double *rho, double E0, double Emin, double gamma, double *E)
{
    for (long i = 0; i < n; ++i) E[i] = Emin + pow(rho[i], gamma) * (E0 - Emin);
}

/* ------------------------------------------------------------------------- */
/*
 * applyK, TensorProductSimulator.hh:905-952 with the thread-private
 * accumulation of ParallelAssembly.hh:49-112: every worker but the first
 * accumulates into its own full-size zeroed copy of the output, then the
 * copies are summed serially into the result.
 *   mode 0: Ke_e = E[e] * K0             (non-cached branch, :933-951)
 *   mode 1: Ke_e = Ke[e] (KS*KS each)    (cached branch, :914-932)
 */
static void apply_element(const grid_t *g, long ei, const double *K, double scale,
                          const double *u, double *out)
{
    const int N = g->N, npe = g->npe, ks = g->ks;
    const long off = first_node_of_element(g, ei);
    double w[24];
    /* Ke_u_local = K[:, 0:N] * u_0 ; += K[:, N m : N m + N] * u_m  (:936-939) */
    for (int r = 0; r < ks; ++r) w[r] = 0.0;
    for (int m = 0; m < npe; ++m) {
        const double *um = u + N * (off + g->noff[m]);
        for (int r = 0; r < ks; ++r) {
            double acc = 0.0;
            for (int c = 0; c < N; ++c) acc += K[r * ks + N * m + c] * um[c];
            w[r] += acc;
        }
    }
    for (int m = 0; m < npe; ++m) {
        double *om = out + N * (off + g->noff[m]);
        for (int c = 0; c < N; ++c) om[c] += scale * w[N * m + c];
    }
}

void ref_apply_k(int N, const long *ne, int mode, const double *K0_or_Ke, const double *E,
                 const double *u, double *out, int nthreads)
{
    grid_t g; grid_init(&g, N, ne);
    const long nd = g.num_nodes * N;
    const int ks2 = g.ks * g.ks;
    memset(out, 0, sizeof(double) * nd);
    if (nthreads <= 1) {
        for (long ei = 0; ei < g.num_elems; ++ei)
            apply_element(&g, ei, mode ? K0_or_Ke + ei * ks2 : K0_or_Ke, mode ? 1.0 : E[ei], u, out);
        return;
    }
#ifdef _OPENMP
    double **priv = (double **) calloc(nthreads, sizeof(double *));
    #pragma omp parallel num_threads(nthreads)
    {
        const int t = omp_get_thread_num();
        double *dst = out;
        if (t > 0) { priv[t] = (double *) calloc(nd, sizeof(double)); dst = priv[t]; }
        #pragma omp for schedule(static)
        for (long ei = 0; ei < g.num_elems; ++ei)
            apply_element(&g, ei, mode ? K0_or_Ke + ei * ks2 : K0_or_Ke, mode ? 1.0 : E[ei], u, dst);
    }
    /* serial reduction, ParallelAssembly.hh:101-102 */
    for (int t = 1; t < nthreads; ++t) {
        if (!priv[t]) continue;
        for (long i = 0; i < nd; ++i) out[i] += priv[t][i];
        free(priv[t]);
    }
    free(priv);
#else
    for (long ei = 0; ei < g.num_elems; ++ei)
        apply_element(&g, ei, mode ? K0_or_Ke + ei * ks2 : K0_or_Ke, mode ? 1.0 : E[ei], u, out);
#endif
}

/* ------------------------------------------------------------------------- */
/* complianceGradient, TensorProductSimulator.hh:730-751:
 * g_e = -0.5 gamma rho^(gamma-1) (E0-Emin) u_e^T K0 u_e */
void ref_compliance_gradient(int N, const long *ne, const double *K0, const double *rho,
                             double E0, double Emin, double gamma, const double *u, double *gout,
                             int nthreads)
{
    grid_t g; grid_init(&g, N, ne);
    const int ks = g.ks, npe = g.npe;
    (void) nthreads;
    #pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (long ei = 0; ei < g.num_elems; ++ei) {
        const long off = first_node_of_element(&g, ei);
        double ue[24];
        for (int m = 0; m < npe; ++m)
            for (int c = 0; c < N; ++c) ue[N * m + c] = u[N * (off + g.noff[m]) + c];
        double e = 0.0;
        for (int r = 0; r < ks; ++r) {
            double acc = 0.0;
            for (int c = 0; c < ks; ++c) acc += K0[r * ks + c] * ue[c];
            e += ue[r] * acc;
        }
        gout[ei] = -0.5 * gamma * pow(rho[ei], gamma - 1.0) * (E0 - Emin) * e;
    }
}

/* ------------------------------------------------------------------------- */
/*
 * One multicoloured block Gauss-Seidel sweep:
 *   visitNodesMulticolored  MultigridSolver.hh:285-326 (colour = local node
 *       index of the reference element; degree 1 => stride 2 per axis; reverse
 *       colour order on the reverse sweep)
 *   m_smoothNode            MultigridSolver.hh:193-265
 *   visitIncidentElements   TensorProductSimulator.hh:1233-1281
 *   mode 0: matrix-free K0 * E[e] (:199-220); mode 1: cached Ke (:221-240).
 * dmask is [numNodes][N] uint8 (ComponentMask per node).
 */
static void smooth_node(const grid_t *g, const long *gi, int mode, const double *K, const double *E,
                        double *u, const double *b, const uint8_t *dmask, int forward)
{
    const int N = g->N, npe = g->npe, ks = g->ks;
    long n = 0;
    for (int d = 0; d < N; ++d) n += gi[d] * g->nstr[d];
    double bms[3], M[9];
    for (int c = 0; c < N; ++c) bms[c] = b[N * n + c];
    for (int i = 0; i < N * N; ++i) M[i] = 0.0;

    /* primary element / local node (TPS.hh:1242-1251) */
    long elem[3]; int loc[3];
    for (int d = 0; d < N; ++d) {
        elem[d] = gi[d];
        if (elem[d] == g->ne[d]) { elem[d] = g->ne[d] - 1; loc[d] = 1; } else loc[d] = 0;
    }
    for (int i = 0; i < npe; ++i) {
        long e_idx[3]; int lnn[3]; int valid = 1;
        for (int d = 0; d < N; ++d) {
            e_idx[d] = elem[d]; lnn[d] = loc[d];
            if ((1 << d) & i) continue;
            if ((loc[d] != 0) || (e_idx[d] == 0)) { valid = 0; break; }
            --e_idx[d]; lnn[d] = 1;
        }
        if (!valid) continue;
        long ei = 0, off = 0; int li = 0;
        for (int d = 0; d < N; ++d) { ei += e_idx[d] * g->estr[d]; off += e_idx[d] * g->nstr[d]; li = 2 * li + lnn[d]; }
        const double *Ke = mode ? K + ei * (long) (ks * ks) : K;
        const double Ee = mode ? 1.0 : E[ei];
        double s[3] = {0.0, 0.0, 0.0};
        for (int m = 0; m < npe; ++m) {
            const double *um = u + N * (off + g->noff[m]);
            for (int r = 0; r < N; ++r)
                for (int c = 0; c < N; ++c) s[r] += Ke[(N * li + r) * ks + N * m + c] * um[c];
        }
        for (int r = 0; r < N; ++r) {
            bms[r] -= Ee * s[r];
            for (int c = 0; c < N; ++c) M[r * N + c] += Ee * Ke[(N * li + r) * ks + N * li + c];
        }
    }
    /* component-sequential solve (MG.hh:254-264) */
    double ud[3] = {0.0, 0.0, 0.0};
    if (forward) {
        for (int i = 0; i < N; ++i) {
            double t = bms[i];
            for (int c = 0; c < N; ++c) t -= M[i * N + c] * ud[c];
            ud[i] = t * ((double) (!dmask[N * n + i]) / M[i * N + i]);
        }
    } else {
        for (int i = N - 1; i >= 0; --i) {
            double t = bms[i];
            for (int c = 0; c < N; ++c) t -= M[i * N + c] * ud[c];
            ud[i] = t * ((double) (!dmask[N * n + i]) / M[i * N + i]);
        }
    }
    for (int c = 0; c < N; ++c) u[N * n + c] += ud[c];
}

void ref_gs_sweep(int N, const long *ne, int mode, const double *K0_or_Ke, const double *E,
                  double *u, const double *b, const uint8_t *dmask, int forward, int nthreads)
{
    grid_t g; grid_init(&g, N, ne);
    const int ncol = g.npe;
    for (int ci = 0; ci < ncol; ++ci) {
        const int lni = forward ? ci : (ncol - ci - 1);
        long l[3] = {0, 0, 0}, cnt[3] = {1, 1, 1};
        long total = 1;
        for (int d = 0; d < N; ++d) {
            l[d] = (lni >> (N - 1 - d)) & 1;           /* last axis fastest */
            cnt[d] = (g.nn[d] - 1 - l[d]) / 2 + 1;     /* MG.hh:309, increment 2 */
            total *= cnt[d];
        }
        #pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
        for (long i = 0; i < total; ++i) {
            long r = i, gi[3] = {0, 0, 0};
            for (int d = N - 1; d >= 0; --d) { gi[d] = l[d] + 2 * (r % cnt[d]); r /= cnt[d]; }
            smooth_node(&g, gi, mode, K0_or_Ke, E, u, b, dmask, forward);
        }
    }
}

/* ------------------------------------------------------------------------- */
/*
 * restriction, MultigridSolver.hh:146-161: every FINE node scatters its value
 * to the 2^N nodes of the coarse element containing it with the coarse shape
 * functions evaluated at the fine node (TensorProductPolynomialRestriction,
 * TensorProductPolynomialInterpolant.hh:91-101).  getElementNDIndex clamps the
 * upper boundary into the last element (TensorProductSimulator.hh:1178-1199).
 * `nec` = COARSE elements per dim; fine grid has 2*nec.
 */
static inline void coarse_elem_and_weights(const grid_t *gc, const long *fi, long *ce, double w[3][2])
{
    for (int d = 0; d < gc->N; ++d) {
        long e = fi[d] / 2;
        if (e >= gc->ne[d]) e = gc->ne[d] - 1;
        const double c = 0.5 * (double) (fi[d] - 2 * e);   /* reference coordinate 0, 0.5 or 1 */
        ce[d] = e; w[d][0] = 1.0 - c; w[d][1] = c;
    }
}

void ref_restrict(int N, const long *nec, int ncomp, const double *fine, double *coarse)
{
    grid_t gc; grid_init(&gc, N, nec);
    long nef[3]; for (int d = 0; d < N; ++d) nef[d] = 2 * nec[d];
    grid_t gf; grid_init(&gf, N, nef);
    memset(coarse, 0, sizeof(double) * gc.num_nodes * ncomp);
    for (long nf = 0; nf < gf.num_nodes; ++nf) {
        long r = nf, fi[3] = {0, 0, 0}, ce[3] = {0, 0, 0};
        for (int d = N - 1; d >= 0; --d) { fi[d] = r % gf.nn[d]; r /= gf.nn[d]; }
        double w[3][2];
        coarse_elem_and_weights(&gc, fi, ce, w);
        long off = 0; for (int d = 0; d < N; ++d) off += ce[d] * gc.nstr[d];
        for (int m = 0; m < gc.npe; ++m) {
            double coef = 1.0;
            for (int d = 0; d < N; ++d) coef *= w[d][(m >> (N - 1 - d)) & 1];
            double *dst = coarse + ncomp * (off + gc.noff[m]);
            for (int c = 0; c < ncomp; ++c) dst[c] += coef * fine[ncomp * nf + c];
        }
    }
}

/* interpolation / accum_interpolation, MultigridSolver.hh:116-141 */
void ref_prolong(int N, const long *nec, int ncomp, const double *coarse, double *fine, int accumulate,
                 int nthreads)
{
    grid_t gc; grid_init(&gc, N, nec);
    long nef[3]; for (int d = 0; d < N; ++d) nef[d] = 2 * nec[d];
    grid_t gf; grid_init(&gf, N, nef);
    #pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (long nf = 0; nf < gf.num_nodes; ++nf) {
        long r = nf, fi[3] = {0, 0, 0}, ce[3] = {0, 0, 0};
        for (int d = N - 1; d >= 0; --d) { fi[d] = r % gf.nn[d]; r /= gf.nn[d]; }
        double w[3][2];
        coarse_elem_and_weights(&gc, fi, ce, w);
        long off = 0; for (int d = 0; d < N; ++d) off += ce[d] * gc.nstr[d];
        double v[3] = {0.0, 0.0, 0.0};
        for (int m = 0; m < gc.npe; ++m) {
            double coef = 1.0;
            for (int d = 0; d < N; ++d) coef *= w[d][(m >> (N - 1 - d)) & 1];
            const double *src = coarse + ncomp * (off + gc.noff[m]);
            for (int c = 0; c < ncomp; ++c) v[c] += coef * src[c];
        }
        for (int c = 0; c < ncomp; ++c) {
            if (accumulate) fine[ncomp * nf + c] += v[c]; else fine[ncomp * nf + c] = v[c];
        }
    }
}

/* ------------------------------------------------------------------------- */
/*
 * Galerkin coarse element matrices, MultigridSolver.hh:604-669.
 * phis[fi][fine_n][coarse_n] (MG.hh:559-583) with fi bit d <-> dimension d
 * (MG.hh:574, 595).  Ke_c = sum_fi I_fi^T Ke_f I_fi, I[N i + c, N j + d] =
 * phi(i,j) delta_cd (MG.hh:623-637).
 *   level 1 (finer has no cached Ke): Ke_c = sum_fi E_f * (I_fi^T K0 I_fi)  (:639-657)
 *   deeper: from the finer level's cached Ke (:658-666).
 */
static void fill_phis(int N, double phis[8][8][8])
{
    const int npe = 1 << N;
    for (int fi = 0; fi < npe; ++fi)
        for (int fn = 0; fn < npe; ++fn) {
            double p[3];
            for (int d = 0; d < N; ++d) {
                const int bit = (fn >> (N - 1 - d)) & 1;       /* fine local node coordinate (0/1) */
                p[d] = 0.5 * bit + ((fi & (1 << d)) ? 0.5 : 0.0);
            }
            for (int cn = 0; cn < npe; ++cn) {
                double coef = 1.0;
                for (int d = 0; d < N; ++d) coef *= ((cn >> (N - 1 - d)) & 1) ? p[d] : (1.0 - p[d]);
                phis[fi][fn][cn] = coef;
            }
        }
}

static void accumulate_coarsened(int N, const double phi[8][8], const double *Kf, double scale, double *Kc)
{
    const int npe = 1 << N, ks = N * npe;
    double T[24 * 24];
    /* T = Kf * I : T[a, N j + d] = sum_i Kf[a, N i + d] phi[i][j] */
    for (int a = 0; a < ks; ++a)
        for (int j = 0; j < npe; ++j)
            for (int d = 0; d < N; ++d) {
                double acc = 0.0;
                for (int i = 0; i < npe; ++i) acc += Kf[a * ks + N * i + d] * phi[i][j];
                T[a * ks + N * j + d] = acc;
            }
    /* Kc += I^T T : Kc[N j + c, b] += sum_i phi[i][j] T[N i + c, b] */
    for (int j = 0; j < npe; ++j)
        for (int c = 0; c < N; ++c)
            for (int bcol = 0; bcol < ks; ++bcol) {
                double acc = 0.0;
                for (int i = 0; i < npe; ++i) acc += phi[i][j] * T[(N * i + c) * ks + bcol];
                Kc[(N * j + c) * ks + bcol] += scale * acc;
            }
}

/* coarsenedK0s[fi] = I_fi^T K0 I_fi  (MG.hh:644-648); out is [2^N][ks*ks] */
void ref_coarsened_k0(int N, const double *K0, double *out)
{
    const int npe = 1 << N, ks = N * npe;
    double phis[8][8][8]; fill_phis(N, phis);
    memset(out, 0, sizeof(double) * npe * ks * ks);
    for (int fi = 0; fi < npe; ++fi) accumulate_coarsened(N, phis[fi], K0, 1.0, out + fi * ks * ks);
}

/* mode 0: from fine moduli E (level 1); mode 1: from the finer level's Ke. nec = coarse elems/dim */
void ref_coarsen_ke(int N, const long *nec, int mode, const double *K0, const double *E_or_Kef,
                    double *Kec, int nthreads)
{
    grid_t gc; grid_init(&gc, N, nec);
    long nef[3]; for (int d = 0; d < N; ++d) nef[d] = 2 * nec[d];
    grid_t gf; grid_init(&gf, N, nef);
    const int npe = gc.npe, ks = gc.ks, ks2 = ks * ks;
    double phis[8][8][8]; fill_phis(N, phis);
    double *cK0 = (double *) malloc(sizeof(double) * npe * ks2);
    if (mode == 0) ref_coarsened_k0(N, K0, cK0);
    #pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (long ec = 0; ec < gc.num_elems; ++ec) {
        long r = ec, ci[3] = {0, 0, 0};
        for (int d = N - 1; d >= 0; --d) { ci[d] = r % gc.ne[d]; r /= gc.ne[d]; }
        double *Kc = Kec + ec * ks2;
        for (int i = 0; i < ks2; ++i) Kc[i] = 0.0;
        for (int fi = 0; fi < npe; ++fi) {
            long ef = 0;
            for (int d = 0; d < N; ++d) ef += (2 * ci[d] + ((fi & (1 << d)) ? 1 : 0)) * gf.estr[d];
            if (mode == 0) {
                const double Ef = E_or_Kef[ef];
                const double *C = cK0 + fi * ks2;
                for (int i = 0; i < ks2; ++i) Kc[i] += Ef * C[i];
            } else {
                accumulate_coarsened(N, phis[fi], E_or_Kef + ef * ks2, 1.0, Kc);
            }
        }
    }
    free(cK0);
}

int ref_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
