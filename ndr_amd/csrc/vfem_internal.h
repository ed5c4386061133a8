// Internal declarations shared by the libvfem translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vfem.h"

namespace vfem {

void set_error(const std::string &msg);

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define VFEM_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            throw ::vfem::Error(std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

// Regular grid of degree-1 hexahedra, row-major with the last axis fastest.
struct Dims {
    int nx, ny, nz;      // elements per dim
    int NX, NY, NZ;      // nodes per dim
    long long nn, ne;    // totals
    Dims() = default;
    Dims(long long ex, long long ey, long long ez)
        : nx((int) ex), ny((int) ey), nz((int) ez), NX((int) ex + 1), NY((int) ey + 1), NZ((int) ez + 1),
          nn((ex + 1) * (ey + 1) * (ez + 1)), ne(ex * ey * ez) {}
};

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void release() { if (p) { (void) hipFree(p); p = nullptr; n = 0; } }
    void alloc(size_t count) {
        if (count == n && p) return;
        release();
        if (count == 0) return;
        VFEM_HIP(hipMalloc((void **) &p, count * sizeof(T)));
        n = count;
    }
    void reserve(size_t count) { if (!(p && n >= count)) alloc(count); }          // scratch: grows, never shrinks
    void zero(hipStream_t s) { if (p) VFEM_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

// Operator representation of a multigrid level.
enum OpKind {
    OP_MF0 = 0,      // matrix-free, Ke = E_e * K0                       (finest level)
    OP_MF1 = 1,      // matrix-free Galerkin, Ke = sum_f E_f * cK0[f]    (first coarse level)
    OP_STENCIL = 2,  // stored 27-point 3x3-block stencil                (deeper levels)
};

// ------------------------------------------------------------------------------------------
// kernel launchers (kernels_*.hip)
// ------------------------------------------------------------------------------------------
void launch_simp(long long n, const double *rho, double E0, double Emin, double gamma, double *E, hipStream_t s);
void launch_fill(long long n, double v, double *x, hipStream_t s);

// out = K u (res = 0) or out = zeroDirichlet(b - K u) (res = 1; mask may be null => no zeroing)
void launch_apply_gather(const Dims &d, OpKind kind, const double *K, const double *E, const double *u,
                         const double *b, const uint8_t *mask, int res, double *out, hipStream_t s);
void launch_apply_stencil(const Dims &d, const double *S, const double *u, const double *b, const uint8_t *mask,
                          int res, double *out, hipStream_t s);
// Per-simulator choices between equivalent (bitwise or to rounding) implementations: cross-checks and tuning, set through
// vfem_sim_set_option.  Nothing here changes results beyond rounding; the wrong-result timing ablations are a separate
// build (-DVFEM_ABLATION, `make ablation`) and do not exist in the shipped library.
struct Tuning {
    int apply_impl = 0;     // 0: LDS-DMA kernel (register-staged kernel when it cannot run), 1: register-staged kernel
    int apply_pd = 2;       // register-staged kernel: node planes in flight (2..4)
    int dma_chunks = 0;     // x-chunks of the marching blocks (0 = default)
    int dma_strip = 1;      // 0: main tile shape only, 1: strip tiles for the left-over node columns, 2: strip with main-length chunks
    int dma_lx = 0;         // line-exclusive z tiling of the LDS-DMA apply: 0 off (default: measured slower), 1 where it costs no extra z tile, 2 always
    int gs_variant = 0;     // 0: row-streaming / symmetric sweeps, 1: plain gather sweeps
    int gs_pair = 1;        // level 0: both z colours of a row in one launch
    int gs_resident = 0;    // level 0: K0 kept in 72 SGPRs (36 distinct values; set when build_gs_coef reproduces K0 bit for bit)
    int l1_split = 4;       // level 1: waves sharing the eight element slots of a node (1: one lane does all eight; 2, 4, 8)
    int stencil_split = 1;  // stored-stencil levels above the wave-per-node threshold: 27 neighbour blocks shared by three waves (1) or one lane (0)
    int gs_march = 1;       // level 0: plane-resident x-marching half sweeps (kernels_gs_march.hip) where whole colour groups are swept; 0: row kernels
    int gs_march_chunks = 0;   // x-chunks of the marching sweep (0 = default)
    int l1_stored = 0;      // level 1: 0 virtual Galerkin operator (sum_f E_f cK0[f] at every visit), 1 stored 27-point block stencil (1944 B per
                            // node, built once per operator update; measured slower), 2 stored HALF stencil (symmetry: 1008 B per node)
    int l1_merged = 2;      // level 1: node rows evaluated per incident element (k_gs_color_mf1_sym*, k_apply_gather<1>; 0), per mirror class by
                            // three waves per node (kernels_l1_merged.hip; 1), the same with the two z colours of a row in one launch (2: the
                            // second colour finds the moduli in L2; bit for bit the result of 1)
    int l1_diag = 0;        // level 1: diagonal blocks precomputed once per operator update instead of inside every sweep (measured
                            // 3 % SLOWER at 512^3, profiles/r02_gs_experiments.json: the sweep is not bound by its arithmetic)
};
#ifdef VFEM_ABLATION
extern int g_ablate_apply, g_ablate_store, g_ablate_mlp;   // vfem_debug_set (ablation build only): wrong results, timing only
inline int ablate_apply() { return g_ablate_apply; }
inline int ablate_store() { return g_ablate_store; }
inline int ablate_mlp() { return g_ablate_mlp; }
#else
constexpr int ablate_apply() { return 0; }
constexpr int ablate_store() { return 0; }
constexpr int ablate_mlp() { return 0; }
#endif

// production level-0 apply (symmetry-reduced, x-marching)
void launch_apply_fast(const Dims &d, const double *Dm_host, const double *E, const double *u, const double *b,
                       const uint8_t *mask, int mode, double *out, hipStream_t s, int planes_in_flight = 2);
// LDS-DMA version of the plain apply (mode 0); returns false when it must not be used for these buffers
bool launch_apply_dma(const Dims &d, const double *Dm_host, const double *E, const double *E_alloc_end, const double *u,
                      double *out, hipStream_t s, int plane_lo = 0, int plane_hi = -1, int chunks = 0, int strip = 1,
                      const double *rhs = nullptr, const uint8_t *fixed = nullptr, int line_exclusive = 0);      // rhs: out = rhs - K u, 0 at fixed components

// colours are processed in the reference order (global parity); `xparity` = global x-parity of local plane 0,
// [first, first+count) selects a sub-range of the 8 colours (half sweeps between halo exchanges)
void launch_gs_sweep_mf(const Dims &d, OpKind kind, const double *K, const double *gs_tab, const double *E, double *u,
                        const double *b, const uint8_t *mask, int forward, int xparity, int first, int count, hipStream_t s,
                        const Tuning &tune, bool mf1_sym, const double *mf1_diag = nullptr);
// level 0: one half sweep (the four colours of local x parity cxl) marching along x with the planes in LDS; out of place:
// relaxed planes read from uR, the others from uO, results to dst != uR.  false: cannot run on these buffers
bool launch_gs_march_mf0(const Dims &d, const double *neighbour_kind_table, const double *E, const double *uR, const double *uO, double *dst, const double *b,
                         const double *solve_data, int cxl, int forward, int chunks, hipStream_t s, int plane_lo = 0, int plane_hi = -1);     // form 2 needs the neighbour-kind table of K0
// per node { 1/M00, 1/M11, 1/M22 (0 where the component is fixed), M10, M20, M21 } of the level-0 diagonal blocks M = sum_e E_e K0[n-block]
void launch_gs_solve_data(const Dims &d, const double *K0, const double *E, const uint8_t *mask, double *sd, hipStream_t s);
void launch_copy_planes(const Dims &d, int par, const double *src, double *dst, hipStream_t s, int plane_lo = 0, int plane_hi = -1);
// level 1: diagonal 3x3 blocks of the virtual Galerkin operator, [nn][9] (once per operator update)
void launch_mf1_diag(const Dims &d, const double *Dtab, const double *E, double *Mdiag, hipStream_t s);
void build_gs_table(const double *K0, double *tab /* 72*12 doubles */);
constexpr int GS_TABLE_DOUBLES = 72 * 12;
bool build_gs_coef(const double *K0, double *coef /* 36 doubles; false: K0 lacks the box-voxel / isotropic structure */);
bool coarsened_matrices_are_mirror_images(const double *cK0_host /* 8 x 576 */);
void build_mf1_diag_table(const double *cK0_0, double *tab /* 8*12 */);
// level 1, node rows per mirror class (kernels_l1_merged.hip); the table holds cK0[0] regrouped by class with the signs folded in
constexpr int L1M_TABLE_DOUBLES = 8 * 8 * 12;
void build_l1_merged_table(const double *cK0_0, double *tab /* L1M_TABLE_DOUBLES */);
bool l1_merged_usable(const Dims &d);
void launch_l1_merged_sweep(const Dims &d, const double *tab, const double *E, double *u, const double *b, const uint8_t *mask,
                            int forward, int xparity, int first, int count, hipStream_t s, int pair = 1);
void launch_l1_merged_apply(const Dims &d, const double *tab, const double *E, const double *u, const double *b, const uint8_t *mask,
                            int res, double *out, hipStream_t s);
void launch_gs_sweep_stencil(const Dims &d, const double *S, double *u, const double *b, const uint8_t *mask,
                             int forward, int xparity, int first, int count, hipStream_t s, const double *S_node_major = nullptr,
                             int stencil_split = 1);
// level 1 stored as half a stencil (kernels_stencil_half.hip): diagonal block + the 13 later neighbours per node, 1008 B
long long stencil_half_storage_doubles(const Dims &d);
void launch_stencil_half_from_mf1(const Dims &d, const double *cK0, const double *Efine, double *Sh, hipStream_t s);
void launch_gs_sweep_stencil_half(const Dims &d, const double *Sh, double *u, const double *b, const uint8_t *mask,
                                  int forward, int xparity, int first, int count, hipStream_t s);
void launch_apply_stencil_half(const Dims &d, const double *Sh, const double *u, const double *b, const uint8_t *mask, int res, double *out,
                               hipStream_t s);
// node-major copy of a level's stencil (levels small enough for the wave-per-node sweep, WAVE_SWEEP_MAX_NODES)
void launch_stencil_node_major(const Dims &d, const double *St, double *Sn, hipStream_t s);
constexpr long long WAVE_SWEEP_MAX_NODES = 40000;

// fine local plane index = 2 * (coarse local plane) + shift + {-1,0,1}; fineNX = fine local node planes
// zeroed (optional): a second coarse field set to zero by the same launch (the V-cycle's coarse initial guess)
void launch_restrict(const Dims &coarse, int fineNX, int shift, const double *fine, double *coarse_out, hipStream_t s, double *zeroed = nullptr);
// fixed (optional, accumulate == 0): the fine level's Dirichlet mask; its components of the interpolated field are set to zero
void launch_prolong(const Dims &coarse, int fineNX, int shift, const double *coarse_in, double *fine, int accumulate, hipStream_t s,
                    const uint8_t *fixed = nullptr);

void launch_zero_dirichlet(long long nn, const uint8_t *mask, double *u, hipStream_t s);
void launch_enforce_dirichlet(long long nn, const uint8_t *mask, const double *vals, double *u, int zero, hipStream_t s);

// Galerkin element matrices: mode 1 = children are virtual level-1 matrices (from fine moduli + cK0),
// mode 2 = children read from Kef.  `c` = dims of the level being built.
void launch_coarsen_ke(const Dims &c, int mode, const double *cK0, const double *Efine, const double *Kef,
                       double *Kec, hipStream_t s);
long long stencil_storage_doubles(const Dims &d);            // size of a level's stencil array (tiles padded per colour)
void launch_stencil_from_ke(const Dims &d, const double *Ke, double *S, hipStream_t s);
void launch_stencil_from_mf(const Dims &d, OpKind kind, const double *K, const double *E, double *S, hipStream_t s);
void launch_dense_from_stencil(const Dims &d, const double *S, const uint8_t *mask, double *A, hipStream_t s);
void launch_dense_finish_inverse(long long n, const uint8_t *mask, double *A, hipStream_t s);
void launch_gemv_sym(long long n, const double *A, const double *x, double *y, hipStream_t s);

void launch_compliance_gradient(const Dims &d, const double *K0, const double *rho, double E0, double Emin,
                                double gamma, const double *u, double *g, hipStream_t s);

// reductions: out[slot] = sum a[i]*b[i]; deterministic two-pass. scratch holds >= 2048 doubles.
void launch_dot(long long n, const double *a, const double *b, double *scratch, double *out, hipStream_t s);
void launch_dot_zero_dirichlet(long long n3, const double *a, double *b_masked_in_place, const uint8_t *mask, double *scratch, double *out, hipStream_t s);
void launch_pcg_step_dot(long long n3, double *x, double *r, const double *dv, const double *Ad, const double *sc, double *scratch,
                         double *rr_out, hipStream_t s);
// PCG vector updates with device-resident scalars (sc: [0]=rMr [1]=rMr_old [2]=dAd [3]=rr)
void launch_pcg_direction(long long n, const double *sv, double *dv, const double *sc, int first, hipStream_t s);
void launch_pcg_step(long long n, double *x, double *r, const double *dv, const double *Ad, const double *sc,
                     hipStream_t s);
void launch_shift_scalar(double *sc, hipStream_t s);   // sc[1] = sc[0]

void launch_box_filter(int nx, int ny, int nz, int r, const double *in, double *out, int transpose, hipStream_t s);
void launch_projection(long long n, double beta, const double *x, const double *g, double *out, int mode, hipStream_t s);
void launch_oc_candidate(long long n, const double *x0, const double *dJ, const double *dc, double lambda, double m, double *out, hipStream_t s);
void launch_sum(long long n, const double *a, double *scratch, double *out, hipStream_t s);

void launch_apply_q2(int nx, int ny, int nz, const double *K0, const double *E, const double *u, double *out, hipStream_t s);
void launch_apply_q2_pencil(int nx, int ny, int nz, const double *mode_table, const double *E, const double *u, double *out, hipStream_t s);
void launch_apply_q2_march(int nx, int ny, int nz, const double *mode_table, const double *E, const double *u, double *out, hipStream_t s);
void launch_gs_sweep_q2_level0(int nx, int ny, int nz, const double *K0, const double *E, double *u, const double *b,
                               const uint8_t *mask, int forward, hipStream_t s, int first = 0, int count = 27);
// Dense SPD inverse of the coarsest level (dense_spd.hip): A (n x n row-major, full symmetric) is replaced by its inverse (full
// symmetric).  The build's own blocked Cholesky / triangular inverse / product on one stream with a fixed summation order:
// bitwise reproducible whatever else shares the device.  Throws when a pivot is not positive.  The workspace grows, never shrinks.
struct DenseWork { DevBuf<double> L, X, Tm, D; DevBuf<int> info; };
void dense_spd_inverse(long long n, double *A, DenseWork &w, hipStream_t s);
void launch_gs_sweep_q2_level1(int nx, int ny, int nz, const double *cK0, const double *Ef, int fx0, double *u, const double *b,
                               const uint8_t *mask, int forward, hipStream_t s, int first = 0, int count = 27);
void launch_apply_q2_level1(int nx, int ny, int nz, const double *cK0, const double *Ef, int fx0, const double *u, const double *b,
                            const uint8_t *mask, int mode, double *out, hipStream_t s);
// the same sweep ordered by neighbour node (each distinct neighbour read once); tab from build_q2_gs_table
void build_q2_gs_table(const double *K0_host, std::vector<double> &tab);
void launch_gs_sweep_q2_level0_nodes(int nx, int ny, int nz, const double *tab, const double *E, double *u, const double *b,
                                     const uint8_t *mask, int forward, hipStream_t s, int first = 0, int count = 27, int rows_in_lds = 0);
void launch_q2_residual_fix(long long nn, const double *b, const uint8_t *mask, int mode, double *out, hipStream_t s);
void launch_gradient_q2(int nx, int ny, int nz, const double *K0, const double *rho, double E0, double Emin, double gamma,
                        const double *u, double *g, hipStream_t s);

struct MlpArgs;
void launch_mlp_forward(const MlpArgs &a, hipStream_t s);
void launch_f32_to_f16(long long n, const float *in, void *out, hipStream_t s);
struct MlpBwdArgs;
struct MlpDwArgs;
void launch_mlp_backward_x3(const MlpBwdArgs &a, long long rows, hipStream_t s);
void launch_mlp_dw(const MlpDwArgs &a, hipStream_t s);
void launch_colsum_split(long long rows, int ncols, const void *Xh, const void *Xl, const float *w, float *partial, hipStream_t s);
// reference-precision fused forward (kernels_mlp_x3.hip): split fp16 operands (hi + lo 2^-11), three MFMA products per product
void launch_f32_to_f16_frag(int N, int K, int transposed, const float *in, void *out, hipStream_t s);
void launch_range_check_f32(long long n, const float *in, float limit, int *flag, hipStream_t s);
void launch_split_f32_frag(int N, int K, const float *in, void *hi, void *lo, hipStream_t s, int pair_es, int transposed = 0, int pair_kc = 64);
void launch_mlp_forward_x3(const MlpArgs &a, const void *W1h, const void *W1l, const void *Whh, const void *Whl, hipStream_t s, int kc);
void launch_split_f32(long long n, const float *in, void *hi, void *lo, hipStream_t s);
void launch_reduce_partials(int nb, long long n, const float *partial, float alpha, float beta, float *out, hipStream_t s);
void launch_sum_f32(long long n, const float *x, float alpha, float beta, float *out, float *scratch, hipStream_t s);
void launch_adam(long long n, float *p, const float *g, float *m, float *v, float lr, float b1, float b2, float eps, int step, hipStream_t s);

}  // namespace vfem

// ------------------------------------------------------------------------------------------
// opaque handle definitions
// ------------------------------------------------------------------------------------------
struct vfem_sim {
    vfem::Dims d;
    double bbmin[3], bbmax[3], h[3];
    double lambda = 0.0, mu = 0.5;              // ETensor(1, 0) default, TPS.hh:1379
    double E0 = 1.0, Emin = 1e-9, gamma = 3.0;  // TPS.hh:1392-1394
    double K0[576];                             // host copy, row-major
    double Dm[64];                              // symmetry-reduced (mode-space) coefficients, host
    bool   fast_ok = false;                     // mode-space sparsity pattern verified for this K0
    bool   gs_resident_ok = false;              // the 36-value structure of K0 verified (build_gs_coef)
    bool k0_mirror_ok = false;                  // K0 commutes with the axis reflections (update_k0)
    vfem::DevBuf<double> dK0, dGsTab, rho, E, dvals, loads;
    vfem::DevBuf<uint8_t> dmask;
    std::vector<uint8_t> hmask;                 // host copy of the Dirichlet mask
    std::vector<double> hvals;
    bool nonzero_dirichlet = false;
    // slab decomposition: element arrays (rho, E) may hold extra x-layers in front of / behind the node grid
    long long ex_lo = 0, ex_hi = 0;
    long long operator_version = 1;             // bumped whenever K(rho) changes (densities, SIMP law, material): hierarchies rebuild
    vfem::Tuning tune;
    vfem::DevBuf<double> red;                   // scratch of the reductions (vfem_compliance)
    long long n_store() const { return (long long) (d.nx + ex_lo + ex_hi) * d.ny * d.nz; }
    const double *Ep() const { return E.p + ex_lo * d.ny * d.nz; }
    const double *rhop() const { return rho.p + ex_lo * d.ny * d.nz; }
    void update_k0();
};

struct MgLevel {
    vfem::Dims d;                               // node grid (local: owned + one ghost element layer per interior side)
    vfem::Dims da;                              // element-array dims (d plus ex_lo/ex_hi extra x-layers)
    long long ex_lo = 0, ex_hi = 0;
    int xshift = 0;                             // finer-level local plane of this level's local plane 0
    int xparity = 0;                            // global x-parity of local plane 0
    int fineNX = 0;                             // node planes of the next finer level (for the transfers)
    vfem::OpKind kind = vfem::OP_MF0;
    vfem::DevBuf<uint8_t> mask;                 // coarsened Dirichlet masks (levels >= 1)
    const uint8_t *maskp = nullptr;
    std::vector<uint8_t> hmask;
    vfem::DevBuf<double> Ke, S;                 // Galerkin element matrices / stencil (levels >= 2)
    vfem::DevBuf<double> Sh;                    // level 1: half stencil (diagonal block + 13 later neighbours per node), option l1_stored = 2
    vfem::DevBuf<double> Sn;                    // node-major copy of S on levels of at most WAVE_SWEEP_MAX_NODES nodes (wave-per-node sweep)
    vfem::DevBuf<double> Mdiag;                 // level 1: precomputed diagonal blocks [nn][9] of the virtual operator
    vfem::DevBuf<double> x, b, r;               // work vectors (m_x, m_b of MG.hh:755-756 + residual)
    vfem::DevBuf<double> tmp;                   // level 0: second copy of the field for the out-of-place marching half sweeps
    vfem::DevBuf<double> gs_sd;                 // level 0: solve data of the marching sweeps [nn][3] (launch_gs_solve_data)
    long long gs_sd_version = 0;                // fine->operator_version gs_sd was computed for
};

struct vfem_mg {
    vfem_sim *fine = nullptr;
    int L = 0;                                  // numCoarseningLevels
    std::vector<MgLevel> lv;
    vfem::DevBuf<double> cK0;                   // 8 x 576 coarsened reference matrices (MG.hh:644-648)
    vfem::DevBuf<double> mf1diag;               // 8 x 12: diagonal blocks of cK0[0] (level-1 Gauss-Seidel)
    vfem::DevBuf<double> l1mtab;                // cK0[0] by mirror class (build_l1_merged_table)
    vfem::DevBuf<double> c2K0;                  // 64 x 576: I_g^T cK0[f] I_g (level-2 element matrices from the fine moduli)
    vfem::DevBuf<double> Ainv;                  // coarsest-level dense inverse
    vfem::DevBuf<double> pr, pd, pAd, ps;       // PCG vectors
    vfem::DevBuf<double> scal, scratch;
    bool slab = false;                          // local part of an x-slab decomposition: no coarsest solver here
    int first_active = 0;                       // levels below are never cycled (replicated coarse hierarchy)
    int external_ke_level = 0;                  // > 0: the element matrices of this level were supplied (vfem_mg_import_level_ke)
    bool symmetric_gs = true;                   // MG.hh:758
    bool operators_valid = false;
    long long operators_version = 0;            // fine->operator_version the coarse operators were built for
    bool mf1_sym = false;                       // cK0[f] are mirror images of cK0[0]: level-1 sweeps read cK0[0] only
    vfem::DenseWork dense;                      // workspace of the coarsest-level inverse
};

struct vfem_mlp {
    int es = 0, nn = 0, n_layers = 0, sigmoid = 0;
    vfem::DevBuf<float> B, bias, wout;
    vfem::DevBuf<float> W1f, Whf;               // fp32 copies of the weights
    vfem::DevBuf<uint16_t> W1, Wh;               // fp16 bit patterns
    vfem::DevBuf<uint16_t> W1h, W1l, Whh, Whl;   // split operands of the reference-precision forward (hi = fp16(w), lo = fp16((w - hi) 2^11))
    float bout = 0.f;
    bool loaded = false;
    // training workspace (vfem_mlp_backward*): transposed hidden weights, per-chunk activations / gradients / features
    vfem::DevBuf<uint16_t> WhTh, WhTl;           // transposed hidden weights, split, fragment order (backward data pass)
    vfem::DevBuf<uint16_t> acts, acts_lo, dz, dz_lo;   // per chunk: saved activations / gradients wrt pre-activations as (hi, lo) pairs
    vfem::DevBuf<float> gs, partial, partial_b, out_chunk;
    int bwd_terms = 3;                           // VFEM_MLP_OPT_BWD_TERMS
    int kc = 64;                                 // feature chunk of the reference-precision forward (the K order W1h / W1l are packed for)
    vfem::DevBuf<int> range_flag;                // raised by the reference-precision kernels when a value leaves fp16's range
    // VFEM_MLP_OPT_KEEP_FIRST: the first layer's activations of the last reference-precision grid forward, (hi, lo) pairs [voxels][nn], and
    // the grid / voxel range they belong to; the backward pass of the same range starts from them instead of recomputing two thirds of
    // the forward's products (68.7 GB at 512 x 256 x 256: the part has 288)
    int keep_first = 0;
    vfem::DevBuf<uint16_t> h0_hi, h0_lo;
    bool h0_valid = false;
    int64_t h0_n[3] = {0, 0, 0}, h0_first = 0, h0_count = 0;
    double h0_lo_c[3] = {0, 0, 0}, h0_hi_c[3] = {0, 0, 0};
};
