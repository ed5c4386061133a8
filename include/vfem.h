/*
 * vfem.h -- C ABI of libvfem, the MI355X-native voxel-FEM hot path.
 *
 * This is the drop-in boundary for the one path of Nikronic/ndr this repository accelerates
 * (SURVEY.md section 8): the matrix-free SIMP stiffness apply, the geometric-multigrid
 * preconditioned CG compliance solve, the compliance sensitivity and the Fourier-feature
 * MLP density field.  Each entry point names the reference interface it replaces
 * (paths relative to the reference checkout; "TPS" = VoxelFEM/TensorProductSimulator.hh,
 * "MG" = VoxelFEM/MultigridSolver.hh, "VoxelFEM.cc" = VoxelFEM/python_bindings/VoxelFEM.cc).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; vfem_last_error() then
 *     holds the message (the reference throws std::runtime_error -> Python RuntimeError).
 *   - all `double*` / `float*` / `uint8_t*` array arguments are DEVICE pointers (HBM) unless
 *     the parameter name ends in `_host`.  vfem_malloc / vfem_copy_* are provided so a caller
 *     without another HIP allocator (cgo, JNI, plain C) can stage data.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls enqueue work
 *     and return; calls that return scalars to the host synchronise that stream.
 *   - grids: row-major, last axis fastest (NDVector.hh:284-292); nodal fields are
 *     [numNodes][3] doubles (TPS.hh:227); densities/gradients are [numElements] doubles.
 *   - vfem_sim / vfem_mg are the tuned degree-1 3-D path (TensorProductSimulator<1,1,1>); vfem_gsim / vfem_gmg
 *     cover the other instantiations (2-D, degree 2) with the same semantics.
 */
#ifndef VFEM_H
#define VFEM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vfem_sim vfem_sim;     /* TensorProductSimulator<1,1,1>                            (TPS.hh:219) */
typedef struct vfem_mg  vfem_mg;      /* MultigridSolver<1,1,1>                                    (MG.hh:11)   */
typedef struct vfem_mlp vfem_mlp;     /* networks.MLP (Fourier features + ReLU MLP)                (networks.py:128) */
typedef struct vfem_gsim vfem_gsim;   /* TensorProductSimulator<p,..,p>, N = 2 or 3, p = 1 or 2; <2,2,2> = 27-node hexahedra,
                                         unbound in the reference (VoxelFEM.cc:226-229) */
typedef struct vfem_gmg vfem_gmg;     /* MultigridSolver<p,..,p> of the generic path */

const char *vfem_last_error(void);
int  vfem_device_count(void);                 /* number of visible HIP devices (0 => no GPU) */
int  vfem_set_device(int device);
int  vfem_version(void);

/* Per-simulator choice between implementations that agree to rounding (cross-checks in tests/, tuning); the defaults are
   the production kernels.  (Timing ablations that produce wrong results are not part of this library: they exist only in
   the separate `make ablation` build used by tools/.) */
enum {
    VFEM_OPT_APPLY_PLANES = 0,   /* register-staged apply: node planes in flight, 2..4 */
    VFEM_OPT_GS_VARIANT   = 2,   /* 0 production sweeps, 1 plain gather sweeps */
    VFEM_OPT_APPLY_IMPL   = 4,   /* 0 LDS-DMA apply, 1 register-staged apply */
    VFEM_OPT_Q2_IMPL      = 6,   /* vfem_gsim: degree-2 apply 0 marching, 1 dense gather, 2 pencil */
    VFEM_OPT_DMA_CHUNKS   = 7,   /* x-chunks of the marching apply (0 = default) */
    VFEM_OPT_DMA_STRIP    = 9,   /* z-remainder strip tiles: 0 off, 1 on, 2 on with the main chunk length */
    VFEM_OPT_DMA_LX       = 11,  /* z tiling of the LDS-DMA apply whose tile boundaries fall on 128-byte lines of the result (tiles advance by 57 columns):
                                    0 off (default: measured 8 % slower at 512^3 -- nine full tiles against eight and a strip --, profiles/r04_apply_lx.txt),
                                    1 where it needs no extra tile (513 = 9 x 57), 2 always */
    VFEM_OPT_GS_PAIR      = 10,  /* level-0 Gauss-Seidel: fused z-colour pairs (1) or one launch per colour (0) */
    VFEM_OPT_L1_DIAG      = 12,  /* level-1 Gauss-Seidel: diagonal blocks precomputed per operator update (1) or inside every sweep (0) */
    VFEM_OPT_GS_RESIDENT  = 13,  /* level-0 Gauss-Seidel: K0 held in SGPRs (1, when K0 has the 36-value structure) or coefficient table (0) */
    VFEM_OPT_L1_SPLIT     = 15,  /* level-1 Gauss-Seidel (degree 1): waves sharing the eight element slots of a node, 1 / 2 / 4 / 8 */
    VFEM_OPT_STENCIL_SPLIT = 18, /* stored-stencil levels: the 27 neighbour blocks of a node shared by three waves (1, default) or one lane (0) */
    VFEM_OPT_GS_MARCH     = 19,  /* level-0 Gauss-Seidel: plane-resident x-marching half sweeps on grids of at least 0.8 M nodes (1, default), always (2),
                                    or the row-streaming kernels (0); the two agree to rounding (different summation order) */
    VFEM_OPT_GS_MARCH_CHUNKS = 20, /* x-chunks of the marching sweep (0 = default) */
    VFEM_OPT_L1_STORED    = 21,  /* level 1 (degree 1): operator evaluated on the fly from the child moduli (0), stored as a 27-point block stencil, 1944 B
                                    per node (1), or stored as half of it using the symmetry, 1008 B per node (2); DESIGN section 3.2 */
    VFEM_OPT_L1_MERGED    = 22,  /* level 1 (degree 1), operator evaluated on the fly: node rows summed per incident element (0), per mirror class of the
                                    child matrices by three waves per node (1), or per class with the two z colours of a row relaxed by one launch (2,
                                    default: the result of 1 bit for bit, half the moduli traffic); 0 and 1 agree to rounding */
    VFEM_OPT_Q2_GS_IMPL   = 16,  /* vfem_gsim: finest-level degree-2 sweep 0 element by element, 1 neighbour node by neighbour node, 2 the same with
                                    the neighbour rows staged through LDS by coalesced loads (default) */
    VFEM_OPT_TRANSFER_AXIS = 17, /* vfem_gsim: restriction / interpolation of 3-D levels above 100 k nodes axis by axis (1, default) or in one pass (0) */
    VFEM_OPT_Q2_L1_VIRTUAL = 14  /* vfem_gsim: level 1 of a degree-2 hierarchy evaluated as sum_f E_f cK0[f] on the fly (1), from stored 81 x 81
                                    element matrices (0), or chosen by their size (2, default: on the fly above 1.5 GB); read by the next
                                    vfem_gmg_update_operators */
};

/* ---- raw device memory helpers (for callers without their own HIP allocator) ---- */
int vfem_malloc(void **ptr, size_t bytes);
int vfem_free(void *ptr);
int vfem_copy_h2d(void *dst, const void *src_host, size_t bytes, void *stream);
int vfem_copy_d2h(void *dst_host, const void *src, size_t bytes, void *stream);
int vfem_copy_d2d(void *dst, const void *src, size_t bytes, void *stream);
int vfem_memset(void *dst, int value, size_t bytes, void *stream);
int vfem_stream_sync(void *stream);

/* ---- simulator: TensorProductSimulator(domain, numElemsPerDim), TPS.hh:252-316 ---- */
int vfem_sim_create(vfem_sim **out, const double bbox_min_host[3], const double bbox_max_host[3],
                    const int64_t nelems_host[3]);
int vfem_sim_destroy(vfem_sim *sim);
int64_t vfem_sim_num_nodes(const vfem_sim *sim);      /* TPS::numNodes,    TPS.hh:1134 */
int64_t vfem_sim_num_elements(const vfem_sim *sim);   /* TPS::numElements, TPS.hh:1137 */

/* ElasticityTensor::setIsotropic via readMaterial/setETensor (TPS.hh:326-339; ElasticityTensor.hh:100-115) */
int vfem_sim_set_isotropic(vfem_sim *sim, double young, double poisson);
/* E_0 / E_min / gamma properties (VoxelFEM.cc:77-79; TPS.hh:1170-1175) */
int vfem_sim_set_simp(vfem_sim *sim, double E0, double Emin, double gamma);
/* fullDensityElementStiffnessMatrix (TPS.hh:755): 24x24 doubles, row-major, to HOST */
int vfem_sim_set_option(vfem_sim *sim, int key, int value);
int vfem_sim_k0(const vfem_sim *sim, double *K0_host);

/* dirichletMask / dirichletValues properties (TPS.hh:413-442): mask[numNodes] bit c = component c fixed;
 * values[numNodes][3].  Loads: the nodal force field of buildLoadVector (TPS.hh:893-901). */
int vfem_sim_set_dirichlet(vfem_sim *sim, const uint8_t *mask_host, const double *values_host);
int vfem_sim_set_loads(vfem_sim *sim, const double *f, void *stream);   /* f: device [numNodes][3] */
int vfem_sim_build_load_vector(const vfem_sim *sim, double *f, void *stream);

/* setElementDensities / setUniformDensities / getDensities (TPS.hh:456-461, 567-570, 1157-1161).
 * Precomputes the SIMP moduli E_e = Emin + rho^gamma (E0-Emin) once (TPS.hh:725-727). */
int vfem_sim_set_densities(vfem_sim *sim, const double *rho, void *stream);
int vfem_sim_set_uniform_density(vfem_sim *sim, double rho, void *stream);
int vfem_sim_get_densities(const vfem_sim *sim, double *rho, void *stream);

/* applyK (TPS.hh:905-952): out = K(rho) u, Dirichlet conditions ignored.
 * variant 0 = production kernel, 1 = plain gather kernel (cross-check). */
int vfem_sim_apply_k(const vfem_sim *sim, const double *u, double *out, int variant, void *stream);
/* the same operator for the output node planes plane_lo..plane_hi (x index, inclusive) only; other planes of `out` are left
 * untouched.  Used by the slab-decomposed apply to overlap the halo exchange with the interior planes. */
int vfem_sim_apply_k_planes(const vfem_sim *sim, const double *u, double *out, int64_t plane_lo, int64_t plane_hi, void *stream);
/* complianceGradient (TPS.hh:730-751): g_e = -1/2 gamma rho^(gamma-1) (E0-Emin) u_e^T K0 u_e */
int vfem_sim_compliance_gradient(const vfem_sim *sim, const double *u, double *g, void *stream);
/* ComplianceObjective::compliance (TopologyOptimizationObjective.hh:39-41): 1/2 sum f.u, to host */
int vfem_compliance(const vfem_sim *sim, const double *f, const double *u, double *value_host, void *stream);

/* ---- multigrid: tps.multigridSolver(numCoarseningLevels), MG.hh:22-90 ---- */
int vfem_mg_create(vfem_mg **out, vfem_sim *fine, int num_coarsening_levels);
int vfem_mg_destroy(vfem_mg *mg);

/* ---- x-slab decomposition support (one process per GPU).  The reference is single-process; these entry
 * points let a host-side driver (ndr_amd/distributed.py) run the same V-cycle / PCG control flow (MG.hh:447-732) over
 * slabs, with one ghost node plane per interior side at every level.
 *   - vfem_sim_set_next_element_padding: the NEXT vfem_sim_create allocates its element arrays (densities, moduli) with
 *     extra_lo / extra_hi additional x-layers in front of / behind the node grid (so that the Galerkin operators of the
 *     ghost elements of coarser levels can be built locally); vfem_sim_set_densities then expects all stored layers.
 *   - vfem_mg_create_slab: local hierarchy with explicit per-level extents and Dirichlet masks (masks_host[l] has the
 *     level's local numNodes bytes); no coarsest-level solver.
 *   - vfem_mg_create_partial: hierarchy on a whole (replicated) grid whose levels < first_active_level are never cycled.
 *   - vfem_mg_smooth_colors: colours [first, first+count) of one sweep (halo exchanges happen between half sweeps).
 *   - vfem_mg_cycle_from_level: V-cycle (x in/out) or full-multigrid cycle (x out) of the residual system at `level`. */
typedef struct {
    int64_t nx;               /* local elements in x (owned layers + one ghost layer per interior side) */
    int64_t elem_extra_lo;    /* extra element x-layers stored in front of the node grid at this level */
    int64_t elem_extra_hi;
    int64_t xshift;           /* level >= 1: (finer level's local plane of this level's local plane 0) = xshift (0 or -1) */
    int32_t xparity;          /* global x-parity of local node plane 0 (Gauss-Seidel colours follow the global grid) */
} vfem_slab_level;
int vfem_sim_set_next_element_padding(int64_t extra_lo, int64_t extra_hi);
int64_t vfem_sim_num_stored_elements(const vfem_sim *sim);
int vfem_mg_create_slab(vfem_mg **out, vfem_sim *fine_local, int n_levels, const vfem_slab_level *levels_host,
                        const uint8_t *const *masks_host);
int vfem_mg_create_partial(vfem_mg **out, vfem_sim *fine, int num_coarsening_levels, int first_active_level);
int vfem_mg_smooth_colors(vfem_mg *mg, int level, double *u, const double *b, int forward, int first, int count, void *stream);
/* Colour group `group` (0: colours 0-3, 1: colours 4-7 of the sweep order = the four colours of one x parity, MG.hh:292-310) on the
 * node planes [plane_lo, plane_hi] of the level's local grid only: a slab rank relaxes its interface planes, starts the halo
 * exchange and relaxes the interior planes meanwhile.  vfem_mg_can_smooth_planes: 1 when the level is swept by the out-of-place
 * marching kernel, which is what makes plane ranges possible. */
int vfem_mg_can_smooth_planes(vfem_mg *mg, int level);
int vfem_mg_smooth_group_planes(vfem_mg *mg, int level, double *u, const double *b, int forward, int group, int64_t plane_lo, int64_t plane_hi,
                                void *stream);
int vfem_mg_cycle_from_level(vfem_mg *mg, int level, double *x, const double *b, int num_smoothing_steps, int fmg, void *stream);
int vfem_mg_num_levels(const vfem_mg *mg);                       /* = numCoarseningLevels + 1 */
int vfem_mg_level_dims(const vfem_mg *mg, int level, int64_t nelems_host[3]);
int64_t vfem_mg_level_num_nodes(const vfem_mg *mg, int level);
int vfem_mg_level_dirichlet_mask(const vfem_mg *mg, int level, uint8_t *mask_host);  /* coarsened masks, MG.hh:57-84 */
int vfem_mg_set_symmetric_gauss_seidel(vfem_mg *mg, int symmetric);                  /* MG.hh:92-94 */
/* debug_get_x / debug_get_b (MG.hh:734-735) and the PCG residual handed to it_callback (MG.hh:726-729):
 * device pointer of an internal field; which = 0: m_x[level], 1: m_b[level], 2: PCG residual r (level ignored). */
const double *vfem_mg_field_ptr(const vfem_mg *mg, int which, int level);

/* updateElementStiffnessMatrices + updateBlockKs (MG.hh:415-441): rebuild the coarse operators
 * (Galerkin, MG.hh:604-669) and the coarsest-level factorisation from the current densities. */
int vfem_mg_update_operators(vfem_mg *mg, void *stream);
/* Slab decomposition without a global density field.  export: the Galerkin element matrices (MG.hh:604-669) of `level` (>= 2)
 * for `count_x` element layers, built from this (local) hierarchy's moduli -- the children start at layer `child_first_layer`
 * of the child level's element array (of the fine array for level 2): [count_x * ny * nz][576] doubles.  import: hand a
 * partial (replicated) hierarchy the element matrices of its first active level; its operators are then rebuilt from them. */
int vfem_mg_export_level_ke(vfem_mg *mg, int level, int64_t child_first_layer, int64_t count_x, double *ke_out, void *stream);
int vfem_mg_import_level_ke(vfem_mg *mg, int level, const double *ke, void *stream);
/* MG::applyK(l,u) (MG.hh:353-358), computeResidual (MG.hh:401-413), smoothingMulticoloredGS
 * (MG.hh:336-340; forward != 0 => forward colour/component order), zeroOutDirichletComponents
 * (MG.hh:364-378), restriction (MG.hh:146-161), interpolation / accum_interpolation (MG.hh:116-141),
 * coarsest TPS::solve (TPS.hh:834-865). `level` fields have vfem_mg_level_num_nodes(level) x 3 doubles. */
int vfem_mg_apply_k(vfem_mg *mg, int level, const double *u, double *out, void *stream);
int vfem_mg_residual(vfem_mg *mg, int level, const double *u, const double *b, double *r, void *stream);
int vfem_mg_smooth(vfem_mg *mg, int level, double *u, const double *b, int forward, void *stream);
/* `sweeps` consecutive calls of smoothingMulticoloredGS in one direction, as vcycle makes them (MG.hh:525-528, 546-549).  On the
 * finest level the sweeps are out-of-place marching half sweeps (kernels_gs_march.hip) alternating between u and a scratch
 * vector: an even count ends in u without a copy. */
int vfem_mg_smooth_sweeps(vfem_mg *mg, int level, double *u, const double *b, int forward, int sweeps, void *stream);
int vfem_mg_zero_dirichlet(vfem_mg *mg, int level, double *u, void *stream);
int vfem_mg_restrict(vfem_mg *mg, int fine_level, const double *fine, double *coarse, void *stream);
int vfem_mg_interpolate(vfem_mg *mg, int fine_level, const double *coarse, double *fine, int accumulate, void *stream);
int vfem_mg_coarsest_solve(vfem_mg *mg, const double *b, double *x, void *stream);
/* The factorisation behind the exact coarsest solve (TPS::solve with CholmodFactorizer, TPS.hh:834-865, SparseMatrices.hh:1875-1965),
 * as this library does it: A (n x n doubles, row-major, symmetric positive definite, both triangles) is replaced in place by
 * its inverse (both triangles).  Own blocked Cholesky / triangular inverse / product kernels on `stream`, fixed summation
 * order: the same matrix gives the same inverse bit for bit in every run.  Fails ("not positive definite") on a pivot <= 0. */
int vfem_dense_spd_inverse(int64_t n, double *A, void *stream);

/* MG::solve (MG.hh:447-472): numSteps V-cycles (first one a full-multigrid cycle if fmg) on K x = f
 * starting from x (in/out). */
int vfem_mg_solve(vfem_mg *mg, double *x, const double *f, int num_steps, int num_smoothing_steps,
                  int stiffness_updated, int zero_dirichlet, int fmg, void *stream);

/* preconditionedConjugateGradient (MG.hh:679-732).  x is in/out (initial guess u -> solution).
 * residual_cb, if non-NULL, is called after every iteration with (it, ||r||) on the calling thread
 * (MultigridComplianceObjective::residual_cb, TopologyOptimizationObjective.hh:87-89, 101). */
typedef void (*vfem_residual_cb)(void *user, int iteration, double residual_norm);
int vfem_mg_pcg(vfem_mg *mg, double *x, const double *b, int max_iter, double tol,
                int mg_iterations, int mg_smoothing_iterations, int fmg,
                vfem_residual_cb residual_cb, void *cb_user,
                int *iterations_out_host, double *relres_out_host, void *stream);

/* A rank's whole slab-decomposed MG-PCG solve in one call (DistributedMGSolver.pcg of ndr_amd/distributed.py, i.e.
 * preconditionedConjugateGradient / vcycle / fullMultigrid of MG.hh:486-553, 679-732 over x-slabs).  `local` = the rank's
 * vfem_mg_create_slab hierarchy (levels 0 .. first_replicated_level), `replicated` = the vfem_mg_create_partial hierarchy of the
 * whole grid that every rank cycles identically from that level down.  All work vectors are the caller's (device memory):
 * per level the iterate / right-hand side / residual of the local node grid, the replicated level's two vectors, two PCG vectors of
 * the local fine grid and 8 doubles of scalars.  The two operations only the caller can do are callbacks, invoked on the calling
 * thread with the stream's work enqueued so far ordered before them:
 *   halo(user, level, field, left, right, phase): refresh the ghost planes of `field` (one of the vectors handed in) from the
 *        left / right neighbour; phase 0 = exchange and return, 1 = start (the interior planes are swept meanwhile), 2 = finish;
 *   allreduce(user, buf, n): sum n doubles at `buf` (device) over the ranks, in place.
 * Both return 0 on success.  Operators must be current (vfem_mg_update_operators on both hierarchies, plus the caller's own
 * assembly of the replicated level for sharded densities). */
typedef struct {
    int64_t n_planes, plane_nodes;      /* local node planes of the level, nodes per plane */
    int64_t first_owned, last_owned;    /* local planes this rank computes (ghost planes lie outside) */
    int64_t xoffn;                      /* global plane index of local plane 0 */
    int32_t gl, gr;                     /* ghost planes towards the left / right neighbour (0 or 1) */
    double *x, *b, *r;                  /* work vectors [n_planes * plane_nodes][3] (r unused on the last level) */
} vfem_dist_level;
typedef int (*vfem_halo_fn)(void *user, int level, double *field, int left, int right, int phase);
typedef int (*vfem_allreduce_fn)(void *user, double *device_buffer, int64_t n);
int vfem_mg_pcg_slab(vfem_mg *local, vfem_mg *replicated, int first_replicated_level, const vfem_dist_level *levels_host, int rank,
                     int world, double *replicated_x, double *replicated_b, double *x, const double *b, double *work_d, double *work_Ad,
                     double *scalars8, int max_iter, double tol, int mg_iterations, int mg_smoothing_iterations, int full_multigrid,
                     int overlap_sweeps, vfem_halo_fn halo, vfem_allreduce_fn allreduce, void *callback_user,
                     vfem_residual_cb residual_cb, void *residual_user, int *iterations_out, double *relres_out, void *stream);

/* ---- generic path: every instantiation other than the tuned <1,1,1> one.  TensorProductSimulator<1,1> / <2,2> (2-D, plane
 * stress, ElasticityTensor.hh:100-133; the reference binds <1,1>, VoxelFEM.cc:226) and <2,2,2>; MultigridSolver of the same
 * degrees (MG.hh, templates generic in Degrees...).  Nodal fields [numNodes][N], node grid (p*ne+1) per axis, last axis
 * fastest; Dirichlet mask 1 byte per node (bit c = component c).  Same semantics as the vfem_sim_ / vfem_mg_ entry points of the
 * same name. */
int vfem_gsim_create(vfem_gsim **out, int dim, int degree, const double *bbox_min_host, const double *bbox_max_host,
                     const int64_t *nelems_host);
/* slab decomposition (no reference counterpart; the domain decomposition north_star asks for, 3-D only): a simulator whose
 * density / modulus arrays hold elem_extra_lo / _hi further element layers below / above the node grid along x.
 * vfem_gsim_set_densities then takes all stored layers (vfem_gsim_num_stored_elements values, x slowest); every other entry
 * point sees the elements of the node grid only. */
int vfem_gsim_create_padded(vfem_gsim **out, int dim, int degree, const double *bbox_min_host, const double *bbox_max_host,
                            const int64_t *nelems_host, int64_t elem_extra_lo, int64_t elem_extra_hi);
int64_t vfem_gsim_num_stored_elements(const vfem_gsim *sim);
int vfem_gsim_destroy(vfem_gsim *sim);
int64_t vfem_gsim_num_nodes(const vfem_gsim *sim);
int64_t vfem_gsim_num_elements(const vfem_gsim *sim);
int vfem_gsim_ke_size(const vfem_gsim *sim);                       /* N * (p+1)^N */
int vfem_gsim_set_isotropic(vfem_gsim *sim, double young, double poisson);
int vfem_gsim_set_simp(vfem_gsim *sim, double E0, double Emin, double gamma);
int vfem_gsim_k0(const vfem_gsim *sim, double *K0_host);           /* ke x ke row-major */
int vfem_gsim_set_dirichlet(vfem_gsim *sim, const uint8_t *mask_host, const double *values_host);
int vfem_gsim_set_densities(vfem_gsim *sim, const double *rho, void *stream);
int vfem_gsim_get_densities(const vfem_gsim *sim, double *rho, void *stream);
int vfem_gsim_set_option(vfem_gsim *sim, int key, int value);
int vfem_gsim_apply_k(const vfem_gsim *sim, const double *u, double *out, void *stream);              /* TPS.hh:905-952 */
int vfem_gsim_compliance_gradient(const vfem_gsim *sim, const double *u, double *g, void *stream);    /* TPS.hh:730-751 */
int vfem_gsim_compliance(const vfem_gsim *sim, const double *f, const double *u, double *value_host, void *stream);  /* 1/2 f.u */
int vfem_gmg_create(vfem_gmg **out, vfem_gsim *fine, int num_coarsening_levels);                      /* MG.hh:22-90 */
/* The same three pieces as vfem_mg_create_slab / vfem_mg_create_partial / vfem_mg_{export,import}_level_ke for the generic path
 * (degree-2 hexahedra over the GPUs of a node, BASELINE config 5).  A rank's local hierarchy keeps two ghost element layers
 * (2 p node planes) towards each neighbour on every level: the restriction to an interface node reaches 2 p - 1 fine planes
 * to either side.  vfem_slab_level.xshift here: local plane of level l-1 = 2 * (local plane of level l) + xshift (<= 0);
 * xparity must be 0 (slabs start at even global elements on every distributed level).  Element matrices are ke x ke doubles
 * per element (81 x 81 for degree 2). */
int vfem_gmg_create_slab(vfem_gmg **out, vfem_gsim *fine_local, int n_levels, const vfem_slab_level *levels_host,
                         const uint8_t *const *masks_host);
int vfem_gmg_create_partial(vfem_gmg **out, vfem_gsim *fine, int num_coarsening_levels, int first_active_level);
int vfem_gmg_smooth_colors(vfem_gmg *mg, int level, double *u, const double *b, int forward, int first, int count, void *stream);
int vfem_gmg_cycle_from_level(vfem_gmg *mg, int level, double *x, const double *b, int num_smoothing_steps, int fmg, void *stream);
int vfem_gmg_export_level_ke(vfem_gmg *mg, int level, int64_t child_first_layer, int64_t count_x, double *ke_out, void *stream);
/* import: into the first active level of a replicated hierarchy -- level `first_active_level` of vfem_gmg_create_partial, or
 * level 0 of an ordinary hierarchy created on the coarse grid itself (its level 0 then applies the imported matrices instead of
 * E_e K0; this is how the slab driver keeps every global object at the size of the first replicated level) */
int vfem_gmg_import_level_ke(vfem_gmg *mg, int level, const double *ke, void *stream);
int vfem_gmg_destroy(vfem_gmg *mg);
int vfem_gmg_num_levels(const vfem_gmg *mg);
int vfem_gmg_level_dims(const vfem_gmg *mg, int level, int64_t nelems_host[3]);
int64_t vfem_gmg_level_num_nodes(const vfem_gmg *mg, int level);
int vfem_gmg_level_dirichlet_mask(const vfem_gmg *mg, int level, uint8_t *mask_host);
int vfem_gmg_set_symmetric_gauss_seidel(vfem_gmg *mg, int symmetric);
int vfem_gmg_update_operators(vfem_gmg *mg, void *stream);                                            /* MG.hh:415-425 */
int vfem_gmg_apply_k(vfem_gmg *mg, int level, const double *u, double *out, void *stream);
int vfem_gmg_residual(vfem_gmg *mg, int level, const double *u, const double *b, double *r, void *stream);
int vfem_gmg_smooth(vfem_gmg *mg, int level, double *u, const double *b, int forward, void *stream);  /* MG.hh:285-340 */
int vfem_gmg_zero_dirichlet(vfem_gmg *mg, int level, double *u, void *stream);
int vfem_gmg_restrict(vfem_gmg *mg, int fine_level, const double *fine, double *coarse, void *stream);
int vfem_gmg_interpolate(vfem_gmg *mg, int fine_level, const double *coarse, double *fine, int accumulate, void *stream);
int vfem_gmg_solve(vfem_gmg *mg, double *x, const double *f, int num_steps, int num_smoothing_steps, int stiffness_updated,
                   int zero_dirichlet, int full_multigrid, void *stream);                             /* MG.hh:447-472 */
int vfem_gmg_pcg(vfem_gmg *mg, double *x, const double *b, int max_iter, double tol, int mg_iterations, int mg_smoothing_iterations,
                 int full_multigrid, vfem_residual_cb residual_cb, void *cb_user, int *iterations_out, double *relres_out,
                 void *stream);                                                                       /* MG.hh:679-732 */

/* ---- design-update path, element-grid arrays [n0][n1][n2] fp64 (2-D grids: n2 = 1) ----
 * SmoothingFilter apply / backprop (TopologyOptimizationFilter.hh:105-162; transpose != 0 => A^T), ProjectionFilter apply /
 * backprop (:55-79), mean for TotalVolumeConstraint (TopologyOptimizationConstraint.hh:21-34), and the OC candidate step
 * clip(x0 sqrt(dJ/(dc lambda)), max(x0-m,0), min(x0+m,1)) of OCOptimizer::step (OptimalityCriterion.hh:47-50). */
int vfem_box_filter(const int64_t n_host[3], int radius, const double *in, double *out, int transpose, void *stream);
int vfem_projection(int64_t n, double beta, const double *x, double *out, void *stream);
int vfem_projection_backprop(int64_t n, double beta, const double *g, const double *vars, double *out, void *stream);
int vfem_oc_candidate(int64_t n, const double *x0, const double *dJ, const double *dc, double lambda, double move, double *out,
                      void *stream);
int vfem_mean(int64_t n, const double *x, double *mean_host, void *stream);

/* ---- Fourier-feature MLP density field: networks.MLP (networks.py:128-185), out_features = 1 ----
 * n_layers counts Linear layers as the reference does: Linear(2 es, nn), (n_layers - 2) x Linear(nn, nn), Linear(nn, 1).
 * Weights are handed over as fp32 arrays (host or device memory) in torch layout ([out][in] row-major) and converted to fp16:
 *   B [es][3] (MLP.B, already multiplied by `scale`), W_first [nn][2 es], W_hidden [(n_layers-2)][nn][nn],
 *   biases [(n_layers-1)][nn] (first layer, then hidden layers), w_out [nn], b_out.
 * Requirements of the MFMA tiling: es % 32 == 0, nn % 32 == 0, nn <= 512. */
int vfem_mlp_create(vfem_mlp **out, int embedding_size, int n_neurons, int n_layers, int sigmoid_output);
int vfem_mlp_destroy(vfem_mlp *mlp);
/* options of one network: VFEM_MLP_OPT_BWD_TERMS = 3 (default; every product of the backward pass is hi hi + hi lo + lo hi of split
 * fp16 operands: the reference's fp32 autograd to rounding) or 1 (hi hi only in the weight-gradient GEMMs, three times fewer MFMAs) */
#define VFEM_MLP_OPT_BWD_TERMS 1
/* VFEM_MLP_OPT_KEEP_FIRST = 1: a reference-precision GRID forward keeps the first layer's activations (2 KB per voxel: 68.7 GB at
 * 512 x 256 x 256, at most 96 GB, dropped silently when the allocation fails), and vfem_mlp_backward_grid* of the same grid, voxel range
 * and weights start from them instead of recomputing the first layer (two thirds of the forward's products).  0 (default): nothing is kept.
 * Results are the same bit for bit. */
#define VFEM_MLP_OPT_KEEP_FIRST 2
int vfem_mlp_set_option(vfem_mlp *mlp, int key, int value);
int vfem_mlp_load_weights(vfem_mlp *mlp, const float *B, const float *W_first, const float *W_hidden, const float *biases,
                          const float *w_out, float b_out);
/* forward on an explicit coordinate list [nvox][3] fp32 (device); either output pointer may be NULL */
int vfem_mlp_forward(vfem_mlp *mlp, const float *coords, int64_t nvox, float *out_f32, double *out_f64, void *stream);
/* forward on the regular grid of utils.get_mgrid (utils.py:35-53): n[d] points linspace(lo[d], hi[d]) incl. both ends,
 * flattened z-fastest = the solver's element order (train_xdg.py:245-247, 287) */
int vfem_mlp_forward_grid(vfem_mlp *mlp, const int64_t n_host[3], const double lo_host[3], const double hi_host[3],
                          float *out_f32, double *out_f64, void *stream);
/* the same for the voxels [first_voxel, first_voxel + num_voxels) of that grid (flat index, z fastest); outputs are indexed from
 * the start of the range.  A rank of an x-slab decomposition evaluates its own planes with this (the field needs no exchange). */
int vfem_mlp_forward_grid_range(vfem_mlp *mlp, const int64_t n_host[3], const double lo_host[3], const double hi_host[3],
                                int64_t first_voxel, int64_t num_voxels, float *out_f32, double *out_f64, void *stream);
/* The same two forwards in the reference's own arithmetic: fp32 features with accurate sin / cos of 2 pi x . B, fp32 GEMMs
 * (networks.py:170-185 runs in torch's default dtype).  Parity mode: several times slower than the fused fp16-operand kernel,
 * which moves the densities by ~1e-3 relative. */
int vfem_mlp_forward_f32(vfem_mlp *mlp, const float *coords, int64_t nvox, float *out_f32, double *out_f64, void *stream);
int vfem_mlp_forward_grid_range_f32(vfem_mlp *mlp, const int64_t n_host[3], const double lo_host[3], const double hi_host[3],
                                    int64_t first_voxel, int64_t num_voxels, float *out_f32, double *out_f64, void *stream);
/* Training (what torch.autograd does for networks.MLP in train_xdg.py:282-329): gradients of a scalar loss wrt the
 * parameters given g_out[v] = dL/d(out[v]) (device, fp32).  Outputs are overwritten, fp32, same layouts as vfem_mlp_load_weights:
 * dW1 [nn][2 es], dWh [n_layers-2][nn][nn], dbias [n_layers-1][nn], dwout [nn], dbout [1].  Split fp16 operands (hi + lo) on the
 * matrix pipe with fp32 accumulation = the reference's fp32 autograd to rounding; no library GEMM, the first layer's Fourier
 * features are regenerated inside the weight-gradient kernel.  `loss_scale` multiplies g_out before it enters fp16 and is
 * divided out of the results (power of two recommended). */
int vfem_mlp_backward(vfem_mlp *mlp, const float *coords, int64_t nvox, const float *g_out, float loss_scale, float *dW1,
                      float *dWh, float *dbias, float *dwout, float *dbout, void *stream);
int vfem_mlp_backward_grid(vfem_mlp *mlp, const int64_t n_host[3], const double lo_host[3], const double hi_host[3],
                           const float *g_out, float loss_scale, float *dW1, float *dWh, float *dbias, float *dwout, float *dbout,
                           void *stream);
/* the same for the voxels [first_voxel, first_voxel + num_voxels) of the grid (g_out indexed from the start of the range): the
 * partial gradient a rank of the slab decomposition contributes; the ranks' results are summed by one all-reduce */
int vfem_mlp_backward_grid_range(vfem_mlp *mlp, const int64_t n_host[3], const double lo_host[3], const double hi_host[3],
                                 int64_t first_voxel, int64_t num_voxels, const float *g_out, float loss_scale, float *dW1,
                                 float *dWh, float *dbias, float *dwout, float *dbout, void *stream);
/* torch.optim.Adam update (amsgrad off, weight_decay 0) of one fp32 parameter tensor, in place; step counts from 1 */
int vfem_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float lr, float beta1,
                   float beta2, float eps, int step, void *stream);

/* ---- timers: BENCHMARK_* registry (MeshFEM GlobalBenchmark.hh / Timer.hh; VoxelFEM.cc:245-255) ---- */
int vfem_timers_reset(void);
int vfem_timers_report(char *buf_host, size_t buf_len);

#ifdef __cplusplus
}
#endif
#endif /* VFEM_H */
