"""ctypes binding of libvfem.so (the C ABI declared in include/vfem.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is visible,
every compute entry point raises.  ``load()`` itself only needs the shared object (so the
symbol/export checks run on a CPU-only box).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VFEM_LIB: another build of the same library (tools/ use the `make ablation` build for their timing experiments)
LIB_PATH = os.environ.get("VFEM_LIB") or os.path.join(_HERE, "csrc", "libvfem.so")
_lib = None

RESIDUAL_CB = ctypes.CFUNCTYPE(None, c_void_p, c_int, c_double)

# name -> (restype, argtypes); kept in one table so tests can check it against include/vfem.h
SIGNATURES = {
    "vfem_last_error": (c_char_p, []),
    "vfem_device_count": (c_int, []),
    "vfem_set_device": (c_int, [c_int]),
    "vfem_version": (c_int, []),
    "vfem_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "vfem_free": (c_int, [c_void_p]),
    "vfem_copy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "vfem_copy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "vfem_copy_d2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "vfem_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "vfem_stream_sync": (c_int, [c_void_p]),
    "vfem_sim_create": (c_int, [POINTER(c_void_p), POINTER(c_double), POINTER(c_double), POINTER(c_int64)]),
    "vfem_sim_destroy": (c_int, [c_void_p]),
    "vfem_sim_num_nodes": (c_int64, [c_void_p]),
    "vfem_sim_num_elements": (c_int64, [c_void_p]),
    "vfem_sim_set_isotropic": (c_int, [c_void_p, c_double, c_double]),
    "vfem_sim_set_simp": (c_int, [c_void_p, c_double, c_double, c_double]),
    "vfem_sim_set_option": (c_int, [c_void_p, c_int, c_int]),
    "vfem_sim_k0": (c_int, [c_void_p, c_void_p]),
    "vfem_sim_set_dirichlet": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vfem_sim_set_loads": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vfem_sim_build_load_vector": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vfem_sim_set_densities": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vfem_sim_set_uniform_density": (c_int, [c_void_p, c_double, c_void_p]),
    "vfem_sim_get_densities": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vfem_sim_apply_k": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vfem_sim_apply_k_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "vfem_sim_compliance_gradient": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_compliance": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_double), c_void_p]),
    "vfem_mg_create": (c_int, [POINTER(c_void_p), c_void_p, c_int]),
    "vfem_mg_destroy": (c_int, [c_void_p]),
    "vfem_sim_set_next_element_padding": (c_int, [c_int64, c_int64]),
    "vfem_sim_num_stored_elements": (c_int64, [c_void_p]),
    "vfem_mg_create_slab": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p, POINTER(c_void_p)]),
    "vfem_mg_create_partial": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int]),
    "vfem_mg_smooth_colors": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vfem_mg_can_smooth_planes": (c_int, [c_void_p, c_int]),
    "vfem_mg_smooth_group_planes": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p]),
    "vfem_mg_cycle_from_level": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vfem_mg_num_levels": (c_int, [c_void_p]),
    "vfem_mg_level_dims": (c_int, [c_void_p, c_int, POINTER(c_int64)]),
    "vfem_mg_level_num_nodes": (c_int64, [c_void_p, c_int]),
    "vfem_mg_level_dirichlet_mask": (c_int, [c_void_p, c_int, c_void_p]),
    "vfem_mg_set_symmetric_gauss_seidel": (c_int, [c_void_p, c_int]),
    "vfem_mg_field_ptr": (c_void_p, [c_void_p, c_int, c_int]),
    "vfem_mg_update_operators": (c_int, [c_void_p, c_void_p]),
    "vfem_mg_export_level_ke": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "vfem_mg_import_level_ke": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "vfem_mg_apply_k": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "vfem_mg_residual": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_mg_smooth": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vfem_mg_smooth_sweeps": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vfem_mg_zero_dirichlet": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "vfem_mg_restrict": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "vfem_mg_interpolate": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vfem_mg_coarsest_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_dense_spd_inverse": (c_int, [c_int64, c_void_p, c_void_p]),
    "vfem_mg_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vfem_mg_pcg": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_int, c_int,
                            RESIDUAL_CB, c_void_p, POINTER(c_int), POINTER(c_double), c_void_p]),
    "vfem_mlp_forward_grid_range": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double), POINTER(c_double), c_int64, c_int64,
                                            c_void_p, c_void_p, c_void_p]),
    "vfem_mlp_forward_grid_range_f32": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double), POINTER(c_double), c_int64, c_int64,
                                                c_void_p, c_void_p, c_void_p]),
    "vfem_mlp_backward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_mlp_backward_grid": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double), POINTER(c_double), c_void_p, c_float,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_mlp_backward_grid_range": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double), POINTER(c_double), c_int64, c_int64, c_void_p,
                                             c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_adam_step": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "vfem_gsim_create": (c_int, [POINTER(c_void_p), c_int, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_int64)]),
    "vfem_gsim_create_padded": (c_int, [POINTER(c_void_p), c_int, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_int64),
                                        c_int64, c_int64]),
    "vfem_gsim_num_stored_elements": (c_int64, [c_void_p]),
    "vfem_gmg_create_slab": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p, POINTER(c_void_p)]),
    "vfem_gmg_create_partial": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int]),
    "vfem_gmg_smooth_colors": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vfem_gmg_cycle_from_level": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vfem_gmg_export_level_ke": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "vfem_gmg_import_level_ke": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "vfem_gsim_destroy": (c_int, [c_void_p]),
    "vfem_gsim_num_nodes": (c_int64, [c_void_p]),
    "vfem_gsim_num_elements": (c_int64, [c_void_p]),
    "vfem_gsim_ke_size": (c_int, [c_void_p]),
    "vfem_gsim_set_isotropic": (c_int, [c_void_p, c_double, c_double]),
    "vfem_gsim_set_simp": (c_int, [c_void_p, c_double, c_double, c_double]),
    "vfem_gsim_k0": (c_int, [c_void_p, c_void_p]),
    "vfem_gsim_set_dirichlet": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vfem_gsim_set_densities": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vfem_gsim_get_densities": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vfem_gsim_set_option": (c_int, [c_void_p, c_int, c_int]),
    "vfem_gsim_apply_k": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_gsim_compliance_gradient": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_gsim_compliance": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_double), c_void_p]),
    "vfem_gmg_create": (c_int, [POINTER(c_void_p), c_void_p, c_int]),
    "vfem_gmg_destroy": (c_int, [c_void_p]),
    "vfem_gmg_num_levels": (c_int, [c_void_p]),
    "vfem_gmg_level_dims": (c_int, [c_void_p, c_int, POINTER(c_int64)]),
    "vfem_gmg_level_num_nodes": (c_int64, [c_void_p, c_int]),
    "vfem_gmg_level_dirichlet_mask": (c_int, [c_void_p, c_int, c_void_p]),
    "vfem_gmg_set_symmetric_gauss_seidel": (c_int, [c_void_p, c_int]),
    "vfem_gmg_update_operators": (c_int, [c_void_p, c_void_p]),
    "vfem_gmg_apply_k": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "vfem_gmg_residual": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_gmg_smooth": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vfem_gmg_zero_dirichlet": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "vfem_gmg_restrict": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "vfem_gmg_interpolate": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vfem_gmg_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vfem_gmg_pcg": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_int, c_int, RESIDUAL_CB, c_void_p,
                             POINTER(c_int), POINTER(c_double), c_void_p]),
    "vfem_box_filter": (c_int, [POINTER(c_int64), c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vfem_projection": (c_int, [c_int64, c_double, c_void_p, c_void_p, c_void_p]),
    "vfem_projection_backprop": (c_int, [c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfem_oc_candidate": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_double, c_double, c_void_p, c_void_p]),
    "vfem_mean": (c_int, [c_int64, c_void_p, POINTER(c_double), c_void_p]),
    "vfem_mg_pcg_slab": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_double, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, RESIDUAL_CB, c_void_p,
                                 POINTER(c_int), POINTER(c_double), c_void_p]),
    "vfem_mlp_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int]),
    "vfem_mlp_destroy": (c_int, [c_void_p]),
    "vfem_mlp_set_option": (c_int, [c_void_p, c_int, c_int]),
    "vfem_mlp_load_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float]),
    "vfem_mlp_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "vfem_mlp_forward_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "vfem_mlp_forward_grid": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double), POINTER(c_double), c_void_p,
                                      c_void_p, c_void_p]),
    "vfem_timers_reset": (c_int, []),
    "vfem_timers_report": (c_int, [c_char_p, c_size_t]),
}


def load():
    """Load libvfem.so; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvfem.so is missing (%s): build it with `make -C ndr_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if hasattr(lib, "vfem_debug_set"):          # ablation build only (wrong-result timing variants for tools/)
        lib.vfem_debug_set.restype = c_int
        lib.vfem_debug_set.argtypes = [c_int, c_int]
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise RuntimeError(load().vfem_last_error().decode("utf-8", "replace"))


_gpu_checked = False


def require_gpu():
    """Fail loudly when the HIP path cannot run (no silent CPU fallback)."""
    global _gpu_checked
    if _gpu_checked:
        return
    lib = load()
    if lib.vfem_device_count() < 1:
        raise RuntimeError("ndr_amd: no HIP device visible; the voxel-FEM path runs only on the GPU")
    _gpu_checked = True
