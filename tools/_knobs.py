"""Knobs of the experiment scripts.  Choices between equivalent kernels are options of one simulator handle
(vfem_sim_set_option / vfem_gsim_set_option); the wrong-result timing ablations (keys 1, 3, 8) exist only in the
ablation build of the library:  make -C ndr_amd/csrc ablation && VFEM_LIB=ndr_amd/csrc/libvfem_ablation.so python tools/..."""
from ndr_amd import _lib

ABLATION_KEYS = (1, 3, 8)


def set_knob(sim, key, value):
    """sim: a ndr_amd.pyVoxelFEM simulator (any degree) or None for the process-wide ablation keys"""
    lib = _lib.load()
    if key in ABLATION_KEYS:
        if not hasattr(lib, "vfem_debug_set"):
            raise SystemExit("key %d is a timing ablation: build `make -C ndr_amd/csrc ablation` and set VFEM_LIB" % key)
        lib.vfem_debug_set(key, value)
    elif key == 6:
        _lib.check(lib.vfem_gsim_set_option(sim._h, key, value))
    else:
        _lib.check(lib.vfem_sim_set_option(sim._h, key, value))
