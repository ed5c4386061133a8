"""Drop-in for MeshFEM's ``benchmark`` python module (``reset / report / to_dict``, Timer.hh:190-204), backed by the
timers of libvfem and of ndr_amd.pyVoxelFEM."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ndr_amd.pyVoxelFEM import benchmark_report as report  # noqa: E402,F401
from ndr_amd.pyVoxelFEM import benchmark_reset as reset  # noqa: E402,F401
from ndr_amd.pyVoxelFEM import benchmark_to_dict as to_dict  # noqa: E402,F401
