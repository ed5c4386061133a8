"""Marching level-0 sweeps and nothing else, for rocprofv3:  python3 tools/gs_march_only.py n reps [march=1] [chunks=0]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
n, reps = int(sys.argv[1]), int(sys.argv[2])
march = int(sys.argv[3]) if len(sys.argv) > 3 else 1
chunks = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
_lib.check(lib.vfem_sim_set_option(tps._h, 19, 2 * march))
_lib.check(lib.vfem_sim_set_option(tps._h, 20, chunks))
mg = tps.multigridSolver(0)
nn = mg._nn(0)
u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
for rep in range(reps):
    _lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(u), _ptr(b), rep % 2, 2, _stream()))
torch.cuda.synchronize()
print("done")
