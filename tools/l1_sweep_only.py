"""A few level-1 sweeps in the mode given (VFEM_OPT_L1_MERGED value) for rocprofv3:  python tools/l1_sweep_only.py n mode [sweeps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
n, mode = int(sys.argv[1]), int(sys.argv[2])
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
mg = tps.multigridSolver(6 if n >= 512 else 5)
mg.updateElementStiffnessMatrices()
nn = mg._nn(1)
u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
_lib.check(lib.vfem_sim_set_option(tps._h, 22, mode))
for rep in range(sweeps):
    _lib.check(lib.vfem_mg_smooth(mg._h, 1, _ptr(u), _ptr(b), rep & 1, _stream()))
torch.cuda.synchronize()
print("done")
