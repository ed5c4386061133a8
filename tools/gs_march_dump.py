"""one forward level-0 sweep (marching kernel forced) on a small grid, result dumped:  [VFEM_LIB=...] python tools/gs_march_dump.py out.npy nx ny nz [forward]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_hip, BC_CANTILEVER
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
out, ne = sys.argv[1], tuple(int(a) for a in sys.argv[2:5])
fwd = int(sys.argv[5]) if len(sys.argv) > 5 else 1
tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER if os.environ.get("BC") else None, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
_lib.check(lib.vfem_sim_set_option(tps._h, 19, int(os.environ.get("MARCH", "2"))))
mg = tps.multigridSolver(0)
nn = mg._nn(0)
u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
_lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(u), _ptr(b), fwd, 1, _stream()))
torch.cuda.synchronize()
np.save(out, u.cpu().numpy().reshape(ne[0] + 1, ne[1] + 1, ne[2] + 1, 3))
