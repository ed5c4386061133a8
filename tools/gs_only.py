"""A few level-l Gauss-Seidel sweeps and nothing else, for rocprofv3:  python3 tools/gs_only.py n level reps [key=value ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
from _knobs import set_knob
lib = _lib.load()
n, level, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    set_knob(tps, int(k), int(v))
mg = tps.multigridSolver(6 if n >= 512 else 5)
mg.updateElementStiffnessMatrices()
nn = mg._nn(level)
u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
for rep in range(reps):
    _lib.check(lib.vfem_mg_smooth(mg._h, level, _ptr(u), _ptr(b), 1, _stream()))
torch.cuda.synchronize()
print("done")
