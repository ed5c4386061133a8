"""Level-0 / level-1 Gauss-Seidel sweep: time per sweep and bitwise agreement of the implementation choices.
   python tools/gs_probe.py [n=256] [level=0]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ne = (n, n, n)
tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
mg = tps.multigridSolver(6 if n >= 512 else 5)
mg.updateElementStiffnessMatrices()
nn = mg._nn(level)
u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
from ndr_amd.pyVoxelFEM import _ptr, _stream

def sweep(fwd, reps=4):
    best = 1e9
    for rep in range(reps):
        uu = u.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(lib.vfem_mg_smooth(mg._h, level, _ptr(uu), _ptr(b), fwd, _stream()))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return uu, best

# (name, [(key, value), ...]); the first entry is the reference the others are compared with bit for bit
configs = [("production", []), ("plain gather sweeps", [(2, 1)])]
if level == 0:
    configs += [("coefficient table (no resident K0)", [(13, 0)]), ("one launch per colour", [(10, 0)])]
if level == 1:
    configs += [("precomputed diagonal blocks", [(12, 1)])]
base = {}
for name, knobs in configs + [configs[0]]:
    for k, v in knobs:
        set_knob(tps, k, v)
    for fwd in (1, 0):
        uu, dt = sweep(fwd)
        base.setdefault(fwd, uu)
        print("level %d %-36s forward %d: %.3f ms per sweep   max |diff| to production: %.3e" %
              (level, name, fwd, dt * 1e3, float((uu - base[fwd]).abs().max())), flush=True)
    for k, v in knobs:
        set_knob(tps, k, {2: 0, 13: 1, 10: 1, 12: 0}[k])
