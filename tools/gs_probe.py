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
if level in (0, 1):
    base = None
    for pair in (1, 0, 1, 0):
        set_knob(tps, 10, pair)
        for fwd in (1, 0):
            for rep in range(3):
                uu = u.clone()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                lib.vfem_mg_smooth(mg._h, level, _ptr(uu), _ptr(b), fwd, _stream())
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            if base is None: base = {}
            base.setdefault(fwd, uu)
            print("level %d fused z-colour pairs %d forward %d: %.3f ms per sweep   max |diff| to first: %.3e" % (level, pair, fwd, dt * 1e3, float((uu - base[fwd]).abs().max())), flush=True)
    set_knob(tps, 10, 1)
res = {}
for variant in (0, 2, 1):
    set_knob(tps, 2, variant)
    for rep in range(3):
        uu = u.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.vfem_mg_smooth(mg._h, level, _ptr(uu), _ptr(b), 1, _stream())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res[variant] = uu
    print("level %d variant %d: %.3f ms per sweep   max |diff| to variant 0: %.3e" % (level, variant, dt * 1e3, float((uu - res[0]).abs().max())), flush=True)
set_knob(tps, 2, 0)
