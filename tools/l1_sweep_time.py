"""Level-1 Gauss-Seidel sweep and residual time (event-timed, with checksums):  [VFEM_LIB=...] python tools/l1_sweep_time.py [n ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
for n in [int(a) for a in sys.argv[1:]] or [512, 256]:
    tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = tps.multigridSolver(6 if n >= 512 else 5)
    mg.updateElementStiffnessMatrices()
    nn = mg._nn(1)
    u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    for rep in range(4):
        _lib.check(lib.vfem_mg_smooth(mg._h, 1, _ptr(u), _ptr(b), rep & 1, _stream()))
    torch.cuda.synchronize()
    best = 1e9
    for trial in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(8):
            _lib.check(lib.vfem_mg_smooth(mg._h, 1, _ptr(u), _ptr(b), rep & 1, _stream()))
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8)
    print("%s n=%d: level-1 sweep %.3f ms   checksum %.12e" % (os.path.basename(os.environ.get("VFEM_LIB", "libvfem.so")), n, best, float(u.abs().sum())), flush=True)
    del mg, tps, u, b
    torch.cuda.empty_cache()
