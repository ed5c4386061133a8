"""Level-0 Gauss-Seidel sweep time (event-timed, marching kernel forced):  [VFEM_LIB=...] python tools/gs_sweep_time.py n [n ...]"""
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
    _lib.check(lib.vfem_sim_set_option(tps._h, 19, 2))
    mg = tps.multigridSolver(0)
    nn = mg._nn(0)
    u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    for rep in range(3):
        _lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(u), _ptr(b), rep % 2, 2, _stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 6
    e0.record()
    for rep in range(reps):
        _lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(u), _ptr(b), rep % 2, 2, _stream()))
    e1.record()
    torch.cuda.synchronize()
    print("%s n=%d: %.3f ms per sweep   checksum %.12e" % (os.path.basename(os.environ.get("VFEM_LIB", "libvfem.so")), n, e0.elapsed_time(e1) / (2 * reps), float(u.double().abs().sum())), flush=True)
    del mg, tps, u, b
    torch.cuda.empty_cache()
