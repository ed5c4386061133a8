"""Level 1 (degree 1): the virtual Galerkin operator (sum_f E_f cK0[f] at every visit) against the same operator stored as a 27-point
block stencil (VFEM_OPT_L1_STORED, 1944 B per node): sweep and residual times, agreement.   python tools/l1_stored_probe.py [n ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
for n in [int(a) for a in sys.argv[1:]] or [256, 512]:
    tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = tps.multigridSolver(6 if n >= 512 else 5)
    nn = mg._nn(1)
    u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    res = {}
    for stored in (0, 1, 2):
        _lib.check(lib.vfem_sim_set_option(tps._h, 21, stored))
        t0 = time.perf_counter()
        mg.updateElementStiffnessMatrices()
        torch.cuda.synchronize()
        t_up = time.perf_counter() - t0
        x = u.clone()
        _lib.check(lib.vfem_mg_smooth(mg._h, 1, _ptr(x), _ptr(b), 1, _stream()))
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            _lib.check(lib.vfem_mg_smooth(mg._h, 1, _ptr(x), _ptr(b), rep & 1, _stream()))
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        x = u.clone()
        _lib.check(lib.vfem_mg_smooth(mg._h, 1, _ptr(x), _ptr(b), 1, _stream()))
        r = torch.empty_like(u)
        _lib.check(lib.vfem_mg_residual(mg._h, 1, _ptr(u), _ptr(b), _ptr(r), _stream()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(3):
            _lib.check(lib.vfem_mg_residual(mg._h, 1, _ptr(u), _ptr(b), _ptr(r), _stream()))
        torch.cuda.synchronize()
        t_res = (time.perf_counter() - t0) / 3
        res[stored] = (x, r.clone())
        print("n %d level 1 (%d nodes) %-8s: operator update %.1f ms, sweep %.3f ms, residual %.3f ms" %
              (n, nn, ("virtual", "stored", "half")[stored], t_up * 1e3, best * 1e3, t_res * 1e3), flush=True)
    for other in (1, 2):
        dx = float((res[0][0] - res[other][0]).abs().max() / res[0][0].abs().max())
        dr = float((res[0][1] - res[other][1]).abs().max() / res[0][1].abs().max())
        print("n %d: %s vs virtual, one sweep %.2e, residual %.2e (relative max)" % (n, ("", "stored", "half")[other], dx, dr), flush=True)
    _lib.check(lib.vfem_sim_set_option(tps._h, 21, 0))
    del tps, mg, u, b, res
    torch.cuda.empty_cache()
