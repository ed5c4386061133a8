"""Level-0 residual b - K u (0 at fixed components) by the LDS-DMA kernel (VFEM_OPT_APPLY_IMPL = 0) against the register-staged kernel (1):
agreement on ragged grids, then times.   python tools/residual_dma_probe.py [n ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()


def residual(tps, mg, u, b, impl):
    _lib.check(lib.vfem_sim_set_option(tps._h, 4, impl))
    r = torch.empty_like(u)
    _lib.check(lib.vfem_mg_residual(mg._h, 0, _ptr(u), _ptr(b), _ptr(r), _stream()))
    torch.cuda.synchronize()
    return r


worst = 0.0
for ne in ((8, 8, 8), (16, 8, 24), (24, 40, 8), (72, 24, 136), (40, 72, 200), (16, 24, 512), (16, 16, 760)):
    tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(5)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g) ** 3)
    mg = tps.multigridSolver(3 if min(ne) >= 16 else 1)
    mg.updateElementStiffnessMatrices()
    nn = mg._nn(0)
    u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    r0, r1 = residual(tps, mg, u, b, 0), residual(tps, mg, u, b, 1)
    err = float((r0 - r1).abs().max() / r1.abs().max())
    fixed = torch.as_tensor(tps.dirichletMask, device="cuda")
    assert float(r0[fixed].abs().max()) == 0.0
    worst = max(worst, err)
    print("grid %s: relative max difference %.2e" % (ne, err), flush=True)
assert worst < 1e-13, worst
for n in [int(a) for a in sys.argv[1:]]:
    tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = tps.multigridSolver(5)
    nn = mg._nn(0)
    u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    for impl in (1, 0):
        residual(tps, mg, u, b, impl)
        t0 = time.perf_counter()
        for rep in range(5):
            residual(tps, mg, u, b, impl)
        print("n %d residual by the %s kernel: %.3f ms" % (n, ("LDS-DMA", "register-staged")[impl], (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    del tps, mg, u, b
    torch.cuda.empty_cache()
