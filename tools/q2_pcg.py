import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, MATERIAL
from ndr_amd import pyVoxelFEM as pv
for n, levels in [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else ((64, 4), (128, 5)):
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [2, 1, 1]), [n, n, n])
    t.readMaterial(MATERIAL); t.applyDisplacementsAndLoadsFromFile(BC_CANTILEVER); t.E_min = 1e-4
    g = torch.Generator(device="cuda").manual_seed(88)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = t.multigridSolver(levels)
    f = t.buildLoadVector_device()
    mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, 1, 1e-4, None, 1, 2, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u = mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, 100, 1e-4, None, 1, 2, True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("Q2 %d^3 (%d nodes) levels %d: %d iterations, %.2f s, %.2f it/s, relres %.2e, mem %.1f GB" % (n, t.numNodes(), levels, mg.last_iterations, dt, mg.last_iterations / dt, mg.last_relative_residual, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del mg, t, f, u
    torch.cuda.empty_cache()
