import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ne = (n, n, n)
tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
mg = tps.multigridSolver(levels)
f = tps.buildLoadVector_device()
x0 = torch.zeros_like(f)
torch.cuda.synchronize(); t0 = time.perf_counter()
u = mg.preconditionedConjugateGradient_device(x0, f, 100, 1e-4, None, 1, 2, True)
torch.cuda.synchronize()
print("iterations", mg.last_iterations, "seconds", time.perf_counter() - t0, flush=True)
