"""ad-hoc perf probe (not the bench contract): apply GVoxel/s and PCG timing on the GPU."""
import sys, time
import numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import BC_CANTILEVER, make_hip, seeded_density

def t_apply(n, variant, reps=10):
    ne = (n, n, n)
    t = make_hip(ne, ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
    rho = torch.rand(t.numElements(), dtype=torch.float64, device='cuda')
    t.setElementDensities(rho)
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device='cuda')
    for _ in range(2): t.applyK_device(u, variant)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = t.applyK_device(u, variant)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    nb = 2 * t.numNodes() * 24 + t.numElements() * 8
    print(f"apply n={n} variant={variant}: {dt*1e3:.3f} ms  {t.numElements()/dt/1e9:.2f} GVoxel/s  {nb/dt/1e12:.3f} TB/s alg", flush=True)

def t_pcg(ne, dom, levels):
    rho = seeded_density(ne, 88, 'proxy')
    t = make_hip(ne, dom, BC_CANTILEVER, rho)
    mg = t.multigridSolver(levels)
    f = t.buildLoadVector_device()
    x0 = torch.zeros_like(f)
    mg.preconditionedConjugateGradient_device(x0, f, 2, 1e-4, None, 1, 2, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u = mg.preconditionedConjugateGradient_device(x0, f, 100, 1e-4, None, 1, 2, True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"pcg {ne} L={levels}: {mg.last_iterations} its, {dt:.3f} s, {mg.last_iterations/dt:.2f} it/s, relres {mg.last_relative_residual:.2e}, compliance {float((f*u).sum()):.6f}", flush=True)

if __name__ == '__main__':
    for n in (128, 256):
        for v in (0, 2, 1): t_apply(n, v)
    t_apply(512, 0, 5); t_apply(512, 2, 5)
    sys.exit(0)
    t_pcg((128, 64, 64), ([0, 0, 0], [2, 1, 1]), 3)
    t_pcg((256, 128, 128), ([0, 0, 0], [2, 1, 1]), 4)
