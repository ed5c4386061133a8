"""CG-MG rate against the number of waves sharing a level-1 node's element slots (VFEM_OPT_L1_SPLIT).  python tools/pcg_split.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
for n, levels in ((256, 5), (512, 6)):
    ne = (n, n, n)
    tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = tps.multigridSolver(levels)
    f = tps.buildLoadVector_device()
    for split in (1, 2, 4, 8):
        set_knob(tps, 15, split)
        x0 = torch.zeros_like(f)
        mg.preconditionedConjugateGradient_device(x0, f, 1, 1e-4, None, 1, 2, True)
        tps.setElementDensities(tps.getDensities_device())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u = mg.preconditionedConjugateGradient_device(x0, f, 100, 1e-4, None, 1, 2, True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("n=%d split=%d iterations %d  %.3f s  %.2f it/s  compliance %.10f" % (n, split, mg.last_iterations, dt, mg.last_iterations / dt, float((f * u).sum())), flush=True)
    del mg, tps, f, u, x0
    torch.cuda.empty_cache()
