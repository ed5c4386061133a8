import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
lib = _lib.load()
for n, levels in ((256, 5), (512, 6)):
    ne = (n, n, n)
    tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = tps.multigridSolver(levels)
    f = tps.buildLoadVector_device()
    for variant in [int(a) for a in sys.argv[1:]] or [0, 1]:
        set_knob(tps, 2, variant)
        x0 = torch.zeros_like(f)
        mg.preconditionedConjugateGradient_device(x0, f, 1, 1e-4, None, 1, 2, True)
        tps.setElementDensities(tps.getDensities_device())          # the timed solve rebuilds the coarse operators (as bench.py)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u = mg.preconditionedConjugateGradient_device(x0, f, 100, 1e-4, None, 1, 2, True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("n=%d gs_variant=%d iterations %d  %.3f s  %.2f it/s  compliance %.10f" % (n, variant, mg.last_iterations, dt, mg.last_iterations / dt, float((f * u).sum())), flush=True)
    set_knob(tps, 2, 0)
    del mg, tps, f, u, x0
    torch.cuda.empty_cache()
