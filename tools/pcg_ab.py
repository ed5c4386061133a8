"""Same-box A/B of one simulator option on the PCG solve (operator update included):  python tools/pcg_ab.py KEY V0 V1 [n ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
key, v0, v1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for n in [int(a) for a in sys.argv[4:]] or [256, 512]:
    tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = tps.multigridSolver(6 if n >= 512 else 5)
    f = tps.buildLoadVector_device()
    for value in (v0, v1, v0, v1, v0, v1):
        set_knob(tps, key, value)
        x0 = torch.zeros_like(f)
        mg.preconditionedConjugateGradient_device(x0, f, 1, 1e-4, None, 1, 2, True)
        tps.setElementDensities(tps.getDensities_device())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u = mg.preconditionedConjugateGradient_device(x0, f, 100, 1e-4, None, 1, 2, True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("n=%d key %d = %d: iterations %d  %.4f s  %.2f it/s  compliance %.13f" % (n, key, value, mg.last_iterations, dt, mg.last_iterations / dt, float((f * u).sum())), flush=True)
    set_knob(tps, key, v1)
    del mg, tps, f, u, x0
    torch.cuda.empty_cache()
