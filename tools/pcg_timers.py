"""Host-side timers of one 512^3 PCG solve (the library's ScopedTimer sections): share of the per-solve operator update."""
import ctypes, os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
levels = int(sys.argv[2]) if len(sys.argv) > 2 else (6 if n >= 512 else 5)
tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
mg = tps.multigridSolver(levels)
f = tps.buildLoadVector_device()
x0 = torch.zeros_like(f)
mg.preconditionedConjugateGradient_device(x0, f, 1, 1e-4, None, 1, 2, True)
torch.cuda.synchronize()
lib.vfem_timers_reset()
t0 = time.perf_counter()
mg.preconditionedConjugateGradient_device(x0, f, 100, 1e-4, None, 1, 2, True)
torch.cuda.synchronize()
print("solve %.3f s, %d iterations" % (time.perf_counter() - t0, mg.last_iterations))
buf = ctypes.create_string_buffer(1 << 16)
lib.vfem_timers_report(buf, len(buf))
print(buf.value.decode())
