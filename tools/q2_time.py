import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd import _lib, pyVoxelFEM as pv
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ne = (n, n, n)
t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1, 1, 1]), list(ne))
t.E_min = 1e-4
g = torch.Generator(device="cuda").manual_seed(88)
t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
res = {}
for impl in (0, 1):
    set_knob(t, 6, impl)
    out = t.applyK_device(u); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): out = t.applyK_device(u)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    res[impl] = out
    ab = 2 * t.numNodes() * 24 + t.numElements() * 8
    print("impl %d: %.3f ms  %.2f GVoxel/s  algorithmic %.0f GB/s (%.3f of 8 TB/s)" % (impl, dt * 1e3, t.numElements() / dt / 1e9, ab / dt / 1e9, ab / dt / 8e12), flush=True)
print("max rel diff", float((res[0] - res[1]).abs().max() / res[1].abs().max()))
set_knob(t, 6, 0)
