"""Degree-2 apply: marching kernel (impl 0) against the pencil kernel (2) and the dense gather kernel (1)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd import _lib, pyVoxelFEM as pv
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
lib = _lib.load()

def run(ne, impls, reps):
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1, 1, 1]), list(ne))
    t.E_min = 1e-4
    g = torch.Generator(device="cuda").manual_seed(88)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    res = {}
    for impl in impls:
        set_knob(t, 6, impl)
        for _ in range(3):
            out = t.applyK_device(u)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = t.applyK_device(u)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        res[impl] = out.clone() if len(impls) > 1 else None
        ab = 2 * t.numNodes() * 24 + t.numElements() * 8
        print("%s impl %d: %.3f ms  %.2f GVoxel/s  algorithmic %.0f GB/s (%.3f of 8 TB/s)" % (ne, impl, dt * 1e3, t.numElements() / dt / 1e9, ab / dt / 1e9, ab / dt / 8e12), flush=True)
        del out
    set_knob(t, 6, 0)
    if len(impls) > 1:
        ref = res[impls[0]]
        for impl in impls[1:]:
            print("   max rel diff impl %d vs %d: %.2e" % (impl, impls[0], float((res[impl] - ref).abs().max() / ref.abs().max())), flush=True)

if sys.argv[1] == "check":
    for ne in ((3, 2, 5), (4, 4, 4), (5, 7, 64), (17, 6, 63), (16, 5, 130), (33, 9, 70), (70, 3, 3)):
        run(ne, (1, 2, 0), 1)
elif sys.argv[1] == "exp":
    n = int(sys.argv[2])
    for exp in (0, 1, 2, 3, 0):
        set_knob(None, 1, exp)
        print("ablation", exp, {0: "production", 1: "no element product", 2: "no row stores", 3: "no row loads"}[exp])
        run((n, n, n), (0,), 5)
    set_knob(None, 1, 0)
else:
    n = int(sys.argv[1])
    run((n, n, n), (0,) if len(sys.argv) > 2 else (2, 0), 5)
