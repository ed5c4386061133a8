"""Finest-level sweep of the degree-2 hierarchy: element-by-element gather (k_gs_q2_level0) against the neighbour-node order
(k_gs_q2_level0_nodes).  Time per sweep, agreement of the relaxed fields, CG-MG rate.   python tools/q2_gs0_probe.py N LEVELS"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import BC_CANTILEVER, MATERIAL
from ndr_amd import _lib, pyVoxelFEM as pv

n, levels = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.load()
t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [2, 1, 1]), [n, n, n])
t.readMaterial(MATERIAL); t.applyDisplacementsAndLoadsFromFile(BC_CANTILEVER); t.E_min = 1e-4
g = torch.Generator(device="cuda").manual_seed(88)
t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
mg = t.multigridSolver(levels)
mg.updateElementStiffnessMatrices()
nn0 = mg._nn(0)
u0 = torch.randn((nn0, 3), dtype=torch.float64, device="cuda", generator=g)
b = torch.randn((nn0, 3), dtype=torch.float64, device="cuda", generator=g)
f = t.buildLoadVector_device()
out, fields = {"grid": n, "levels": levels}, {}
for impl, name in ((0, "by element"), (1, "by neighbour node"), (2, "by neighbour node, rows through LDS")):
    _lib.check(lib.vfem_gsim_set_option(t._h, 16, impl))
    res = {}
    for fwd in (1, 0):
        u = u0.clone()
        _lib.check(lib.vfem_gmg_smooth(mg._h, 0, pv._ptr(u), pv._ptr(b), fwd, pv._stream()))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            _lib.check(lib.vfem_gmg_smooth(mg._h, 0, pv._ptr(u), pv._ptr(b), fwd, pv._stream()))
        torch.cuda.synchronize()
        res["sweep_ms_forward" if fwd else "sweep_ms_backward"] = (time.perf_counter() - t0) / 3 * 1e3
        fields[(impl, fwd)] = u
    mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, 1, 1e-4, None, 1, 2, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, 100, 1e-4, None, 1, 2, True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res.update({"pcg_iterations": mg.last_iterations, "iterations_per_s": mg.last_iterations / dt, "compliance": float((f * x).sum())})
    out[name] = res
    print(json.dumps(out), flush=True)
out["max_rel_diff_after_4_sweeps"] = max(float((fields[(0, w)] - fields[(m, w)]).abs().max() / fields[(0, w)].abs().max()) for w in (0, 1) for m in (1, 2))
print(json.dumps(out), flush=True)
