"""Level 1 (degree 1): node rows per mirror class (VFEM_OPT_L1_MERGED = 1, kernels_l1_merged.hip) against the per-element kernels (0):
agreement of sweeps (both directions) and residuals on small ragged grids, then sweep / residual times.
    python tools/l1_merged_probe.py [check] [n ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
args = sys.argv[1:]


def both(tps, mg, u, b, what):
    out = {}
    for merged in (0, 1, 2):
        _lib.check(lib.vfem_sim_set_option(tps._h, 22, merged))
        if what == "residual":
            r = torch.empty_like(u)
            _lib.check(lib.vfem_mg_residual(mg._h, 1, _ptr(u), _ptr(b), _ptr(r), _stream()))
        else:
            r = u.clone()
            _lib.check(lib.vfem_mg_smooth(mg._h, 1, _ptr(r), _ptr(b), 1 if what == "forward" else 0, _stream()))
        torch.cuda.synchronize()
        out[merged] = r
    return out


if "check" in args:
    args.remove("check")
    worst = 0.0
    for ne in ((8, 8, 8), (16, 8, 24), (24, 40, 8), (136, 24, 264), (40, 72, 200), (8, 24, 512), (24, 8, 504), (16, 16, 760)):
        tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
        g = torch.Generator(device="cuda").manual_seed(5)
        tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g) ** 3)
        mg = tps.multigridSolver(3 if min(ne) >= 24 else 2)
        mg.updateElementStiffnessMatrices()
        nn = mg._nn(1)
        u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
        b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
        for what in ("residual", "forward", "backward"):
            o = both(tps, mg, u, b, what)
            err = float((o[0] - o[1]).abs().max() / o[0].abs().max())
            worst = max(worst, err)
            assert float((o[1] - o[2]).abs().max()) == 0.0, "the z-pair launch must reproduce the two colour launches bit for bit"
            print("grid %s level-1 %s: relative max difference %.2e (z pairs = colour by colour bit for bit)" % (ne, what, err), flush=True)
    print("worst %.2e" % worst)
    assert worst < 1e-12

for n in [int(a) for a in args]:
    tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = tps.multigridSolver(6 if n >= 512 else 5)
    mg.updateElementStiffnessMatrices()
    nn = mg._nn(1)
    u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    for merged in (0, 1, 2):
        _lib.check(lib.vfem_sim_set_option(tps._h, 22, merged))
        x, r = u.clone(), torch.empty_like(u)
        _lib.check(lib.vfem_mg_smooth(mg._h, 1, _ptr(x), _ptr(b), 1, _stream()))
        _lib.check(lib.vfem_mg_residual(mg._h, 1, _ptr(u), _ptr(b), _ptr(r), _stream()))
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            _lib.check(lib.vfem_mg_smooth(mg._h, 1, _ptr(x), _ptr(b), rep & 1, _stream()))
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        t0 = time.perf_counter()
        for rep in range(4):
            _lib.check(lib.vfem_mg_residual(mg._h, 1, _ptr(u), _ptr(b), _ptr(r), _stream()))
        torch.cuda.synchronize()
        t_res = (time.perf_counter() - t0) / 4
        print("n %d level 1 (%d nodes) %-11s: sweep %.3f ms, residual %.3f ms" % (n, nn, ("per element", "per class", "per class, z pairs")[merged], best * 1e3, t_res * 1e3), flush=True)
    del tps, mg, u, b
    torch.cuda.empty_cache()
