"""Level-0 Gauss-Seidel: the plane-resident marching half sweeps (kernels_gs_march.hip) against the row-streaming kernels.
   python tools/gs_march_probe.py check            agreement on a set of grid shapes (tile seams, ragged edges, tiny grids)
   python tools/gs_march_probe.py time [n ...]     ms per sweep, both kernels, and the chunk-count sweep of the marching one"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
GS_MARCH, GS_MARCH_CHUNKS, GS_MARCH_FORM = 19, 20, 23


def setup(ne, seed=88):
    tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER if ne[1] % 2 == 0 and ne[2] % 2 == 0 else None, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(seed)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = tps.multigridSolver(0)
    nn = mg._nn(0)
    u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    return tps, mg, u, b


def sweeps(tps, mg, u, b, march, seq, form=2):
    _lib.check(lib.vfem_sim_set_option(tps._h, GS_MARCH, 2 * march))
    _lib.check(lib.vfem_sim_set_option(tps._h, GS_MARCH_FORM, form))
    uu = u.clone()
    for fwd in seq:
        _lib.check(lib.vfem_mg_smooth(mg._h, 0, _ptr(uu), _ptr(b), fwd, _stream()))
    torch.cuda.synchronize()
    return uu


def check():
    worst = 0.0
    shapes = [(2, 2, 2), (3, 5, 4), (8, 8, 8), (9, 12, 58), (5, 13, 59), (6, 24, 116), (7, 25, 117), (4, 11, 57), (16, 30, 70), (33, 40, 130),
              (12, 64, 64), (10, 100, 20), (64, 64, 64)]
    for ne in shapes:
        tps, mg, u, b = setup(ne)
        for seq in ([1], [0], [1, 0], [1, 1, 0]):
            a = sweeps(tps, mg, u, b, 0, seq)
            errs = [float((a - sweeps(tps, mg, u, b, 1, seq, form)).abs().max() / a.abs().max()) for form in (1, 2)]
            err = max(errs)
            worst = max(worst, err)
            flag = "" if err < 1e-12 else "   <-- MISMATCH"
            print("grid %-14s sweeps %-10s max rel diff vs rows: mirrored half waves %.3e, node per lane %.3e%s" % (ne, seq, errs[0], errs[1], flag), flush=True)
            if err >= 1e-12:
                d = (a - m).abs().view(ne[0] + 1, ne[1] + 1, ne[2] + 1, 3).amax(3)
                idx = torch.nonzero(d > 1e-12 * float(a.abs().max()))
                print("   first differing nodes:", idx[:12].tolist(), "count", idx.shape[0], flush=True)
    print("worst", worst)
    return worst


def timing(ns):
    for n in ns:
        tps, mg, u, b = setup((n, n, n))
        for march, chunks in [(0, 0), (1, 0), (2, 0), (1, 13), (2, 13), (1, 26), (2, 26)]:       # march: 0 rows, 1 / 2 the marching forms
            _lib.check(lib.vfem_sim_set_option(tps._h, GS_MARCH, 2 * min(march, 1)))
            _lib.check(lib.vfem_sim_set_option(tps._h, GS_MARCH_FORM, max(march, 1)))
            _lib.check(lib.vfem_sim_set_option(tps._h, GS_MARCH_CHUNKS, chunks))
            uu = u.clone()
            _lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(uu), _ptr(b), 1, 2, _stream()))
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for fwd in (1, 0):          # two sweeps per call, as inside a V-cycle: the marching kernel ends in u without a copy
                    _lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(uu), _ptr(b), fwd, 2, _stream()))
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 4)
            print("n %d  %-22s chunks %2d: %.3f ms per sweep" % (n, ("rows", "marching, half waves", "marching, node per lane")[march], chunks, best * 1e3), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(0 if check() < 1e-12 else 1)
    timing([int(a) for a in sys.argv[2:]] or [256])
