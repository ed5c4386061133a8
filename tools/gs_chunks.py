"""Marching level-0 sweep by number of x-chunks, and against the row kernels on smaller grids: python tools/gs_chunks.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()

def sweep_ms(mg, u, b, reps=4):
    for rep in range(2):
        _lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(u), _ptr(b), rep % 2, 2, _stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(reps):
        _lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(u), _ptr(b), rep % 2, 2, _stream()))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * reps)

for ne in ((512, 512, 512), (256, 256, 256), (192, 192, 192), (160, 160, 160), (128, 128, 128), (96, 96, 96), (64, 512, 512), (32, 256, 256)):
    tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), None, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = tps.multigridSolver(0)
    nn = mg._nn(0)
    u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
    row = {}
    _lib.check(lib.vfem_sim_set_option(tps._h, 19, 0))
    row["rows"] = sweep_ms(mg, u, b)
    _lib.check(lib.vfem_sim_set_option(tps._h, 19, 2))
    for ch in (0, 4, 8, 12, 16, 24, 32, 48):
        _lib.check(lib.vfem_sim_set_option(tps._h, 20, ch))
        row["march_ch%d" % ch] = sweep_ms(mg, u, b)
    print(ne, "  ".join("%s %.3f" % kv for kv in row.items()), flush=True)
    del mg, tps, u, b
    torch.cuda.empty_cache()
