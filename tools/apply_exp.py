import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_hip
from ndr_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
ne = (512, 512, 512)
tps = make_hip(ne, ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
u = torch.randn((tps.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
out = torch.empty_like(u)
names = {0: "production", 1: "no scatter LDS", 2: "no Dm compute", 3: "no compute, no scatter", 4: "no compute/scatter/u reads", 5: "DMA loads + barriers only", 6: "stores + barriers only", 7: "stores only, non-temporal", 8: "production, non-temporal stores", 9: "skeleton, stores to 2 planes", 10: "skeleton, loads from 4 planes", 11: "stores only, row-contiguous 16 B"}
ref = None
for exp in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 0]:
    set_knob(None, 1, exp)
    for _ in range(3):
        lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), 0, _stream())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), 0, _stream())
    b.record(); torch.cuda.synchronize()
    if exp == 0 and ref is None:
        ref = out.clone()
    dev = float((out - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
    print("EXP %d %-32s %.3f ms   max rel deviation from production %.2e" % (exp, names.get(exp, ""), a.elapsed_time(b) / 20, dev), flush=True)
set_knob(None, 1, 0)
