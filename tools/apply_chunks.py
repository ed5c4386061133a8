import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_hip
from ndr_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ne = (n, n, n)
tps = make_hip(ne, ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
u = torch.randn((tps.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
out = torch.empty_like(u)
ref = None
for nc in [int(a) for a in sys.argv[2:]] or (8, 3, 4, 5, 6, 7, 9, 10, 12, 8):
    set_knob(tps, 7, nc)
    for _ in range(3): lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), 0, _stream())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), 0, _stream())
    b.record(); torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    print("chunks %2d: %.3f ms  (max dev %.1e)" % (nc, a.elapsed_time(b) / 20, float((out - ref).abs().max() / ref.abs().max())), flush=True)
set_knob(tps, 7, 0)
