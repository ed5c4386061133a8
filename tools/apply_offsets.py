"""Sensitivity of the apply kernel to the relative placement of u and out (HBM channel / bank interleaving)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
ne = (512, 512, 512)
tps = make_hip(ne, ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
nn = tps.numNodes()
pool = torch.empty(2 * 3 * nn + (64 << 20) // 8, dtype=torch.float64, device="cuda")
u = pool[:3 * nn].view(nn, 3)
u.normal_(generator=g)
for off_kb in (0, 1, 4, 16, 64, 256, 1024, 2048, 3000, 8192, 16384, 33333):
    o = 3 * nn + off_kb * 128
    out = pool[o:o + 3 * nn].view(nn, 3)
    for _ in range(3):
        lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), 0, _stream())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), 0, _stream())
    b.record(); torch.cuda.synchronize()
    print("out - u_end = %6d KiB: %.3f ms" % (off_kb, a.elapsed_time(b) / 20), flush=True)
