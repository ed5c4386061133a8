"""degree-2 marching apply: event-timed, with a checksum:  [VFEM_LIB=...] python tools/q2_time.py [n]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd import pyVoxelFEM as pv
for n in [int(a) for a in sys.argv[1:]] or [256, 512]:
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1, 1, 1]), [n, n, n])
    t.E_min = 1e-4
    g = torch.Generator(device="cuda").manual_seed(88)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    for _ in range(3): out = t.applyK_device(u)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): out = t.applyK_device(u)
    e1.record(); torch.cuda.synchronize()
    print("%s n=%d: %.3f ms   checksum %.12e" % (os.path.basename(os.environ.get("VFEM_LIB", "libvfem.so")), n, e0.elapsed_time(e1) / 4, float(out.abs().sum())), flush=True)
    del t, u, out; torch.cuda.empty_cache()
