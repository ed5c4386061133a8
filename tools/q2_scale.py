import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd import _lib, pyVoxelFEM as pv
for n in (256, 320, 384, 448, 512):
    ne = (n, n, n)
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1, 1, 1]), list(ne))
    t.E_min = 1e-4
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda"))
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda")
    out = t.applyK_device(u); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = t.applyK_device(u)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3): out.copy_(u)
    torch.cuda.synchronize(); dc = (time.perf_counter() - t0) / 3
    print("n=%d apply %.2f ms (%.2f GVoxel/s, %.1f ns/voxel)  copy %.2f ms (%.2f TB/s)" % (n, dt * 1e3, n ** 3 / dt / 1e9, dt / n ** 3 * 1e9, dc * 1e3, 2 * u.numel() * 8 / dc / 1e12), flush=True)
    del t, u, out
    torch.cuda.empty_cache()
