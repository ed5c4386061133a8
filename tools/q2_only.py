import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd import pyVoxelFEM as pv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1, 1, 1]), [n, n, n])
t.E_min = 1e-4
t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda"))
u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = t.applyK_device(u); torch.cuda.synchronize()
    print("apply %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
