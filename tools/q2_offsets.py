import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd import _lib, pyVoxelFEM as pv
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1, 1, 1]), [n, n, n])
t.E_min = 1e-4
t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda"))
nn = t.numNodes() * 3
u = torch.randn(nn, dtype=torch.float64, device="cuda")
big = torch.empty(nn + (1 << 22), dtype=torch.float64, device="cuda")
print("u %% 2MB = %d, big %% 2MB = %d" % (u.data_ptr() % (1 << 21), big.data_ptr() % (1 << 21)))
for pad in (0, 16, 128, 1024, 8 * 1024, 37 * 1024 + 8, 128 * 1024, 1 << 20, (1 << 21) + 4096):
    out = big[pad:pad + nn]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(lib.vfem_gsim_apply_k(t._h, _ptr(u), _ptr(out), _stream()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pad %8d doubles (%9d B): %.2f ms" % (pad, pad * 8, dt * 1e3), flush=True)
