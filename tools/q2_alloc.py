import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd import _lib, pyVoxelFEM as pv
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
n = 512
t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1, 1, 1]), [n, n, n])
t.E_min = 1e-4
g = torch.Generator(device="cuda").manual_seed(88)
t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
def timed(fn, label):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-28s %.2f ms   ptr %% 2MB = %d" % (label, (time.perf_counter() - t0) * 1e3, r.data_ptr() % (1 << 21)), flush=True)
    return r
out = None
for i in range(4):
    out = timed(lambda: t.applyK_device(u), "applyK_device call %d" % i)
fixed = torch.empty_like(u)
def into_fixed():
    _lib.check(lib.vfem_gsim_apply_k(t._h, _ptr(u), _ptr(fixed), _stream())); return fixed
for i in range(3):
    timed(into_fixed, "into fixed buffer %d" % i)
print(torch.cuda.memory_summary(abbreviated=True)[:600])
