"""Time of the library's dense SPD inverse (the coarsest-level operator) for a 2187-dof matrix (9^3 nodes), and its error."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd import _lib
lib = _lib.load()
for n in (375, 2187):
    rng = np.random.default_rng(5)
    q = rng.standard_normal((n, n))
    a = q @ q.T + n * np.eye(n)
    ref = np.linalg.inv(a)
    d0 = torch.from_numpy(a).cuda()
    for rep in range(4):
        d = d0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert lib.vfem_dense_spd_inverse(n, ctypes.c_void_p(d.data_ptr()), None) == 0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    err = np.abs(d.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("n=%d  %.3f ms  max rel err %.2e" % (n, dt * 1e3, err), flush=True)
