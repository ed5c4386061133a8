import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from ndr_amd.mlp import MLP
es, nn_, nl = 1024, 512, 4
rng = np.random.default_rng(0)
B = (rng.standard_normal((es, 3)) * 4.0).astype(np.float32)
Ws = [rng.standard_normal((nn_, 2 * es)).astype(np.float32) / np.sqrt(2 * es)] + \
     [rng.standard_normal((nn_, nn_)).astype(np.float32) / np.sqrt(nn_) for _ in range(nl - 2)] + \
     [rng.standard_normal((1, nn_)).astype(np.float32) / np.sqrt(nn_)]
bs = [rng.standard_normal(nn_).astype(np.float32) * 0.1 for _ in range(nl - 1)] + [np.array([0.4], np.float32)]
m = MLP(3, 1, nn_, nl, es, 4.0)
m.load_arrays(B, Ws, bs)
# accuracy vs float64 numpy on a small grid
side = (8, 8, 8)
axes = [np.linspace(0, 1, n, dtype=np.float32) for n in side]
c = np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, 3).astype(np.float64)
proj = 2 * np.pi * c @ B.T.astype(np.float64)
h = np.concatenate([np.sin(proj), np.cos(proj)], -1)
for i in range(nl):
    h = h @ Ws[i].T.astype(np.float64) + bs[i]
    if i < nl - 1: h = np.maximum(h, 0)
got = m.forward_grid(side).cpu().numpy().reshape(-1)
print("max abs err vs fp64 reference: %.3e (output range %.2f..%.2f)" % (np.abs(got - h[:, 0]).max(), h.min(), h.max()))
for side in [(256, 128, 128), (512, 256, 256)]:
    nv = int(np.prod(side))
    m.forward_grid(side); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = m.forward_grid(side)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    fl = 2.0 * (2 * es * nn_ + (nl - 2) * nn_ * nn_ + nn_) * nv
    print("grid %s: %.2f ms  %.3f Gvoxel/s  %.1f TFLOP/s (%.1f%% of 2.5 PF)" % (side, dt * 1e3, nv / dt / 1e9, fl / dt / 1e12, fl / dt / 2.5e15 * 100))
