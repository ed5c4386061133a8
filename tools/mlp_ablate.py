"""MLP forward timing ablations (ablation build, key 8): where the 0.12 s of the 512x256x256 forward go."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
from ndr_amd.mlp import MLP
lib = _lib.load()
rng = np.random.default_rng(88); es, nn_, nl, sigma = 1024, 512, 4, 4.0
B = (rng.standard_normal((es, 3)) * sigma).astype(np.float32)
Ws = [rng.standard_normal((nn_, 2 * es)).astype(np.float32) / np.sqrt(2 * es)] + [rng.standard_normal((nn_, nn_)).astype(np.float32) / np.sqrt(nn_) for _ in range(nl - 2)] + [rng.standard_normal((1, nn_)).astype(np.float32) / np.sqrt(nn_)]
bs = [rng.standard_normal(nn_).astype(np.float32) * 0.1 for _ in range(nl - 1)] + [np.array([0.4], np.float32)]
m = MLP(3, 1, nn_, nl, es, sigma); m.load_arrays(B, Ws, bs)
side = (512, 256, 256)
names = {0: "production", 1: "no feature generation", 2: "no hidden layers", 3: "no layer-1 MFMAs"}
for ab in (0, 1, 2, 3, 0):
    set_knob(None, 8, ab)
    for _ in range(2): m.forward_grid(side)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m.forward_grid(side)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("ablation %d %-24s %.4f s" % (ab, names[ab], dt), flush=True)
set_knob(None, 8, 0)
