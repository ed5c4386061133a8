import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd.mlp import MLP
rng = np.random.default_rng(88); es, nn_, nl, sigma = 1024, 512, 4, 4.0
B = (rng.standard_normal((es, 3)) * sigma).astype(np.float32)
Ws = [rng.standard_normal((nn_, 2 * es)).astype(np.float32) / np.sqrt(2 * es)] + [rng.standard_normal((nn_, nn_)).astype(np.float32) / np.sqrt(nn_) for _ in range(nl - 2)] + [rng.standard_normal((1, nn_)).astype(np.float32) / np.sqrt(nn_)]
bs = [rng.standard_normal(nn_).astype(np.float32) * 0.1 for _ in range(nl - 1)] + [np.array([0.4], np.float32)]
m = MLP(3, 1, nn_, nl, es, sigma); m.load_arrays(B, Ws, bs)
side = (128, 256, 256)
g = torch.randn(int(np.prod(side)), device="cuda")
for _ in range(2): m.backward_grid(side, g)
torch.cuda.synchronize()
print("done")
