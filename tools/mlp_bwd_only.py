import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ndr_amd.mlp import MLP
from helpers import seeded_mlp_weights
side = (128, 256, 256)
es, nn_, nl, sigma = 1024, 512, 4, 4.0
B, Ws, bs = seeded_mlp_weights(es, nn_, nl, sigma, 88)
m = MLP(3, 1, nn_, nl, es, sigma); m.load_arrays(B, Ws, bs)
if len(sys.argv) > 1: m.set_backward_terms(int(sys.argv[1]))
g = torch.randn(int(np.prod(side)), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
for _ in range(2): m.backward_grid(side, g)
torch.cuda.synchronize()
print("done")
