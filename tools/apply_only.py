import sys, time
import numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import make_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t = make_hip((n, n, n), ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device='cuda'))
u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device='cuda')
for _ in range(reps): out = t.applyK_device(u, 0)
torch.cuda.synchronize()
print("done")
