import glob, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndr_amd.mlp import MLP
rl2 = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mlp_*.npz"))):
    z = np.load(path); es, nn_, nl, sig = [int(v) for v in z["cfg"]]
    m = MLP(3, 1, nn_, nl, es, float(z["sigma"][0]), output_act=torch.nn.Sigmoid() if sig else None)
    m.load_arrays(z["B"], [z["W%d" % i] for i in range(nl)], [z["b%d" % i] for i in range(nl)])
    gw, gb = m.backward(torch.from_numpy(z["coords"]).cuda(), torch.from_numpy(z["gout"]).cuda())
    print(os.path.basename(path), ["%.1e/%.1e" % (rl2(gw[i].cpu().numpy().reshape(z["gW%d" % i].shape), z["gW%d" % i]),
                                                  rl2(gb[i].cpu().numpy().reshape(z["gb%d" % i].shape), z["gb%d" % i])) for i in range(nl)])
rng = np.random.default_rng(88); es, nn_, nl, sigma = 1024, 512, 4, 4.0
B = (rng.standard_normal((es, 3)) * sigma).astype(np.float32)
Ws = [rng.standard_normal((nn_, 2 * es)).astype(np.float32) / np.sqrt(2 * es)] + [rng.standard_normal((nn_, nn_)).astype(np.float32) / np.sqrt(nn_) for _ in range(nl - 2)] + [rng.standard_normal((1, nn_)).astype(np.float32) / np.sqrt(nn_)]
bs = [rng.standard_normal(nn_).astype(np.float32) * 0.1 for _ in range(nl - 1)] + [np.array([0.4], np.float32)]
m = MLP(3, 1, nn_, nl, es, sigma); m.load_arrays(B, Ws, bs)
side = (512, 256, 256); nv = int(np.prod(side))
g = torch.randn(nv, device="cuda")
m.backward_grid(side, g); torch.cuda.synchronize()
t0 = time.perf_counter(); m.backward_grid(side, g); torch.cuda.synchronize(); dt = time.perf_counter() - t0
fl = 2.0 * ((3 * es + 2 * es * nn_ + (nl - 2) * nn_ * nn_ + nn_) + (nl - 2) * nn_ * nn_ + (2 * es * nn_ + (nl - 2) * nn_ * nn_ + nn_)) * nv
print("backward %s: %.3f s, %.1f Mvoxel/s, %.0f TFLOP/s (incl. recomputed forward)" % (side, dt, nv / dt / 1e6, fl / dt / 1e12))
print("mem GB", torch.cuda.mem_get_info())
