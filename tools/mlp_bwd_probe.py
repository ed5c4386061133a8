"""Backward pass of the MLP: achieved gradient errors on every fixture (rel. L2 per layer, weights / biases) for the two settings of
VFEM_MLP_OPT_BWD_TERMS, and the whole-grid backward time at the run.md sizes.  usage: python tools/mlp_bwd_probe.py [notime]"""
import glob, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ndr_amd.mlp import MLP
from helpers import seeded_mlp_weights

def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

out = {}
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mlp_*.npz"))) + [os.path.join(ROOT, "tests", "golden", "mlpfull_es1024_nn512_nl4_s4.npz")]:
    z = np.load(path)
    es, nn_, nl, sig = [int(v) for v in z["cfg"]]
    full = "mlpfull" in path
    if full:
        B, Ws, bs = seeded_mlp_weights(es, nn_, nl, float(z["sigma"][0]), int(z["seed"][0])); sig = 0
    else:
        B, Ws, bs = z["B"], [z["W%d" % i] for i in range(nl)], [z["b%d" % i] for i in range(nl)]
    m = MLP(3, 1, nn_, nl, es, float(z["sigma"][0]), output_act=torch.nn.Sigmoid() if sig else None)
    m.load_arrays(B, Ws, bs)
    coords, gout = torch.from_numpy(z["coords"]).cuda(), torch.from_numpy(z["gout"]).cuda()
    row = {}
    for terms in (3, 1):
        m.set_backward_terms(terms)
        gw, gb = m.backward(coords, gout)
        errs = []
        for i in range(nl):
            w = gw[i].cpu().numpy().reshape(Ws[i].shape)
            if full and w.shape[0] > 1: w = w[::int(z["row_stride"][0])]
            errs.append((rel(w, z["gW%d" % i]), rel(gb[i].cpu().numpy(), z["gb%d" % i])))
        row["terms%d" % terms] = errs
    out[os.path.basename(path)] = row
print(json.dumps(out, indent=1))
if len(sys.argv) < 2:
    import bench
    side = (512, 256, 256)
    rng = np.random.default_rng(88); es, nn_, nl, sigma = 1024, 512, 4, 4.0
    B, Ws, bs = seeded_mlp_weights(es, nn_, nl, sigma, 88)
    m = MLP(3, 1, nn_, nl, es, sigma); m.load_arrays(B, Ws, bs)
    g = torch.randn(int(np.prod(side)), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    for terms in (3, 1):
        m.set_backward_terms(terms)
        m.backward_grid(side, g); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.backward_grid(side, g); torch.cuda.synchronize()
        print("terms", terms, "backward seconds", time.perf_counter() - t0, flush=True)
    m.set_backward_terms(3)
    m.set_keep_first_layer(True)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m.forward_grid(side); torch.cuda.synchronize(); tf = time.perf_counter() - t0
        t0 = time.perf_counter(); m.backward_grid(side, g); torch.cuda.synchronize()
        print("first layer kept: forward", tf, "backward", time.perf_counter() - t0, flush=True)
