import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip, seeded_density
from ndr_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
ne = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (24, 8, 16)
tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, seeded_density(ne, 3))
mg = tps.multigridSolver(1)
mg.updateElementStiffnessMatrices()
g = torch.Generator(device="cuda").manual_seed(1)
nn = tps.numNodes()
u0 = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
for fwd in (1, 0):
    res = {}
    for variant in (0, 1, 2):
        set_knob(tps, 2, variant)
        u = u0.clone()
        lib.vfem_mg_smooth_colors(mg._h, 0, _ptr(u), _ptr(b), fwd, 2, 2, _stream())
        torch.cuda.synchronize()
        res[variant] = u.cpu().numpy().reshape(ne[0] + 1, ne[1] + 1, ne[2] + 1, 3)
    print("fwd", fwd, "pair-plain", np.abs(res[0] - res[2]).max(), "rows-plain", np.abs(res[1] - res[2]).max())
    d = np.abs(res[0] - res[1]).max(axis=3)
    bad = np.argwhere(d > 1e-9 * np.abs(res[1]).max())
    print("fwd", fwd, "max diff", d.max(), "bad nodes", len(bad))
    if len(bad):
        print("x:", sorted(set(bad[:, 0]))[:40]); print("y:", sorted(set(bad[:, 1]))[:40]); print("z:", sorted(set(bad[:, 2]))[:40])
        print(bad[:12].tolist()); i=tuple(bad[0]); print('pair', res[0][i], 'rows', res[1][i], 'orig', u0.cpu().numpy().reshape(res[0].shape)[i])
set_knob(tps, 2, 0)
