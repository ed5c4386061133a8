"""Fourier-feature MLP: golden vectors produced by the reference's networks.MLP (tests/golden/make_mlp_fixtures.py).
CPU: the oracle restatement must reproduce them (fp32).  GPU: the fused fp16-MFMA kernel against the same vectors
(tolerance below) and the grid entry point against the explicit-coordinate one."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mlp_*.npz")))

# fp16 operands, fp32 accumulation: absolute tolerance on the (pre-sigmoid O(1)) output, set from measurement
TOL_F16 = 6e-3


def _load(path):
    z = np.load(path)
    es, nn_, nl, sig = [int(v) for v in z["cfg"]]
    Ws = [z["W%d" % i] for i in range(nl)]
    bs = [z["b%d" % i] for i in range(nl)]
    return z, es, nn_, nl, bool(sig), Ws, bs


@pytest.mark.parametrize("path", FIXTURES)
def test_oracle_reproduces_reference_mlp(path):
    from oracle import vfem_oracle as vo
    z, es, nn_, nl, sig, Ws, bs = _load(path)
    out = vo.mlp_forward(z["coords"], z["B"], Ws, bs, sig).reshape(z["out"].shape)
    assert np.abs(out - z["out"]).max() < 2e-5


def test_mgrid_rule():
    from oracle import vfem_oracle as vo
    z = np.load(FIXTURES[0])
    g = vo.get_mgrid(z["coords"].shape[1:4])
    assert np.array_equal(g, z["coords"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES)
def test_hip_mlp_matches_reference(path):
    import torch
    from ndr_amd.mlp import MLP
    z, es, nn_, nl, sig, Ws, bs = _load(path)
    m = MLP(3, 1, nn_, nl, es, float(z["sigma"][0]), output_act=torch.nn.Sigmoid() if sig else None)
    m.load_arrays(z["B"], Ws, bs)
    got = m.forward(torch.from_numpy(z["coords"]).cuda()).cpu().numpy().reshape(z["out"].shape)
    err = np.abs(got - z["out"]).max()
    assert err < TOL_F16, err
    side = z["coords"].shape[1:4]
    grid = m.forward_grid(side).cpu().numpy().reshape(z["out"].shape)
    assert np.abs(grid - got).max() < 2e-3      # same kernel, coordinates regenerated on the fly in fp32
    rho64 = torch.empty(int(np.prod(side)), dtype=torch.float64, device="cuda")
    m.forward_grid(side, out_f64=rho64)
    assert np.abs(rho64.cpu().numpy().reshape(z["out"].shape) - grid).max() < 1e-7


@pytest.mark.gpu
def test_hip_mlp_requires_weights_and_valid_shapes():
    import torch
    from ndr_amd.mlp import MLP
    with pytest.raises(RuntimeError):
        MLP(3, 1, 100, 4, 64, 1.0)                     # n_neurons not a multiple of 32
    m = MLP(3, 1, 64, 3, 32, 1.0)
    with pytest.raises(RuntimeError):
        m.forward(torch.zeros(4, 3))
