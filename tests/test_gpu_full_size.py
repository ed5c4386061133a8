"""-m gpu, full-size known answers: the reference's own logged compliance trajectory of the 3-D cantilever
(256x128x128, v0 = 0.5, 3 coarsening levels, OC; logs/slurm/gt/c1001.log:137-140) replayed through the drop-in
driver API on the HIP path."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_cantilever_256x128x128_matches_reference_log():
    from helpers import BC_CANTILEVER, GOLDEN, MATERIAL
    from ndr_amd import fem
    logs = json.load(open(os.path.join(GOLDEN, "reference_logs.json")))["3d_cantilever_256x128x128"]
    tps, final, binary, hist = fem.ground_truth_topopt(MATERIAL, BC_CANTILEVER, [1, 1, 1], [[0, 0, 0], [2, 1, 1]],
                                                       [256, 128, 128], 3, 0.5, 'OC', 3, use_multigrid=True,
                                                       max_iter=3, obj_history=True, verbose=False)
    want = logs["compliance"]
    for got, ref in zip(hist, want):
        assert abs(got - ref) < 2e-5 * ref, (hist, want)


def test_voxelfem_function_autograd():
    import torch
    from helpers import BC_CANTILEVER, MATERIAL
    from ndr_amd import fem, pyVoxelFEM as pv
    tps = fem.initializeTensorProductSimulator([1, 1, 1], [np.zeros(3), np.array([2.0, 1.0, 1.0])], [32, 16, 16], 0.5, 1, 1e-4, 3,
                                               MATERIAL, BC_CANTILEVER)
    obj = pv.MultigridComplianceObjective(tps.multigridSolver(2))
    obj.tol = 1e-8
    top = pv.TopologyOptimizationProblem(tps, obj, [pv.TotalVolumeConstraint(0.5)], [])
    rho = torch.full((32 * 16 * 16,), 0.5, device="cuda", requires_grad=True)
    c = fem.VoxelFEMFunction.apply(rho, top)
    c.backward()
    g = rho.grad.cpu().numpy()
    assert np.all(g <= 0) and np.isfinite(g).all()
    # directional finite difference
    d = np.random.default_rng(0).uniform(-1, 1, g.size)
    h = 1e-4
    vals = []
    for s in (+1, -1):
        top.setVars((np.full(g.size, 0.5) + s * h * d))
        vals.append(2.0 * top.evaluateObjective())
    fd = (vals[0] - vals[1]) / (2 * h)
    assert abs(fd - 2.0 * float(g @ d)) < 2e-3 * abs(fd)
