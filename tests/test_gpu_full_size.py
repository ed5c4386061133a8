"""-m gpu, full-size known answers: the reference's own logged compliance trajectories (logs/slurm/gt/) replayed through
the drop-in driver API on the HIP path -- 3-D cantilever 256x128x128 (c1001.log:137-140) and 3-D bridge 320x160x80
(b1000.log:141-144), each: uniform start, 3 coarsening levels, tol 1e-4, OC updates with smoothing + projection filters.

Tolerance.  north_star asks 1e-5 relative on compliance for the same grid and BCs.  The logs print six decimals, i.e. they
carry a rounding of up to 2e-8 relative on the smallest value; the HIP path reproduces all eight logged values to 1e-8
(measured: cantilever <= 4e-10, bridge <= 9.2e-9, profiles/r02_parity_deltas.json), so the test holds them to 1e-7 -- a
hundred times tighter than the bar.  (The iterates are stopped at ||r|| <= 1e-4 ||b||, fem.py:66, like the reference's; that
the compliances still agree to the printed digits says the stopping points coincide: same iteration counts, same cycle.)"""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _replay(key, bc, dom, grid, v0):
    from helpers import GOLDEN, MATERIAL, record_deltas
    from ndr_amd import fem
    want = json.load(open(os.path.join(GOLDEN, "reference_logs.json")))[key]["compliance"]
    n = len(want)
    tps, final, binary, hist = fem.ground_truth_topopt(MATERIAL, bc, [1, 1, 1], dom, list(grid), 3, v0, 'OC', 3,
                                                       use_multigrid=True, max_iter=n, obj_history=True, verbose=False)
    rel = [abs(g - r) / r for g, r in zip(hist, want)]
    record_deltas("log_replay_" + key, {"logged": want, "hip": hist, "relative_delta": rel})
    assert max(rel) < 1e-7, (hist, want, rel)


def test_cantilever_256x128x128_matches_reference_log():
    from helpers import BC_CANTILEVER
    _replay("3d_cantilever_256x128x128", BC_CANTILEVER, [[0, 0, 0], [2, 1, 1]], (256, 128, 128), 0.5)


def test_bridge_320x160x80_matches_reference_log():
    """face load split over 26 001 nodes + x-roller + clamp, non-cubic voxels, v0 = 0.4 (projection filter active from
    iteration 0: proj(0.4) = 0.392)"""
    from helpers import BC_BRIDGE
    _replay("3d_bridge_320x160x80", BC_BRIDGE, [[0, 0, 0], [4, 2, 1]], (320, 160, 80), 0.4)


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
