"""-m gpu: the coarsest-level factorisation (dense_spd.hip, the library's own blocked Cholesky / triangular inverse / product;
the reference's exact coarsest solve is CHOLMOD, TPS.hh:834-865) through the C ABI, against numpy."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inverse(a):
    from ndr_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    d = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    status = lib.vfem_dense_spd_inverse(a.shape[0], ctypes.c_void_p(d.data_ptr()), None)
    if status != 0:
        raise RuntimeError(lib.vfem_last_error().decode())
    return d.cpu().numpy()


def _spd(n, seed, cond=1e6):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    w = np.geomspace(1.0, cond, n)
    a = (q * w) @ q.T
    return 0.5 * (a + a.T)


@pytest.mark.parametrize("n", [1, 3, 63, 64, 65, 128, 200, 449, 1000])
def test_dense_spd_inverse_matches_numpy(n):
    """sizes around the 64-wide tiles (padding), one to sixteen tile rows (every level shape of the recursive triangular inverse)"""
    a = _spd(n, 100 + n)
    x = _inverse(a)
    ref = np.linalg.inv(a)
    assert np.array_equal(x, x.T)                                     # both triangles, mirrored exactly
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max(), np.abs(x - ref).max() / np.abs(ref).max()
    r = a @ x - np.eye(n)
    assert np.abs(r).max() < 1e-8, np.abs(r).max()


def test_dense_spd_inverse_is_reproducible_and_rejects_indefinite():
    a = _spd(300, 7)
    x1, x2 = _inverse(a), _inverse(a)
    assert np.array_equal(x1, x2)                                     # fixed summation order: bit for bit
    b = a.copy()
    b[200, 200] = -1.0
    with pytest.raises(RuntimeError, match="not positive definite"):
        _inverse(b)


def test_coarsest_solve_is_exact_on_the_free_dofs():
    """the hierarchy's coarsest solve (vfem_mg_coarsest_solve; TPS.hh:834-865) solves the reduced system: (K x - b) = 0 on the free
    dofs, x = 0 on the fixed ones"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import BC_CANTILEVER, make_hip, seeded_density
    ne, dom = (32, 16, 16), ([0, 0, 0], [2, 1, 1])
    t = make_hip(ne, dom, BC_CANTILEVER, seeded_density(ne, 88, "proxy"))
    mg = t.multigridSolver(2)
    lvl = 2
    mask = mg.getSimulator(lvl).dirichletMask
    b = np.random.default_rng(3).standard_normal(mask.shape)
    x = mg.coarsestSolve_device(b).cpu().numpy()
    kx = mg.applyK(lvl, x)
    assert np.abs(x[mask]).max() == 0.0
    assert np.abs((kx - b)[~mask]).max() <= 1e-9 * np.abs(b).max()
