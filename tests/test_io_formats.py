"""On-disk formats around the path (host only): .msh field files and .vtr grids round-trip."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _hex_grid(ne):
    nn = np.array(ne) + 1
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in nn], indexing="ij"), -1).reshape(-1, 3)
    V = idx * np.array([0.5, 0.25, 1.0])
    eidx = np.stack(np.meshgrid(*[np.arange(n) for n in ne], indexing="ij"), -1).reshape(-1, 3)
    nstr = np.array([nn[1] * nn[2], nn[2], 1])
    order = [0, 1, 3, 2, 4, 5, 7, 6]
    loc = [np.array([(m >> 2) & 1, (m >> 1) & 1, m & 1]) @ nstr for m in range(8)]
    F = np.stack([eidx @ nstr + loc[m] for m in order], axis=1)
    return V, F


def test_msh_field_round_trip(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "VoxelFEM", "python"))
    import mesh
    ne = (3, 2, 4)
    V, F = _hex_grid(ne)
    rho = np.random.default_rng(0).uniform(size=F.shape[0])
    u = np.random.default_rng(1).standard_normal((V.shape[0], 3))
    p = str(tmp_path / "d.msh")
    w = mesh.MSHFieldWriter(p, V, F)
    w.addField("density", rho)
    w.addField("u", u)
    r = mesh.MSHFieldParser3(mshPath=p)
    assert np.array_equal(r.scalarField("density"), rho)          # %.17g round-trips doubles exactly
    assert np.array_equal(r.vectorField("u"), u)
    assert np.array_equal(r.vertices(), V) and np.array_equal(r.elements(), F)
    with pytest.raises(RuntimeError):
        r.scalarField("nope")
    with pytest.raises(RuntimeError):
        w.addField("bad", np.zeros(5))


def test_vtr_round_trip(tmp_path):
    from ndr_amd import io
    d = np.random.default_rng(2).uniform(size=(4, 3, 5))
    f = io.grid_to_vtr(str(tmp_path / "g"), np.arange(5), np.arange(4), np.arange(6), cellData={"data": d})
    assert f.endswith(".vtr")
    x, y, z, cd, pd = io.read_vtr(f)
    assert np.array_equal(cd["data"], d) and x.size == 5 and z.size == 6 and pd == {}
    with pytest.raises(RuntimeError):
        io.grid_to_vtr(str(tmp_path / "h"), np.arange(5), np.arange(4), np.arange(6), cellData={"data": d.T})


def test_side_module_shims_import():
    sys.path.insert(0, os.path.join(ROOT, "VoxelFEM", "python"))
    import parallelism
    parallelism.set_max_num_tbb_threads(8)
    parallelism.set_gradient_assembly_num_threads(4)
    parallelism.unset_max_num_tbb_threads()
    with pytest.raises(RuntimeError):
        parallelism.set_max_num_tbb_threads(0)
