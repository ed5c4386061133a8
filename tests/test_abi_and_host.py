"""CPU-only checks: the C-ABI library loads and exports every symbol include/vfem.h declares (no compute
calls), the ctypes table matches the header, and the host-side logic (filters, constraint, BC parsing,
Dirichlet coarsening inputs) agrees with the oracle."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _header_functions():
    src = open(os.path.join(ROOT, "include", "vfem.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vfem_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ndr_amd import _lib
    import ctypes
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) > 40
    for n in names:
        assert hasattr(lib, n), n


def test_ctypes_table_matches_header():
    from ndr_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_functions()
    lib = _lib.load()
    assert lib.vfem_version() >= 100
    assert lib.vfem_device_count() >= 0


def test_no_gpu_means_loud_failure():
    from ndr_amd import _lib
    if _lib.load().vfem_device_count() > 0:
        pytest.skip("GPU present")
    from ndr_amd import pyVoxelFEM as pv
    with pytest.raises(RuntimeError):
        pv.TensorProductSimulator([1, 1, 1], [np.zeros(3), np.ones(3)], [4, 4, 4])


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ndr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "voxel_ref" not in txt, (dirpath, f)


@pytest.mark.gpu
def test_filters_and_constraint_match_oracle():
    from ndr_amd import pyVoxelFEM as pv
    from oracle import vfem_oracle as vo
    rng = np.random.default_rng(0)
    for grid in [(7, 5), (6, 4, 5)]:
        x = rng.uniform(0, 1, size=int(np.prod(grid)))
        g = rng.standard_normal(x.size)
        for radius in (1, 2):
            a, b = pv.SmoothingFilter(), vo.OracleSmoothingFilter(radius)
            a._set_grid(grid)
            a.radius = radius
            b.set_grid(grid)
            assert np.abs(a.apply(x) - b.apply(x)).max() < 1e-14
            assert np.abs(a.backprop(g, x) - b.backprop(g, x)).max() < 1e-14
            assert np.abs(pv.applyFilter(a, x) - b.apply(x)).max() < 1e-14
        for beta in (1.0, 4.0):
            a, b = pv.ProjectionFilter(), vo.OracleProjectionFilter(beta)
            a.beta = beta
            assert np.abs(a.apply(x) - b.apply(x)).max() < 1e-15
            assert np.abs(a.backprop(g, x) - b.backprop(g, x)).max() < 1e-15
        c, d = pv.TotalVolumeConstraint(0.4), vo.OracleVolumeConstraint(0.4)
        assert abs(c.evaluate(x) - d.evaluate(x)) < 1e-15
        assert np.abs(c.backprop(x) - d.backprop(x)).max() < 1e-18
    with pytest.raises(RuntimeError):
        pv.ProjectionFilter().beta = -1.0
    with pytest.raises(RuntimeError):
        pv.applyFilter(pv.SmoothingFilter(), np.zeros(4))
    with pytest.raises(RuntimeError):
        f = pv.PythonFilter()
        f._set_grid((2, 2))
        f.apply(np.zeros(4))


def test_region_parser_matches_reference_semantics(tmp_path):
    from ndr_amd.pyVoxelFEM import _parse_regions
    golden = os.path.join(ROOT, "bcs", "3d", "bridge.bc")
    regs = _parse_regions(golden)
    assert [r[0] for r in regs] == ["dirichlet", "dirichlet", "force"]
    assert regs[0][1] == "xyz" and regs[1][1] == "x"
    bad = tmp_path / "bad.bc"
    bad.write_text('{"regions": [{"type": "traction", "value": [0,0,0], "box%": {"minCorner": [0,0,0], "maxCorner": [1,1,1]}}]}')
    with pytest.raises(RuntimeError):
        _parse_regions(str(bad))


@pytest.mark.parametrize("source,kernels,windows", [
    ("kernels_q2.hip", ["k_apply_q2_marchILi0"], 84),
    ("kernels_l1_merged.hip", ["k_l1_mergedILi0E", "k_l1_mergedILi1E", "k_l1_mergedILi2E", "k_l1_pair_rowsILi0E", "k_l1_pair_rowsILi1E"], None),
    ("kernels_gs_march.hip", ["k_gs_march_mf0ILi%dELi%dELi%dEEE" % (a, f, m) for a in (0, 1) for f in (0, 1) for m in (0, 1)], None)])
def test_pipelined_scalar_loads_are_hazard_free(tmp_path, source, kernels, windows):
    """k_apply_q2_march (sload12_issue / sload12_wait), the level-1 per-class kernels and the node-per-lane marching sweep (coef_rows.h:
    srow_issue / srow_wait) leave scalar loads in flight across compiler-generated code; that is only safe while no instruction
    touches the destination SGPRs before the wait.  Compile the kernels to ISA (cross-compile, no GPU needed) and scan them."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "ndr_amd", "csrc", source)
    asm = str(tmp_path / "k.s")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-S", "--cuda-device-only",
                           src, "-o", asm], stderr=subprocess.DEVNULL)
    for k in kernels:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_sload_pipeline.py"), asm, k], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout
        assert " 0 violations" in out.stdout and " 0 request..wait windows" not in out.stdout, out.stdout
        if windows is not None:
            assert "%d request..wait windows" % windows in out.stdout, out.stdout


def test_host_helpers_of_getk_constant_strain_load_read_densities(tmp_path):
    """host logic behind TensorProductSimulator.getK / constantStrainLoad / readDensities (VoxelFEM.cc:54,62,66) against the
    oracle, 2-D and 3-D (no GPU: the helpers take plain arrays)"""
    import scipy.sparse as sp
    import torch
    from ndr_amd import io, pyVoxelFEM as pv
    from oracle import vfem_oracle as vo
    for ne, dom in (((5, 3), ([0, 0], [2.0, 1.0])), ((4, 3, 2), ([0, 0, 0], [1.5, 1.0, 0.8]))):
        N = len(ne)
        o = vo.OracleSim(dom, ne, vo.lame(1.0, 0.3, N))
        o.Emin = 1e-4
        rho = np.random.default_rng(3).uniform(0.05, 1.0, o.num_elems)
        o.set_densities(rho)
        nodes, _ = o.element_dofs()
        K = pv._assemble_upper(nodes, o.K0, o.young(), N, o.num_nodes)
        A = o.assemble()
        assert abs(K.full() - A).max() < 1e-13 and sp.tril(K.toSciPy(), -1).nnz == 0
        assert K.nz == K.Ap[-1] == len(K.Ai) == len(K.Ax) and K.symmetry_mode == "UPPER_TRIANGLE"
        eps = np.random.default_rng(4).standard_normal((N, N))
        eps = eps + eps.T
        lam, mu = o.lam_mu
        F = pv._constant_strain_load(eps, lam, mu, o.h, 1, torch.from_numpy(rho.reshape(ne))).numpy()
        ref = o.constant_strain_load(eps)
        assert np.abs(F - ref).max() < 1e-13 * np.abs(ref).max()
    # element field of a hexahedral .msh, elements in reverse order
    ne = (4, 3, 2)
    idx = np.stack(np.meshgrid(*[np.arange(n + 1) for n in ne], indexing="ij"), -1).reshape(-1, 3)
    V = idx * np.array([0.4, 0.3, 0.2]) + np.array([1.0, -2.0, 0.5])
    nstr = np.array([(ne[1] + 1) * (ne[2] + 1), ne[2] + 1, 1])
    eidx = np.stack(np.meshgrid(*[np.arange(n) for n in ne], indexing="ij"), -1).reshape(-1, 3)
    Fh = np.stack([eidx @ nstr + np.array([(m >> 2) & 1, (m >> 1) & 1, m & 1]) @ nstr for m in (0, 1, 3, 2, 4, 5, 7, 6)], 1)
    rho = np.random.default_rng(5).uniform(size=len(Fh))
    path = str(tmp_path / "field.msh")
    io.MSHFieldWriter(path, V, Fh[::-1]).addField("density", rho[::-1])
    assert np.array_equal(pv._densities_from_msh(path, "density", ne, 3), rho)
    with pytest.raises(RuntimeError):
        pv._densities_from_msh(str(tmp_path / "field.obj"), "density", ne, 3)
