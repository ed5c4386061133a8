"""Degree-2 simulator (27-node hexahedra): HIP path against the numpy oracle (oracle.vfem_oracle.OracleSimQ2)."""
import numpy as np
import pytest

from helpers import MATERIAL

pytestmark = pytest.mark.gpu


def _pair(ne, dom, seed):
    from ndr_amd import pyVoxelFEM as pv
    from oracle import vfem_oracle as vo
    t = pv.TensorProductSimulator([2, 2, 2], dom, ne)
    t.readMaterial(MATERIAL)
    young, poisson = pv._read_isotropic_material(MATERIAL)
    o = vo.OracleSimQ2(ne, dom, young, poisson)
    t.E_min = o.Emin = 1e-4
    rho = np.random.default_rng(seed).uniform(0.05, 1.0, size=o.num_elems)
    o.rho = rho.copy()
    t.setElementDensities(rho)
    return t, o


@pytest.mark.parametrize("ne,dom", [((1, 1, 1), ([0, 0, 0], [1, 1, 1])), ((3, 2, 4), ([0, 0, 0], [1.5, 1, 2])),
                                    ((5, 7, 33), ([-1, 0, 0], [1, 3, 7])), ((2, 9, 70), ([0, 0, 0], [1, 1, 1]))])
def test_q2_apply_and_gradient_match_oracle(ne, dom):
    t, o = _pair(ne, dom, 5)
    assert type(t).__name__ == "TensorProductSimulator2_2_2"
    assert t.numNodes() == o.num_nodes and t.numElements() == o.num_elems
    K0 = t.fullDensityElementStiffnessMatrix()
    assert np.abs(K0 - o.K0).max() < 1e-13 * np.abs(o.K0).max()
    u = np.random.default_rng(7).standard_normal((o.num_nodes, 3))
    a, b = t.applyK(u), o.apply_k(u)
    assert np.abs(a - b).max() < 1e-12 * np.abs(b).max()
    g, go = t.complianceGradient_device(u).cpu().numpy(), o.compliance_gradient(u)
    assert np.abs(g - go).max() < 1e-12 * np.abs(go).max()
    assert np.array_equal(t.getDensities(), o.rho)


def test_q2_operator_properties_at_size():
    """symmetry <v, K u> = <u, K v>, rigid-body null space and positivity on a grid too large for the oracle"""
    import torch
    from ndr_amd import pyVoxelFEM as pv
    ne = (48, 40, 64)
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1.2, 1.0, 1.6]), ne)
    t.readMaterial(MATERIAL)
    g = torch.Generator(device="cuda").manual_seed(3)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    Ku, Kv = t.applyK_device(u), t.applyK_device(v)
    a, b = float((v * Ku).sum()), float((u * Kv).sum())
    assert abs(a - b) < 1e-11 * max(abs(a), abs(b), float(Ku.norm() * v.norm()))
    assert float((u * Ku).sum()) > 0
    ones = torch.ones_like(u)
    assert float(t.applyK_device(ones).abs().max()) < 1e-10 * float(Ku.abs().max())


def test_q2_apply_at_config5_size_512_cubed():
    """BASELINE config 5's own grid (512^3 elements of degree 2 = 1025^3 nodes, 3.2 G dofs) on one GPU: the marching apply against
    the dense-matrix gather kernel (which shares no code with it) on the whole field, symmetry, rigid-body null space, positivity"""
    import torch
    from ndr_amd import _lib, pyVoxelFEM as pv
    lib = _lib.load()
    ne = (512, 512, 512)
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1.0, 1.0, 1.0]), ne)
    t.readMaterial(MATERIAL)
    g = torch.Generator(device="cuda").manual_seed(17)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    Ku = t.applyK_device(u)
    _lib.check(lib.vfem_gsim_set_option(t._h, 6, 1))              # VFEM_OPT_Q2_IMPL = dense gather
    Kg = t.applyK_device(u)
    _lib.check(lib.vfem_gsim_set_option(t._h, 6, 0))
    scale = float(Kg.abs().max())
    Kg -= Ku
    assert float(Kg.abs().max()) < 1e-12 * scale
    del Kg
    v = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    Kv = t.applyK_device(v)
    a, b = float((v * Ku).sum()), float((u * Kv).sum())
    assert abs(a - b) < 1e-11 * float(Ku.norm() * v.norm())
    assert float((u * Ku).sum()) > 0
    del Kv, v
    assert float(t.applyK_device(torch.ones_like(u)).abs().max()) < 1e-10 * scale
    # rigid rotation about z: u = (-y, x, 0)
    n1 = 2 * ne[0] + 1
    ax = torch.arange(n1, dtype=torch.float64, device="cuda") / (n1 - 1)
    rot = torch.zeros_like(u).reshape(n1, n1, n1, 3)
    rot[..., 0] = -ax[None, :, None]
    rot[..., 1] = ax[:, None, None]
    assert float(t.applyK_device(rot.reshape(-1, 3)).abs().max()) < 1e-10 * scale


@pytest.mark.parametrize("ne", [(17, 6, 63), (16, 5, 130), (33, 9, 70), (70, 3, 3)])
def test_q2_apply_kernels_agree_across_chunk_seams(ne):
    """the marching kernel (x-chunks with a lead-in element, z-chunks of 63 elements, two y colours), the pencil kernel and the
    dense gather kernel are three independent evaluations of K u"""
    import torch
    from ndr_amd import _lib, pyVoxelFEM as pv
    lib = _lib.load()
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1.0, 0.7, 1.3]), ne)
    t.readMaterial(MATERIAL)
    g = torch.Generator(device="cuda").manual_seed(11)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    res = {}
    for impl in (1, 0, 2):
        _lib.check(lib.vfem_gsim_set_option(t._h, 6, impl))        # VFEM_OPT_Q2_IMPL, a property of this simulator only
        res[impl] = t.applyK_device(u).clone()
    scale = float(res[1].abs().max())
    assert float((res[0] - res[1]).abs().max()) < 1e-12 * scale
    assert float((res[2] - res[1]).abs().max()) < 1e-12 * scale


def test_unknown_degrees_are_refused():
    from ndr_amd import pyVoxelFEM as pv
    with pytest.raises(RuntimeError):
        pv.TensorProductSimulator([3, 3, 3], ([0, 0, 0], [1, 1, 1]), (2, 2, 2))


@pytest.mark.parametrize("ne,levels", [((8, 8, 8), 2), ((16, 12, 8), 2), ((16, 16, 16), 3)])
def test_q2_level1_on_the_fly_equals_stored_element_matrices(ne, levels):
    """VFEM_OPT_Q2_L1_VIRTUAL: level 1 evaluated as sum_f E_f cK0[f] per node (k_q2_level1) against the stored 81 x 81 Galerkin
    matrices (MG.hh:604-669): the same operator, sweep, residual, level-2 matrices (built through the scratch buffer) and
    therefore the same PCG run"""
    import torch
    from helpers import BC_CANTILEVER
    from ndr_amd import _lib, pyVoxelFEM as pv
    lib = _lib.load()
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [2, 1, 1]), list(ne))
    t.readMaterial(MATERIAL)
    t.applyDisplacementsAndLoadsFromFile(BC_CANTILEVER)
    t.E_min = 1e-4
    g = torch.Generator(device="cuda").manual_seed(4)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = t.multigridSolver(levels)
    n1 = mg._nn(1)
    u = torch.randn((n1, 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((n1, 3), dtype=torch.float64, device="cuda", generator=g)
    f = t.buildLoadVector_device()
    got = {}
    for mode in (0, 1):
        _lib.check(lib.vfem_gsim_set_option(t._h, 14, mode))
        mg.updateElementStiffnessMatrices()
        hist = []
        x = mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, 50, 1e-9, None, 1, 2, True,
                                                      residual_cb=lambda it, r: hist.append(r))
        got[mode] = (mg.applyK_device(1, u), mg.computeResidual_device(1, u, b), mg.smoothing_device(1, u, b, True),
                     mg.smoothing_device(1, u, b, False), mg.applyK_device(2, mg.restriction_device(1, u)) if levels >= 2 else u,
                     x, hist)
    for a, c in zip(got[0][:6], got[1][:6]):
        assert float((a - c).abs().max()) < 1e-12 * float(a.abs().max())
    assert len(got[0][6]) == len(got[1][6])
    assert max(abs(p - q) / p for p, q in zip(got[0][6], got[1][6])) < 1e-8
    with pytest.raises(RuntimeError):
        _lib.check(lib.vfem_gsim_set_option(t._h, 14, 3))


@pytest.mark.parametrize("ne", [(6, 4, 10), (2, 2, 140)])
def test_q2_finest_level_sweep_orders_agree(ne):
    """VFEM_OPT_Q2_GS_IMPL: the sweep ordered by neighbour node (each distinct neighbour read once) against the element-by-element
    gather and against the dense gather kernels of the generic path, forward and backward, on a grid with every colour class and
    Dirichlet nodes"""
    import torch
    from helpers import BC_CANTILEVER
    from ndr_amd import _lib, pyVoxelFEM as pv
    lib = _lib.load()
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [2, 1, 1]), list(ne))     # (2, 2, 140): rows of more than one wave
    t.readMaterial(MATERIAL)
    t.applyDisplacementsAndLoadsFromFile(BC_CANTILEVER)
    t.E_min = 1e-4
    g = torch.Generator(device="cuda").manual_seed(9)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = t.multigridSolver(1)
    mg.updateElementStiffnessMatrices()
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    got = {}
    for name, opts in (("rows", ((16, 2),)), ("nodes", ((16, 1),)), ("elements", ((16, 0),)), ("dense", ((6, 1),))):
        for k, v in opts:
            _lib.check(lib.vfem_gsim_set_option(t._h, k, v))
        got[name] = (mg.smoothing_device(0, u, b, True), mg.smoothing_device(0, u, b, False))
        _lib.check(lib.vfem_gsim_set_option(t._h, 6, 0))
        _lib.check(lib.vfem_gsim_set_option(t._h, 16, 2))
    for w in (0, 1):
        scale = float(got["dense"][w].abs().max())
        assert float((got["nodes"][w] - got["dense"][w]).abs().max()) < 1e-12 * scale
        assert float((got["nodes"][w] - got["elements"][w]).abs().max()) < 1e-13 * scale
        assert float((got["rows"][w] - got["elements"][w]).abs().max()) < 1e-13 * scale


def test_q2_axis_by_axis_transfers_equal_the_single_pass():
    """VFEM_OPT_TRANSFER_AXIS: restriction and interpolation as three one-dimensional passes (levels above 100 k nodes) against
    the single-pass gathers, plus the adjoint relation <R r, c> = <r, P c> between the two"""
    import torch
    from ndr_amd import _lib, pyVoxelFEM as pv
    lib = _lib.load()
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1.0, 0.8, 1.2]), [24, 20, 36])      # 49 x 41 x 73 = 147 k nodes
    t.readMaterial(MATERIAL)
    mg = t.multigridSolver(2)
    g = torch.Generator(device="cuda").manual_seed(21)
    r = torch.randn((mg._nn(0), 3), dtype=torch.float64, device="cuda", generator=g)
    c = torch.randn((mg._nn(1), 3), dtype=torch.float64, device="cuda", generator=g)
    base = torch.randn((mg._nn(0), 3), dtype=torch.float64, device="cuda", generator=g)
    got = {}
    for mode in (1, 0):
        _lib.check(lib.vfem_gsim_set_option(t._h, 17, mode))
        got[mode] = (mg.restriction_device(0, r), mg.interpolation_device(0, c), mg.interpolation_device(0, c, out=base.clone()))
    _lib.check(lib.vfem_gsim_set_option(t._h, 17, 1))
    for a, b in zip(got[1], got[0]):
        assert float((a - b).abs().max()) < 1e-13 * float(b.abs().max())
    lhs, rhs = float((got[1][0] * c).sum()), float((r * got[1][1]).sum())
    assert abs(lhs - rhs) < 1e-11 * max(abs(lhs), abs(rhs), 1.0)
