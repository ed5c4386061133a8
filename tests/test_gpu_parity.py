"""-m gpu parity tests: every HIP operator (through the C ABI, via ndr_amd.pyVoxelFEM) against the CPU
oracle on the same seeded inputs.  fp64 stencil-class operators must agree to 1e-11 relative (different
summation order only); the converged solve to 1e-8 on compliance (north_star asks 1e-5)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import BC_BRIDGE, BC_CANTILEVER, make_hip, make_oracle, relerr, seeded_density

pytestmark = pytest.mark.gpu

TOL_OP = 1e-11

GRIDS = [
    ((8, 4, 4), ([0, 0, 0], [2, 1, 1])),
    ((5, 7, 3), ([0, 0, 0], [1, 1, 1])),          # ragged, odd
    ((16, 8, 8), ([0, 0, 0], [2, 1, 1])),
    ((6, 10, 70), ([0, 0, 0], [4, 2, 1])),        # two z tiles, non-cubic voxels
    ((70, 9, 65), ([0, 0, 0], [1, 2, 3])),        # several x chunks, y tiles, z tiles
    ((1, 1, 1), ([0, 0, 0], [1, 1, 1])),
]


def test_k0_matches_oracle():
    from oracle import vfem_oracle as vo
    for h in ([1, 1, 1], [2 / 256, 1 / 256, 1 / 256], [4 / 512, 2 / 256, 1 / 256]):
        tps = make_hip((2, 2, 2), ([0, 0, 0], [2 * h[0], 2 * h[1], 2 * h[2]]), None)
        lam, mu = vo.lame(1.0, 0.3, 3)
        K0 = vo.element_stiffness([1, 1, 1], h, lam, mu)
        assert relerr(tps.fullDensityElementStiffnessMatrix(), K0) < 1e-13


@pytest.mark.parametrize("ne,dom", GRIDS)
@pytest.mark.parametrize("variant", [0, 1])
def test_apply_k(ne, dom, variant):
    rho = seeded_density(ne, 88)
    o = make_oracle(ne, dom, None, rho)
    t = make_hip(ne, dom, None, rho)
    rng = np.random.default_rng(1)
    u = rng.standard_normal((o.num_nodes, 3))
    ref = o.apply_k(u)
    got = t.applyK_device(u, variant).cpu().numpy()
    assert relerr(got, ref) < TOL_OP


@pytest.mark.parametrize("ne", [(9, 70, 64), (6, 50, 77), (5, 13, 78), (4, 95, 126), (7, 48, 140)])
def test_apply_k_tile_shapes_agree(ne):
    """z remainders of 2, 15, 16 (no strip), 1 and 15 node columns: the apply with the strip tile shape (4 x 16 lanes, default),
    with the strip at the main tiles' chunk length and with main tiles only are bitwise the same, and match the oracle"""
    from ndr_amd import _lib
    from ndr_amd.pyVoxelFEM import _ptr, _stream
    lib = _lib.load()
    dom = ([0, 0, 0], [0.7, 1.3, 1.0])
    rho = seeded_density(ne, 21)
    t, o = make_hip(ne, dom, None, rho), make_oracle(ne, dom, None, rho)
    u = np.random.default_rng(4).standard_normal((o.num_nodes, 3))
    ud = torch.as_tensor(u, device="cuda")
    res = {}
    _lib.check(lib.vfem_sim_set_option(t._h, 11, 0))               # VFEM_OPT_DMA_LX off: the plain 63-column tiling
    for mode in (1, 2, 0):
        _lib.check(lib.vfem_sim_set_option(t._h, 9, mode))         # VFEM_OPT_DMA_STRIP, a property of this simulator only
        out = torch.empty_like(ud)
        _lib.check(lib.vfem_sim_apply_k(t._h, _ptr(ud), _ptr(out), 0, _stream()))
        res[mode] = out
    assert torch.equal(res[1], res[0]) and torch.equal(res[2], res[0])
    assert relerr(res[1].cpu().numpy(), o.apply_k(u)) < 1e-12
    # the line-exclusive tiling (57-column tiles, rows split on 128-byte boundaries of the result): every double is written by exactly
    # one tile and has the value the plain tiling gives it, also into a result buffer that starts 8 or 24 bytes off a line
    _lib.check(lib.vfem_sim_set_option(t._h, 11, 2))
    for off in (0, 1, 3):
        buf = torch.full((ud.numel() + 16,), float("nan"), dtype=torch.float64, device="cuda")
        out = buf[off:off + ud.numel()]
        _lib.check(lib.vfem_sim_apply_k(t._h, _ptr(ud), ctypes.c_void_p(out.data_ptr()), 0, _stream()))
        assert torch.equal(out.view_as(res[0]), res[0]), off
        assert bool(torch.isnan(buf[:off]).all()) and bool(torch.isnan(buf[off + ud.numel():]).all())


def test_apply_k_linearity_and_symmetry_large():
    """size-independent properties at a size the oracle would not finish quickly: <Ku,v> = <u,Kv>."""
    import torch
    ne, dom = (160, 96, 130), ([0, 0, 0], [2, 1, 1])
    t = make_hip(ne, dom, None, seeded_density(ne, 3, "proxy"))
    g = torch.Generator(device="cuda").manual_seed(0)
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    Ku, Kv = t.applyK_device(u), t.applyK_device(v)
    a, b = float((Ku * v).sum()), float((u * Kv).sum())
    assert abs(a - b) < 1e-10 * abs(a)
    Kg = t.applyK_device(u, 1)
    assert float((Ku - Kg).abs().max() / Kg.abs().max()) < TOL_OP
    K2 = t.applyK_device(2.0 * u - 3.0 * v)
    assert float((K2 - (2.0 * Ku - 3.0 * Kv)).abs().max() / Ku.abs().max()) < 1e-12
    # rigid translation is in the null space
    one = torch.ones_like(u)
    assert float(t.applyK_device(one).abs().max()) < 1e-9 * float(Ku.abs().max())


@pytest.mark.parametrize("ne,dom", GRIDS[:5])
def test_compliance_gradient(ne, dom):
    rho = seeded_density(ne, 5)
    o = make_oracle(ne, dom, None, rho)
    t = make_hip(ne, dom, None, rho)
    u = np.random.default_rng(2).standard_normal((o.num_nodes, 3))
    assert relerr(t.complianceGradient_device(u).cpu().numpy(), o.compliance_gradient(u)) < TOL_OP


def _mg_pair(ne, dom, bc, levels, kind="uniform"):
    from oracle import vfem_oracle as vo
    rho = seeded_density(ne, 88, kind)
    o = make_oracle(ne, dom, bc, rho)
    t = make_hip(ne, dom, bc, rho)
    omg = vo.OracleMG(o, levels, nthreads=4)
    omg.update_element_stiffness()
    tmg = t.multigridSolver(levels)
    tmg.updateElementStiffnessMatrices()
    return o, t, omg, tmg


@pytest.mark.parametrize("bc", [BC_CANTILEVER, BC_BRIDGE])
def test_dirichlet_coarsening(bc):
    o, t, omg, tmg = _mg_pair((16, 8, 8), ([0, 0, 0], [2, 1, 1]), bc, 3)
    assert np.array_equal(t.dirichletMask, o.dmask.astype(bool))
    for l in range(1, 4):
        assert np.array_equal(tmg.getSimulator(l).dirichletMask, omg.sims[l].dmask.astype(bool)), l


@pytest.mark.parametrize("ne,levels", [((16, 8, 8), 3), ((24, 8, 16), 2), ((8, 8, 8), 1)])
def test_mg_operators(ne, levels):
    dom = ([0, 0, 0], [2, 1, 1])
    o, t, omg, tmg = _mg_pair(ne, dom, BC_CANTILEVER, levels)
    rng = np.random.default_rng(7)
    for l in range(levels + 1):
        n = omg.sims[l].num_nodes
        u = rng.standard_normal((n, 3))
        b = rng.standard_normal((n, 3))
        assert relerr(tmg.applyK(l, u), omg.apply_k(l, u)) < TOL_OP, ("applyK", l)
        assert relerr(tmg.computeResidual(l, u, b), omg.residual(l, u.copy(), b)) < TOL_OP, ("residual", l)
        if l < levels:
            for fwd in (True, False):
                uo = u.copy()
                omg.enforce_dirichlet(l, uo, True)
                us = uo.copy()
                omg.smoothing(l, us, b, fwd)
                got = tmg.smoothing_device(l, uo, b, fwd).cpu().numpy()
                assert relerr(got, us) < 1e-10, ("smoothing", l, fwd)
            r = rng.standard_normal((n, 3))
            assert relerr(tmg.restriction_device(l, r).cpu().numpy(), omg.restriction(l, r)) < TOL_OP, ("restrict", l)
            c = rng.standard_normal((omg.sims[l + 1].num_nodes, 3))
            assert relerr(tmg.interpolation_device(l, c).cpu().numpy(), omg.interpolation(l, c)) < TOL_OP
    bL = rng.standard_normal((omg.sims[levels].num_nodes, 3))
    ref = omg.sims[levels].solve(bL)
    assert relerr(tmg.coarsestSolve_device(bL).cpu().numpy(), ref) < 1e-8


def test_sweep_kernel_choices_agree():
    """the choices between sweep kernels that share the arithmetic (include/vfem.h VFEM_OPT_*): level-1 element slots shared by
    1 / 2 / 4 / 8 waves, stored-stencil neighbour blocks shared by three waves or not -- against each other and, through the
    oracle comparison of test_mg_operators, against the reference rules; grid with a stencil level above and below the
    wave-per-node threshold"""
    import torch
    from ndr_amd import _lib
    lib = _lib.load()
    ne, dom = (160, 144, 176), ([0, 0, 0], [2, 1, 1])     # level 2: 41 x 37 x 45 = 68 k nodes (above the threshold), level 3: 9 k
    t = make_hip(ne, dom, BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(3)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = t.multigridSolver(4)
    mg.updateElementStiffnessMatrices()
    fields = {l: (torch.randn((mg._nn(l), 3), dtype=torch.float64, device="cuda", generator=g),
                  torch.randn((mg._nn(l), 3), dtype=torch.float64, device="cuda", generator=g)) for l in (1, 2)}
    ref = {}
    # key 21 (VFEM_OPT_L1_STORED): level 1 evaluated on the fly (0), stored as a 27-point stencil (1), as half of it by symmetry (2)
    # key 22 (VFEM_OPT_L1_MERGED): level-1 node rows summed per incident element (0), per mirror class (1), per class with the two z
    # colours of a row in one launch (2); the slot sharing of key 15 belongs to the per-element kernels
    _lib.check(lib.vfem_sim_set_option(t._h, 22, 0))
    for key, values, level in ((15, (1, 2, 4, 8), 1), (22, (0, 1, 2), 1), (21, (0, 1, 2), 1), (18, (0, 1), 2)):
        for v in values:
            _lib.check(lib.vfem_sim_set_option(t._h, key, v))
            if key == 21:
                mg.updateElementStiffnessMatrices()
            for fwd in (True, False):
                got = mg.smoothing_device(level, fields[level][0], fields[level][1], fwd)
                r = ref.setdefault((key, fwd), got)
                assert float((got - r).abs().max()) < 1e-12 * float(r.abs().max()), (key, v, fwd)
        _lib.check(lib.vfem_sim_set_option(t._h, key, {15: 4, 18: 1, 21: 0, 22: 2}[key]))
    with pytest.raises(RuntimeError):
        _lib.check(lib.vfem_sim_set_option(t._h, 15, 3))


def test_transfer_adjointness():
    o, t, omg, tmg = _mg_pair((16, 8, 8), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, 2)
    rng = np.random.default_rng(3)
    r = rng.standard_normal((omg.sims[0].num_nodes, 3))
    c = rng.standard_normal((omg.sims[1].num_nodes, 3))
    lhs = float(np.sum(tmg.restriction_device(0, r).cpu().numpy() * c))
    rhs = float(np.sum(r * tmg.interpolation_device(0, c).cpu().numpy()))
    assert abs(lhs - rhs) < 1e-12 * abs(lhs)


@pytest.mark.parametrize("fmg", [False, True])
def test_mg_solve_cycles(fmg):
    o, t, omg, tmg = _mg_pair((16, 8, 8), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, 2)
    f = o.build_load_vector()
    ref = omg.solve(np.zeros_like(f), f, 2, 2, True, False, fmg)
    got = tmg.solve(np.zeros_like(f), f, 2, 2, True, False, None, fmg)
    assert relerr(got, ref) < 1e-9


@pytest.mark.parametrize("bc,ne,dom,levels,kind", [
    (BC_CANTILEVER, (32, 16, 16), ([0, 0, 0], [2, 1, 1]), 2, "uniform"),
    (BC_BRIDGE, (32, 16, 8), ([0, 0, 0], [4, 2, 1]), 2, "proxy"),
    (BC_CANTILEVER, (32, 16, 16), ([0, 0, 0], [2, 1, 1]), 3, "proxy"),
])
def test_pcg_matches_oracle(bc, ne, dom, levels, kind):
    o, t, omg, tmg = _mg_pair(ne, dom, bc, levels, kind)
    f = o.build_load_vector()
    assert relerr(t.buildLoadVector(), f) < 1e-15
    uo = omg.pcg(np.zeros_like(f), f, 100, 1e-6, 1, 2, True)
    ug = tmg.preconditionedConjugateGradient(np.zeros_like(f), f, 100, 1e-6, None, 1, 2, True)
    assert tmg.last_iterations == omg.last_iters
    co, cg = float(np.sum(f * uo)), float(np.sum(f * ug))
    assert abs(co - cg) < 1e-8 * abs(co)
    assert relerr(ug, uo) < 1e-6
    # sensitivity parity (north_star: 1e-5 relative)
    assert relerr(t.complianceGradient_device(ug).cpu().numpy(), o.compliance_gradient(uo)) < 1e-6
    # direct-solve cross check of the converged answer
    ud = o.solve(f)
    assert abs(float(np.sum(f * ud)) - cg) < 1e-5 * abs(cg)


@pytest.mark.parametrize("mgit,nsm,fmg,sym,levels,tol", [
    (1, 1, False, True, 2, 1e-6),       # the binding's own defaults (VoxelFEM.cc:120-125): V-cycle preconditioner, one smoothing step
    (1, 2, False, True, 2, 1e-6),
    (2, 1, True, True, 2, 1e-6),
    (2, 2, False, False, 2, 1e-6),      # setSymmetricGaussSeidel(False): forward sweeps only (MG.hh:549)
    (1, 1, True, False, 3, 1e-6),
    (2, 2, True, True, 1, 1e-7),        # eval/eval_fourfeat.py:149-151: one coarsening level, mgIterations 2, tol 1e-7
])
def test_pcg_parameterisations_and_callback_match_oracle(mgit, nsm, fmg, sym, levels, tol):
    """the other solver parameterisations the same API reaches (MG.hh:679-732), iterate by iterate through it_callback(i, x, r)"""
    ne, dom = (32, 16, 16), ([0, 0, 0], [2, 1, 1])
    o, t, omg, tmg = _mg_pair(ne, dom, BC_CANTILEVER, levels, "proxy")
    omg.symmetric_gs = sym
    tmg.setSymmetricGaussSeidel(sym)
    f = o.build_load_vector()
    seen_o, seen_g = [], []
    uo = omg.pcg(np.zeros_like(f), f, 100, tol, mgit, nsm, fmg, callback=lambda i, x, r: seen_o.append((i, x.copy(), r.copy())))
    ug = tmg.preconditionedConjugateGradient(np.zeros_like(f), f, 100, tol, lambda i, x, r: seen_g.append((i, x.copy(), r.copy())), mgit, nsm, fmg)
    assert tmg.last_iterations == omg.last_iters == len(seen_o) == len(seen_g)
    un, rn = np.abs(uo).max(), np.abs(f).max()
    for (io, xo, ro), (ig, xg, rg) in zip(seen_o, seen_g):
        assert io == ig
        assert np.abs(xg.reshape(xo.shape) - xo).max() < 1e-7 * un, (io, "x")
        assert np.abs(rg.reshape(ro.shape) - ro).max() < 1e-7 * rn, (io, "r")
    co, cg = float(np.sum(f * uo)), float(np.sum(f * ug))
    assert abs(co - cg) < 1e-8 * abs(co)
    assert relerr(ug, uo) < 1e-6
    # the final residual really is below the tolerance asked for (forward-only sweeps make the preconditioner non-symmetric: CG may
    # then run into maxIter -- on both sides alike, which the comparisons above have checked)
    if omg.last_iters < 100:
        assert np.linalg.norm(o.apply_k(ug)[o.dmask == 0] - f[o.dmask == 0]) <= 1.01 * tol * np.linalg.norm(f)


def test_apply_k_at_the_headline_size_512_cubed():
    """k_apply_dma at the size the metric is quoted on (9 z-tiles + strip, 8 x-chunks): against the independent gather kernel,
    symmetry, linearity, rigid translations in the null space"""
    ne, dom = (512, 512, 512), ([0, 0, 0], [1, 1, 1])
    t = make_hip(ne, dom, None)
    g = torch.Generator(device="cuda").manual_seed(5)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    Ku, Kv = t.applyK_device(u), t.applyK_device(v)
    Kg = t.applyK_device(u, 1)
    scale = float(Kg.abs().max())
    assert float((Ku - Kg).abs().max()) < TOL_OP * scale
    del Kg
    a, b = float((Ku * v).sum()), float((u * Kv).sum())
    assert abs(a - b) < 1e-10 * float(Ku.norm() * v.norm())
    w = 2.0 * u - 3.0 * v
    K2 = t.applyK_device(w)
    K2 -= 2.0 * Ku
    K2 += 3.0 * Kv
    assert float(K2.abs().max()) < 1e-12 * scale
    del K2, w, Kv, v
    assert float(t.applyK_device(torch.ones_like(u)).abs().max()) < 1e-9 * scale
    assert float((u * Ku).sum()) > 0


def test_objective_and_problem_api():
    """the pyVoxelFEM call sequence of fem.ground_truth_topopt (fem.py:31-87) on a small cantilever"""
    from ndr_amd import pyVoxelFEM as pv
    from oracle import vfem_oracle as vo
    ne, dom = (16, 8, 8), ([0, 0, 0], [2, 1, 1])
    t = make_hip(ne, dom, BC_CANTILEVER, None, v0=0.5)
    obj = pv.MultigridComplianceObjective(t.multigridSolver(2))
    top = pv.TopologyOptimizationProblem(t, obj, [pv.TotalVolumeConstraint(0.5)],
                                         [pv.SmoothingFilter(), pv.ProjectionFilter()])
    obj.tol, obj.mgIterations, obj.fullMultigrid, obj.zeroInit, obj.mgSmoothingIterations = 1e-8, 1, True, False, 2
    oco = pv.OCOptimizer(top)
    top.setVars(t.getDensities())
    o = make_oracle(ne, dom, BC_CANTILEVER, None, v0=0.5)
    oobj = vo.OracleComplianceObjective(o)
    otop = vo.OracleProblem(o, oobj, [vo.OracleVolumeConstraint(0.5)], [vo.OracleSmoothingFilter(), vo.OracleProjectionFilter()])
    ooc = vo.OracleOC(otop)
    otop.set_vars(o.rho.copy())
    for _ in range(3):
        a, b = 2.0 * top.evaluateObjective(), 2.0 * otop.evaluate_objective()
        assert abs(a - b) < 1e-6 * abs(b)
        assert relerr(top.evaluateObjectiveGradient(), otop.evaluate_objective_gradient()) < 1e-5
        oco.step()
        ooc.step()
    assert relerr(top.getVars(), otop.cached[0]) < 1e-4


def test_apply_plane_range_equals_full_apply():
    """vfem_sim_apply_k_planes writes exactly the requested output planes, bit-identical to the whole-grid launch"""
    import ctypes
    from ndr_amd import _lib
    from ndr_amd.pyVoxelFEM import _ptr, _stream
    ne = (37, 20, 70)
    t = make_hip(ne, ([0, 0, 0], [1, 1, 1]), None, seeded_density(ne, 4))
    g = torch.Generator(device="cuda").manual_seed(1)
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    full = t.applyK_device(u).view(ne[0] + 1, -1)
    for lo, hi in [(0, 0), (ne[0], ne[0]), (1, ne[0] - 1), (5, 5), (3, 20), (0, ne[0])]:
        out = torch.full_like(u, float("nan"))
        _lib.check(t._lib.vfem_sim_apply_k_planes(t._h, _ptr(u), _ptr(out), lo, hi, _stream()))
        ov = out.view(ne[0] + 1, -1)
        assert torch.equal(ov[lo:hi + 1], full[lo:hi + 1]), (lo, hi)
        assert bool(torch.isnan(ov[:lo]).all()) and bool(torch.isnan(ov[hi + 1:]).all()), (lo, hi)
    with pytest.raises(RuntimeError):
        _lib.check(t._lib.vfem_sim_apply_k_planes(t._h, _ptr(u), _ptr(out), 0, ne[0] + 1, _stream()))


def test_get_k_constant_strain_load_read_densities(tmp_path):
    """the three TensorProductSimulator methods of VoxelFEM.cc:54,62,66 against the oracle: assembled matrix (upper
    triangle, compressed columns), unit-strain load, densities from a Gmsh element field"""
    import scipy.sparse as sp
    from ndr_amd import io
    ne, dom = (6, 4, 4), ([0, 0, 0], [1.5, 1.0, 0.8])      # even ny, nz: the cantilever's point load sits on the mid node
    rho = seeded_density(ne, 5)
    t, o = make_hip(ne, dom, BC_CANTILEVER, rho), make_oracle(ne, dom, BC_CANTILEVER, rho)
    K = t.getK()
    A = o.assemble()
    assert K.symmetry_mode == "UPPER_TRIANGLE" and K.m == K.n == 3 * o.num_nodes and K.nz == len(K.Ax) == K.Ap[-1]
    assert abs(sp.triu(K.toSciPy(), 1) - sp.triu(A, 1)).max() < 1e-13 and sp.tril(K.toSciPy(), -1).nnz == 0
    assert abs(K.full() - A).max() < 1e-13 and abs(K.trace() - A.diagonal().sum()) < 1e-12
    u = np.random.default_rng(1).standard_normal(3 * o.num_nodes)
    assert relerr(K.apply(u), t.applyK(u.reshape(-1, 3)).reshape(-1)) < 1e-12        # the matrix-free apply is this matrix
    eps = np.array([[0.3, 0.1, -0.2], [0.1, -0.5, 0.4], [-0.2, 0.4, 1.0]])
    assert relerr(t.constantStrainLoad(eps), o.constant_strain_load(eps)) < 1e-13
    # densities through a .msh element field, elements written in reverse order: the cell comes from the element centroid
    V, F = t.getMesh()
    path = str(tmp_path / "rho.msh")
    w = io.MSHFieldWriter(path, V, F[::-1])
    w.addField("density", rho[::-1])
    del w
    t2 = make_hip(ne, dom, None, None)
    t2.readDensities(path)
    assert np.array_equal(t2.getDensities(), rho)
    with pytest.raises(RuntimeError):
        t2.readDensities(str(tmp_path / "rho.vtk"))


def test_direct_solve_cache_follows_boundary_condition_changes():
    """TPS::solve after a change of the Dirichlet mask must not reuse the hierarchy (coarse masks) built for the old one"""
    ne, dom = (16, 8, 8), ([0, 0, 0], [2, 1, 1])
    rho = seeded_density(ne, 9)
    t, o = make_hip(ne, dom, BC_CANTILEVER, rho), make_oracle(ne, dom, BC_CANTILEVER, rho)
    f = o.build_load_vector()
    assert relerr(t.solve(f), o.solve(f)) < 1e-8
    mask = t.dirichletMask
    last = np.arange(o.num_nodes).reshape(17, 9, 9)[-1].reshape(-1)
    mask[last, 0] = True                                   # add an x-roller on the far face
    t.dirichletMask = mask
    o.dmask[last, 0] = 1
    o._lu = None
    assert relerr(t.solve(f), o.solve(f)) < 1e-8


def test_tuning_and_cross_check_options_agree():
    """Every runtime option of `vfem_sim_set_option` that no other test touches selects between implementations of the same
    arithmetic (apply: LDS-DMA / register-staged, planes in flight, x-chunks; level-0 sweeps: plain gather / row streaming with or
    without fused z colours and resident coefficients / marching with different chunk counts; level 1: diagonal blocks precomputed or
    not): each value must reproduce the default's result (no compiled path without a test)."""
    import torch
    from ndr_amd import _lib
    lib = _lib.load()
    ne, dom = (48, 40, 72), ([0, 0, 0], [2, 1, 1])
    t = make_hip(ne, dom, BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(11)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = t.multigridSolver(3)
    mg.updateElementStiffnessMatrices()
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    x = {l: torch.randn((mg._nn(l), 3), dtype=torch.float64, device="cuda", generator=g) for l in (0, 1)}
    b = {l: torch.randn((mg._nn(l), 3), dtype=torch.float64, device="cuda", generator=g) for l in (0, 1)}

    def opt(key, value):
        _lib.check(lib.vfem_sim_set_option(t._h, key, value))

    def close(a, r, tol, what):
        assert float((a - r).abs().max()) <= tol * float(r.abs().max()), what

    # ---- apply: key 4 implementation, key 0 planes in flight of the register-staged one, key 7 x-chunks of the LDS-DMA one
    ref = t.applyK_device(u).clone()
    opt(4, 1)
    for planes in (2, 3, 4):
        opt(0, planes)
        close(t.applyK_device(u), ref, 1e-13, ("register-staged apply", planes))
    opt(4, 0)
    for chunks in (1, 2, 5, 0):
        opt(7, chunks)
        close(t.applyK_device(u), ref, 1e-13, ("x-chunks of the apply", chunks))

    # ---- level-0 sweeps: key 19 marching off / forced, key 20 its chunks, key 2 plain gather, key 10 fused z colours, key 13 resident K0
    def sweeps0():
        return [mg.smoothing_device(0, x[0], b[0], fwd).clone() for fwd in (True, False)]

    opt(19, 0)
    ref0 = sweeps0()
    for key, values in ((10, (0, 1)), (13, (0, 1)), (2, (1, 0))):
        for v in values:
            opt(key, v)
            for got, r in zip(sweeps0(), ref0):
                close(got, r, 1e-12, ("level-0 sweep", key, v))
    opt(19, 2)
    for chunks in (1, 3, 0):
        opt(20, chunks)
        for got, r in zip(sweeps0(), ref0):
            close(got, r, 1e-12, ("marching sweep chunks", chunks))
    opt(19, 1)

    # ---- level 1: key 12 precomputed diagonal blocks (per-element kernels, key 22 = 0)
    opt(22, 0)
    ref1 = None
    for v in (0, 1, 0):
        opt(12, v)
        mg.updateElementStiffnessMatrices()
        got = [mg.smoothing_device(1, x[1], b[1], fwd).clone() for fwd in (True, False)]
        if ref1 is None:
            ref1 = got
        for a, r in zip(got, ref1):
            close(a, r, 1e-12, ("level-1 diagonal blocks", v))
    opt(22, 2)
