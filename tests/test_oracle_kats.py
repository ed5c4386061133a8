"""Pins the CPU oracle against the reference's own numbers: compliance values the reference logged
(tests/golden/reference_logs.json, from logs/slurm/gt/*.log), textbook element-stiffness entries, exactness of
its Gauss rules on monomials (what VoxelFEM/tests/test_tp_gauss_quadrature.cc checks), and structural
identities of the operators."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import BC_BRIDGE, BC_CANTILEVER, GOLDEN, MATERIAL, make_oracle, seeded_density  # noqa: E402
from oracle import vfem_oracle as vo  # noqa: E402

LOGS = json.load(open(os.path.join(GOLDEN, "reference_logs.json")))


def test_gauss_rules_exact_on_monomials():
    for npts in range(1, 6):
        x, w = vo.gauss_rule(npts)
        assert abs(w.sum() - 1.0) < 1e-15
        for deg in range(0, 2 * npts):
            assert abs(np.sum(w * x ** deg) - 1.0 / (deg + 1)) < 5e-16, (npts, deg)
    # tensor rules as VoxelFEM/tests/test_tp_gauss_quadrature.cc tests them (tp_quadrature_3var_test.inl: monomials
    # x^a y^b z^c, exact integral 1/((a+1)(b+1)(c+1)), tolerance 5e-16): every monomial a degree-D-per-axis rule must
    # integrate exactly, for the rules of the degree-1 (D = 3: 2 points) and degree-2 (D = 5: 3 points) elements and D = 1
    for D in (1, 3, 5):
        for a in range(D + 1):
            for b in range(D + 1):
                for c in range(D + 1):
                    v = vo.integrate_tensor(lambda x, y, z: x ** a * y ** b * z ** c, [D, D, D])
                    assert abs(v - 1.0 / ((a + 1) * (b + 1) * (c + 1))) < 5e-16, (D, a, b, c)


def test_k0_textbook_entries_and_null_space():
    lam, mu = vo.lame(1.0, 0.3, 3)
    K0 = vo.element_stiffness([1, 1, 1], [1, 1, 1], lam, mu)
    assert abs(K0[0, 0] - 0.2350427350427350) < 1e-15      # top3d KE(1,1)
    assert abs(K0[0, 1] - 0.0801282051282051) < 1e-15      # top3d KE(1,2)
    assert np.abs(K0 - K0.T).max() == 0.0
    assert np.linalg.matrix_rank(K0) == 18                   # 6 rigid-body modes
    lam2, mu2 = vo.lame(1.0, 0.3, 2)
    K2 = vo.element_stiffness([1, 1], [1, 1], lam2, mu2)
    assert abs(K2[0, 0] - (0.5 - 0.3 / 6) / (1 - 0.09)) < 1e-15   # 99-line top.m KE(1,1), plane stress
    assert np.linalg.matrix_rank(K2) == 5


def _mbb(ne, dom, bc, v0):
    sim = vo.OracleSim(dom, ne)
    sim.read_material(MATERIAL)
    sim.set_uniform_densities(v0)
    sim.apply_bc_file(bc)
    sim.E0, sim.Emin, sim.gamma = 1.0, 1e-4, 3.0
    obj = vo.OracleComplianceObjective(sim)
    top = vo.OracleProblem(sim, obj, [vo.OracleVolumeConstraint(v0)], [vo.OracleSmoothingFilter(), vo.OracleProjectionFilter()])
    top.set_vars(sim.rho.copy())
    return sim, top


def test_2d_mbb_log_kat():
    k = LOGS["2d_mbb_300x100"]
    sim, top = _mbb([300, 100], ([0, 0], [3, 1]), os.path.join(ROOT, "bcs", "2d", "mbb_beam.bc"), 0.3)
    assert abs(2 * top.evaluate_objective() - k["compliance"][0]) < 5e-7 * k["compliance"][0]
    oc = vo.OracleOC(top)
    obj, con, lam = oc.step()
    assert abs(obj - k["oc_step1"]["objective"]) < 1e-5 * obj              # logged with 6 significant digits
    assert abs(lam - k["oc_step1"]["lambda"]) < 1e-5 * lam
    assert abs(con - k["oc_step1"]["constraint"]) < 2e-10
    assert abs(2 * top.evaluate_objective() - k["compliance"][1]) < 5e-7 * k["compliance"][1]


def test_2d_bridge_log_kat():
    k = LOGS["2d_bridge_250x125"]
    sim, top = _mbb([250, 125], ([0, 0], [2, 1]), os.path.join(ROOT, "bcs", "2d", "bridge.bc"), 0.4)
    assert abs(2 * top.evaluate_objective() - k["compliance"][0]) < 5e-7 * k["compliance"][0]


def test_3d_small_mg_pcg_equals_direct_solve():
    ne, dom = (16, 8, 8), ([0, 0, 0], [2, 1, 1])
    for bc in (BC_CANTILEVER, BC_BRIDGE):
        o = make_oracle(ne, dom, bc, seeded_density(ne, 88))
        f = o.build_load_vector()
        ud = o.solve(f)
        mg = vo.OracleMG(o, 2, nthreads=4)
        u = mg.pcg(np.zeros_like(f), f, 100, 1e-9, 1, 2, True)
        assert mg.last_iters < 60
        assert abs(np.sum(f * u) - np.sum(f * ud)) < 1e-9 * abs(np.sum(f * ud))


def test_galerkin_and_adjointness_identities():
    ne, dom = (8, 4, 4), ([0, 0, 0], [2, 1, 1])
    o = make_oracle(ne, dom, BC_CANTILEVER, seeded_density(ne, 1))
    mg = vo.OracleMG(o, 2)
    mg.update_element_stiffness()
    rng = np.random.default_rng(0)
    for l in (0, 1):
        xc = rng.standard_normal((mg.sims[l + 1].num_nodes, 3))
        lhs = mg.apply_k(l + 1, xc)
        rhs = mg.restriction(l, mg.apply_k(l, mg.interpolation(l, xc)))
        assert np.abs(lhs - rhs).max() < 1e-13 * np.abs(lhs).max()          # K_c = P^T K_f P
        r = rng.standard_normal((mg.sims[l].num_nodes, 3))
        assert abs(np.sum(mg.restriction(l, r) * xc) - np.sum(r * mg.interpolation(l, xc))) < 1e-12 * abs(np.sum(r * r))


def test_sensitivity_finite_difference():
    """Numerical_Derivatives.ipynb recipe: centred FD of the compliance vs the analytic gradient."""
    ne, dom = (8, 4, 4), ([0, 0, 0], [2, 1, 1])
    rho = 0.3 + 0.4 * seeded_density(ne, 2)
    o = make_oracle(ne, dom, BC_CANTILEVER, rho)
    f = o.build_load_vector()
    u = o.solve(f)
    g = o.compliance_gradient(u)
    for e in (0, 17, 100):
        h = 1e-5
        vals = []
        for s in (+1, -1):
            r2 = rho.copy()
            r2[e] += s * h
            o.set_densities(r2)
            vals.append(0.5 * np.sum(f * o.solve(f)))
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(fd - g[e]) < 1e-5 * abs(g[e]) + 1e-10


def test_3d_cantilever_log_kat_full_size():
    """c1001.log:137, iteration 0 of the 256x128x128 cantilever (about a minute on 8 cores)"""
    k = LOGS["3d_cantilever_256x128x128"]
    ne, dom = (256, 128, 128), ([0, 0, 0], [2, 1, 1])
    o = make_oracle(ne, dom, BC_CANTILEVER, None, v0=0.5)
    mg = vo.OracleMG(o, 3, nthreads=8)
    f = o.build_load_vector()
    u = mg.pcg(np.zeros_like(f), f, 100, 1e-6, 1, 2, True)
    assert abs(np.sum(f * u) - k["compliance"][0]) < 1e-5 * k["compliance"][0]


def test_3d_bridge_log_kat_full_size():
    """b1000.log:141 (= b01.log:46), iteration 0 of the 320x160x80 bridge on [0,4]x[0,2]x[0,1], v0 = 0.4: pins the face
    load (force split evenly over the matched nodes, TPS.hh:383-388) and the x-roller mask at full size"""
    k = LOGS["3d_bridge_320x160x80"]
    ne, dom = (320, 160, 80), ([0, 0, 0], [4, 2, 1])
    # the drivers pass the uniform design x = v0 through the smoothing (uniform in, uniform out) and projection filters
    # (Filter.hh:55-79, beta = 1): the physical density of iteration 0 is proj(0.4) = 0.39216, not 0.4 (proj(0.5) = 0.5)
    beta = 1.0
    rho_phys = 0.5 * (np.tanh(0.5 * beta) + np.tanh(beta * (0.4 - 0.5))) / np.tanh(0.5 * beta)
    o = make_oracle(ne, dom, BC_BRIDGE, None, v0=rho_phys)
    mg = vo.OracleMG(o, 3, nthreads=8)
    f = o.build_load_vector()
    u = mg.pcg(np.zeros_like(f), f, 100, 1e-6, 1, 2, True)
    assert abs(np.sum(f * u) - k["compliance"][0]) < 1e-5 * k["compliance"][0]


def test_degree2_element_analytic_properties():
    """The reference binds no degree-2 simulator, so its tests hold no numbers for it; pin the restatement on what
    the element must satisfy analytically: symmetry, rigid-body null space of dimension 6, and the exact strain
    energy of linear displacement fields (patch test)."""
    from oracle import vfem_oracle as vo
    h = np.array([0.5, 0.25, 0.4])
    E, nu = 3.0, 0.3
    lam, mu = nu * E / ((1 + nu) * (1 - 2 * nu)), E / (2 + 2 * nu)
    K = vo.q2_reference_stiffness(h, lam, mu)
    assert np.abs(K - K.T).max() < 1e-13
    w = np.linalg.eigvalsh(K)
    assert (np.abs(w) < 1e-10).sum() == 6 and w.min() > -1e-10
    loc = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), axis=-1).reshape(-1, 3)
    X = loc * 0.5 * h
    rng = np.random.default_rng(3)
    A = rng.standard_normal((3, 3))
    u = (X @ A.T).reshape(-1)                        # u(x) = A x: constant strain eps = sym(A)
    eps = 0.5 * (A + A.T)
    energy = np.prod(h) * (lam * np.trace(eps) ** 2 + 2 * mu * (eps * eps).sum())
    assert abs(u @ K @ u - energy) < 1e-12 * abs(energy)
    # translations and infinitesimal rotations
    for c in range(3):
        t = np.zeros((27, 3)); t[:, c] = 1.0
        assert np.abs(K @ t.reshape(-1)).max() < 1e-12
    W = A - A.T
    assert np.abs(K @ (X @ W.T).reshape(-1)).max() < 1e-11
    # interior nodes of a 2x2x2 patch carry no force under a linear field
    o = vo.OracleSimQ2((2, 2, 2), ([0, 0, 0], [1.0, 0.5, 0.8]), E, nu)
    o.rho[:] = 1.0
    g = np.stack(np.meshgrid(*[np.arange(5)] * 3, indexing="ij"), axis=-1).reshape(-1, 3) * 0.5 * o.h
    f = o.apply_k(g @ A.T).reshape(5, 5, 5, 3)
    assert np.abs(f[1:4, 1:4, 1:4]).max() < 1e-12


@pytest.mark.parametrize("N,ne,dom,bc", [(2, (16, 8), ([0, 0], [2, 1]), "2d/mbb_beam.bc"),
                                        (3, (8, 4, 4), ([0, 0, 0], [2, 1, 1]), "3d/cantilever_flexion.bc")])
def test_generic_oracle_equals_element_loop_oracle_at_degree_1(N, ne, dom, bc):
    """The sparse-matrix generic restatement (used to check the degree-2 and 2-D HIP paths) must agree with the
    element-loop restatement that reproduces the reference's logs, operator by operator, at degree 1."""
    from oracle import generic_oracle as go
    from oracle import vfem_oracle as vo
    import helpers
    bcp = os.path.join(helpers.ROOT, "bcs", bc)
    if not os.path.exists(bcp):
        pytest.skip("no such golden BC file")
    a = vo.OracleSim(dom, ne)
    a.read_material(helpers.MATERIAL)
    a.apply_bc_file(bcp)
    rho = np.random.default_rng(2).uniform(0.1, 1, size=a.num_elems)
    a.Emin = 1e-4
    a.set_densities(rho)
    g = go.GenericSim(N, 1, dom, ne, 1.0, 0.3)
    g.Emin = 1e-4
    g.rho = rho.copy()
    g.apply_bc_file(bcp)
    assert np.abs(g.K0 - a.K0).max() < 1e-14
    assert np.array_equal(g.mask, a.dmask != 0)
    assert np.abs(g.loads - a.build_load_vector()).max() < 1e-15
    u = np.random.default_rng(3).standard_normal((a.num_nodes, N))
    assert np.abs(g.apply_k(u) - a.apply_k(u)).max() < 1e-12
    assert np.abs(g.compliance_gradient(u) - a.compliance_gradient(u)).max() < 1e-12
    ma, mg = vo.OracleMG(a, 2), go.GenericMG(g, 2)
    ma.update_element_stiffness()
    mg.update_element_stiffness()
    for l in range(3):
        assert np.array_equal(mg.sims[l].mask, ma.sims[l].dmask != 0)
        v = np.random.default_rng(l).standard_normal((ma.sims[l].num_nodes, N))
        assert np.abs(mg.apply_k(l, v) - ma.apply_k(l, v)).max() < 1e-12
        if l < 2:
            b = np.random.default_rng(9).standard_normal(v.shape)
            x1, x2 = v.copy(), v.copy()
            ma.zero_dirichlet(l, x1), mg.zero_dirichlet(l, x2)
            for fwd in (True, False):
                ma.smoothing(l, x1, b, fwd)
                mg.smoothing(l, x2, b, fwd)
                assert np.abs(x1 - x2).max() < 1e-11 * np.abs(x1).max()
            assert np.abs(mg.restriction(l, v) - ma.restriction(l, v)).max() < 1e-13
            w = np.random.default_rng(5).standard_normal((ma.sims[l + 1].num_nodes, N))
            assert np.abs(mg.interpolation(l, w) - ma.interpolation(l, w)).max() < 1e-14
    f = a.build_load_vector()
    xa = ma.pcg(np.zeros_like(f), f, 50, 1e-8, 1, 2, True)
    xg = mg.pcg(np.zeros_like(f), f, 50, 1e-8, 1, 2, True)
    assert ma.last_iters == mg.last_iters
    assert np.abs(xa - xg).max() < 1e-9 * np.abs(xa).max()
