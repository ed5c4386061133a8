"""CPU ORACLE for the voxel-FEM hot path -- TEST INFRASTRUCTURE, not a product path.

A numpy/scipy (+ plain C loops in ``voxel_ref.c``) restatement of the algorithm the
reference (Nikronic/ndr, vendored VoxelFEM) runs on this path.  Only ``tests/``,
``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import it; the
product package ``ndr_amd`` never does.

Parity status: PINNED.  ``tests/test_oracle_kats.py`` checks this module against the
compliance values the reference itself logged (``tests/golden/reference_logs.json``,
taken from ``logs/slurm/gt/*.log``), the textbook K0 entries and the rule of the reference's
quadrature test (``VoxelFEM/tests/test_tp_gauss_quadrature.cc`` with ``tp_quadrature_{1,2,3}var_test.inl``:
every monomial a degree-D-per-axis rule must integrate exactly has the integral prod 1/(d_i+1)).  The reference C++ cannot
be compiled here (Eigen/TBB/CHOLMOD/Boost are not vendored), so there is no
``oracle/_ref`` build.

Every function cites the reference lines it follows (paths relative to the reference
checkout; TPS = VoxelFEM/TensorProductSimulator.hh, MG = VoxelFEM/MultigridSolver.hh).
"""
import ctypes
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_lib(force=False, native=False):
    """Compile voxel_ref.c -> oracle/_build/libvoxel_ref.so (gcc -O2, OpenMP): the checker the tests use.
    native=True: a second copy with the reference's own optimisation flags (VoxelFEM/CMakeLists.txt:43: -O3 -march=native
    -ffast-math ...), always rebuilt on the host it will run on -- only bench.py's cpu_baseline leg times that one."""
    out_dir = os.path.join(_HERE, "_build")
    src = os.path.join(_HERE, "voxel_ref.c")
    os.makedirs(out_dir, exist_ok=True)
    if native:
        so = os.path.join(out_dir, "libvoxel_ref_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffast-math", "-funroll-loops", "-fopenmp", "-fPIC", "-shared", "-o", so, src, "-lm"])
        return so
    so = os.path.join(out_dir, "libvoxel_ref.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_lib())
        _LIB.ref_max_threads.restype = ctypes.c_int
    return _LIB


def use_native_build(on=True):
    """bench.py only: route the C loops through the -O3 -march=native -ffast-math copy (on) or back through the -O2 checker (off)."""
    global _LIB
    _LIB = ctypes.CDLL(build_lib(native=True)) if on else None
    if _LIB is not None:
        _LIB.ref_max_threads.restype = ctypes.c_int


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _dims(ne):
    return (ctypes.c_long * 3)(*([int(x) for x in ne] + [1] * (3 - len(ne))))


# --------------------------------------------------------------------------------------
# L0 numerics: Gauss rules, Lagrange bases, elasticity tensor, reference element stiffness
# --------------------------------------------------------------------------------------

def gauss_rule(npts):
    """Gauss-Legendre points/weights on [0,1] (TensorProductQuadrature.hh:118-175)."""
    if npts == 1:
        return np.array([0.5]), np.array([1.0])
    if npts == 2:
        return np.array([0.21132486540518711775, 0.78867513459481288225]), np.array([0.5, 0.5])
    if npts == 3:
        return (np.array([0.11270166537925831148, 0.5, 0.88729833462074168852]),
                np.array([5 / 18.0, 8 / 18.0, 5 / 18.0]))
    if npts == 4:
        return (np.array([0.06943184420297371239, 0.33000947820757186760,
                          0.66999052179242813240, 0.93056815579702628761]),
                np.array([0.1739274225687269287, 0.326072577431273071,
                          0.326072577431273071, 0.1739274225687269287]))
    if npts == 5:
        return (np.array([0.04691007703066800360, 0.23076534494715845448, 0.5,
                          0.76923465505284154552, 0.95308992296933199640]),
                np.array([0.1184634425280945438, 0.239314335249683234, 64 / 225.0,
                          0.239314335249683234, 0.1184634425280945438]))
    raise ValueError("no rule")


def gauss_npts_for_degree(degree):
    """UnivariateGaussQuadrature<Degree> = GaussQuadratureRule<Degree/2 + 1>
    (TensorProductQuadrature.hh:25-26)."""
    return degree // 2 + 1


def integrate_tensor(f, degrees):
    """TensorProductQuadrature<degrees...>::integrate on [0,1]^N (TensorProductQuadrature.hh:33-57)."""
    rules = [gauss_rule(gauss_npts_for_degree(d)) for d in degrees]
    total = 0.0
    for idx in np.ndindex(*[len(r[0]) for r in rules]):
        w = 1.0
        p = []
        for d, i in enumerate(idx):
            w *= rules[d][1][i]
            p.append(rules[d][0][i])
        total = total + w * f(*p)
    return total


def lagrange(deg, i, x):
    """i-th degree-`deg` Lagrange polynomial on equispaced nodes j/deg (LagrangePolynomial.hh:9,42-56)."""
    xs = [j / deg for j in range(deg + 1)]
    v = 1.0
    for j in range(deg + 1):
        if j != i:
            v *= (x - xs[j]) / (xs[i] - xs[j])
    return v


def dlagrange(deg, i, x):
    xs = [j / deg for j in range(deg + 1)]
    s = 0.0
    for k in range(deg + 1):
        if k == i:
            continue
        t = 1.0 / (xs[i] - xs[k])
        for j in range(deg + 1):
            if j != i and j != k:
                t *= (x - xs[j]) / (xs[i] - xs[j])
        s += t
    return s


def lame(young, poisson, N):
    """ElasticityTensor::setIsotropic (MeshFEM ElasticityTensor.hh:100-115): 3-D Lame; 2-D = plane stress."""
    lam = poisson * young / ((1.0 + poisson) * (1.0 - 2.0 * poisson))
    mu = young / (2.0 + 2.0 * poisson)
    if N == 2:
        lam = poisson * young / (1.0 - poisson * poisson)
    return lam, mu


def element_stiffness(degrees, h, lam, mu):
    """Reference element stiffness at density 1.

    Element_T::Stiffness + m_updateK0 (TPS.hh:127-140, 1358-1366): upper triangle by
    tensor Gauss quadrature of degree 2*deg per axis, scaled by the element volume, then
    mirrored.  Strains of the vector basis function (node n, component a) are
    sym(e_a (x) grad N_n) with physical gradients (1/h_d) (TensorProductPolynomialInterpolant.hh:161-264).
    dof order = N*node + comp, local node index last-axis-fastest.
    """
    N = len(degrees)
    h = np.asarray(h, dtype=np.float64)
    shape = [d + 1 for d in degrees]
    nodes = list(np.ndindex(*shape))
    npe = len(nodes)
    ks = N * npe

    def grad(n, p):
        g = np.zeros(N)
        for d in range(N):
            v = 1.0
            for e in range(N):
                v *= dlagrange(degrees[e], nodes[n][e], p[e]) if e == d else lagrange(degrees[e], nodes[n][e], p[e])
            g[d] = v / h[d]
        return g

    def strain(n, a, p):
        g = grad(n, p)
        eps = np.zeros((N, N))
        eps[a, :] += 0.5 * g
        eps[:, a] += 0.5 * g
        return eps

    qdeg = [2 * d for d in degrees]
    K = np.zeros((ks, ks))
    for i in range(ks):
        for j in range(i, ks):
            ni, ai = divmod(i, N)
            nj, aj = divmod(j, N)

            def f(*p):
                ei = strain(ni, ai, p)
                ej = strain(nj, aj, p)
                return lam * np.trace(ei) * np.trace(ej) + 2.0 * mu * np.sum(ei * ej)
            K[i, j] = integrate_tensor(f, qdeg)
    K *= np.prod(h)
    K = np.triu(K) + np.triu(K, 1).T
    return K


# --------------------------------------------------------------------------------------
# Boundary conditions / material files
# --------------------------------------------------------------------------------------

def read_material(path, N):
    """Materials::Constant<N>(path) for {"type":"isotropic_material","young","poisson"}."""
    with open(path) as fh:
        m = json.load(fh)
    if m.get("type", "isotropic_material") != "isotropic_material":
        raise RuntimeError("only isotropic_material is supported")
    return lame(float(m["young"]), float(m["poisson"]), N)


def parse_bc_file(path, N):
    """readBoundaryConditions (MeshFEM BoundaryConditions.cc:255-370), region-box subset:
    returns [(kind, mask(N bools), value(N), minCorner_rel, maxCorner_rel, relative?)]"""
    with open(path) as fh:
        cfg = json.load(fh)
    out = []
    for reg in cfg["regions"]:
        t = reg["type"]
        mask = [True] * 3
        if t.startswith("dirichlet"):
            comps = ""
            for ch in t[9:]:
                if ch < "x" or ch > "z":
                    break
                comps += ch
            rest = t[9 + len(comps):]
            if rest != "":
                raise RuntimeError('Illegal constraint type, only "dirichlet" and "force" accepted')
            if comps:
                mask = [c in comps for c in "xyz"]
            kind = "dirichlet"
        elif t == "force":
            kind = "force"
        else:
            raise RuntimeError('Illegal constraint type, only "dirichlet" and "force" accepted')
        if "box%" in reg:
            rel = True
            box = reg["box%"]
        elif "box" in reg:
            rel = False
            box = reg["box"]
        else:
            raise RuntimeError("only box regions are supported")
        lo = np.array(box["minCorner"], dtype=np.float64)[:N]
        hi = np.array(box["maxCorner"], dtype=np.float64)[:N]
        val = np.array(reg["value"], dtype=np.float64)[:N]
        out.append((kind, np.array(mask[:N]), val, lo, hi, rel))
    return out


# --------------------------------------------------------------------------------------
# TensorProductSimulator restatement (degree 1; degree 2 only for K0)
# --------------------------------------------------------------------------------------

class OracleSim:
    """TensorProductSimulator<1,..,1> (TPS.hh:219-1419) -- grid, BCs, K0, SIMP, applyK,
    complianceGradient, direct solve."""

    def __init__(self, domain, ne, lam_mu=None):
        self.N = len(ne)
        self.ne = np.array(ne, dtype=np.int64)
        self.nn = self.ne + 1                                       # TPS.hh:267
        self.bbmin = np.array(domain[0], dtype=np.float64)
        self.bbmax = np.array(domain[1], dtype=np.float64)
        self.h = (self.bbmax - self.bbmin) / self.ne                # TPS.hh:287
        self.num_nodes = int(np.prod(self.nn))
        self.num_elems = int(np.prod(self.ne))
        self.E0, self.Emin, self.gamma = 1.0, 1e-9, 3.0             # TPS.hh:1392-1394
        self.rho = np.zeros(self.num_elems)
        self.dmask = np.zeros((self.num_nodes, self.N), dtype=np.uint8)
        self.dvals = np.zeros((self.num_nodes, self.N))
        self.force = np.zeros((self.num_nodes, self.N))
        self.has_force = np.zeros(self.num_nodes, dtype=bool)
        self.Ke = None                                              # cached custom element matrices (coarse levels)
        self.lam_mu = None
        self.K0 = None
        if lam_mu is None:
            lam_mu = (0.0, 0.5)        # ETensor(1, 0) default, TPS.hh:1379 (identity tensor: lambda 0, mu 1/2)
        self.set_lame(*lam_mu)
        self._lu = None

    # --- material ---
    def set_lame(self, lam, mu):
        self.lam_mu = (lam, mu)
        self.K0 = element_stiffness([1] * self.N, self.h, lam, mu)   # m_updateK0, TPS.hh:1358-1366

    def read_material(self, path):
        self.set_lame(*read_material(path, self.N))

    # --- geometry ---
    def node_positions(self):
        idx = np.stack(np.meshgrid(*[np.arange(n) for n in self.nn], indexing="ij"), axis=-1).reshape(-1, self.N)
        spacing = (self.bbmax - self.bbmin) / (self.nn - 1.0)       # TPS.hh:275-278
        return self.bbmin + idx * spacing

    # --- BCs: applyDisplacementsAndLoads, TPS.hh:358-409 ---
    def apply_bc_file(self, path):
        pos = self.node_positions()
        size = self.bbmax - self.bbmin
        for kind, mask, val, lo, hi, rel in parse_bc_file(path, self.N):
            if rel:                                                 # bbox.interpolatePoint
                lo = self.bbmin + lo * size
                hi = self.bbmin + hi * size
            inside = np.all((pos >= lo) & (pos <= hi), axis=1)      # BBox::containsPoint, Geometry.hh:276-278
            cnt = int(inside.sum())
            if kind == "force":
                if cnt == 0:
                    raise RuntimeError("Force constraint region unmatched")
                self.has_force[inside] = True
                self.force[inside] = val / cnt                      # TPS.hh:383-388 (setForce: overwrite)
            else:
                if cnt == 0:
                    raise RuntimeError("Dirichlet region unmatched")
                for c in range(self.N):
                    if not mask[c]:
                        continue
                    already = inside & (self.dmask[:, c] != 0)
                    if np.any(np.abs(self.dvals[already, c] - val[c]) > 1e-10):
                        raise RuntimeError("Conflicting dirichlet displacements.")
                    new = inside & (self.dmask[:, c] == 0)
                    self.dmask[new, c] = 1
                    self.dvals[new, c] = val[c]
                self._lu = None

    def build_load_vector(self):
        f = np.zeros((self.num_nodes, self.N))                      # TPS.hh:893-901
        f[self.has_force] += self.force[self.has_force]
        return f

    # --- densities / SIMP ---
    def set_uniform_densities(self, v):
        if v > 1.0 or v < 0:
            raise RuntimeError("Density value (%f) has to be in between 0 and 1" % v)
        self.rho[:] = v
        self._lu = None

    def set_densities(self, rho):
        rho = np.asarray(rho, dtype=np.float64).reshape(-1)
        if rho.size != self.num_elems:
            raise RuntimeError("size mismatch")
        self.rho = rho.copy()
        self._lu = None

    def young(self):
        E = np.empty(self.num_elems)
        lib().ref_simp(ctypes.c_long(self.num_elems), _p(np.ascontiguousarray(self.rho)),
                       ctypes.c_double(self.E0), ctypes.c_double(self.Emin), ctypes.c_double(self.gamma), _p(E))
        return E

    def set_cached_ke(self, Ke):
        self.Ke = Ke                                                # cacheCustomElementStiffnessMatrices, TPS.hh:773-776
        self._lu = None

    # --- operators ---
    def apply_k(self, u, nthreads=1):
        u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.empty_like(u)
        if self.Ke is not None:
            lib().ref_apply_k(self.N, _dims(self.ne), 1, _p(self.Ke), None, _p(u), _p(out), nthreads)
        else:
            E = self.young()
            lib().ref_apply_k(self.N, _dims(self.ne), 0, _p(self.K0), _p(E), _p(u), _p(out), nthreads)
        return out

    def compliance_gradient(self, u, nthreads=1):
        u = np.ascontiguousarray(u, dtype=np.float64)
        g = np.empty(self.num_elems)
        lib().ref_compliance_gradient(self.N, _dims(self.ne), _p(self.K0), _p(np.ascontiguousarray(self.rho)),
                                      ctypes.c_double(self.E0), ctypes.c_double(self.Emin),
                                      ctypes.c_double(self.gamma), _p(u), _p(g), nthreads)
        return g

    def gs_sweep(self, u, b, forward=True, nthreads=1):
        """smoothingMulticoloredGS (MG.hh:336-340); u updated in place."""
        assert u.flags.c_contiguous and u.dtype == np.float64
        b = np.ascontiguousarray(b, dtype=np.float64)
        if self.Ke is not None:
            lib().ref_gs_sweep(self.N, _dims(self.ne), 1, _p(self.Ke), None, _p(u), _p(b), _p(self.dmask),
                               int(forward), nthreads)
        else:
            E = self.young()
            lib().ref_gs_sweep(self.N, _dims(self.ne), 0, _p(self.K0), _p(E), _p(u), _p(b), _p(self.dmask),
                               int(forward), nthreads)

    # --- assembly + direct solve (TPS.hh:590-625, 834-865; CHOLMOD replaced by SuperLU) ---
    def element_dofs(self):
        nstr = np.ones(self.N, dtype=np.int64)
        for d in range(self.N - 2, -1, -1):
            nstr[d] = nstr[d + 1] * self.nn[d + 1]
        eidx = np.stack(np.meshgrid(*[np.arange(n) for n in self.ne], indexing="ij"), axis=-1).reshape(-1, self.N)
        first = eidx @ nstr
        loc = np.array(list(np.ndindex(*([2] * self.N)))) @ nstr
        nodes = first[:, None] + loc[None, :]                       # [ne, npe]
        dofs = (self.N * nodes[:, :, None] + np.arange(self.N)[None, None, :]).reshape(self.num_elems, -1)
        return nodes, dofs

    def constant_strain_load(self, eps):
        """TPS::constantStrainLoad (TPS.hh:792-821): per element l(i, j) = rho_e * vol * int strain(node j, comp i) : (C : eps)
        by the element's own quadrature (Element::constantStressLoad, TPS.hh:145-160; the RAW density scales it, :172),
        accumulated to the element's nodes in element order."""
        N = self.N
        lam, mu = self.lam_mu
        eps = np.asarray(eps, dtype=np.float64).reshape(N, N)
        sig = lam * np.trace(eps) * np.eye(N) + 2.0 * mu * 0.5 * (eps + eps.T)        # E_tensor.doubleContract(cstrain)
        loc = list(np.ndindex(*([2] * N)))
        l = np.zeros((N, len(loc)))
        for j, nj in enumerate(loc):
            for i in range(N):
                def f(*p):
                    g = np.zeros(N)
                    for d in range(N):
                        v = 1.0
                        for e in range(N):
                            v *= dlagrange(1, nj[e], p[e]) if e == d else lagrange(1, nj[e], p[e])
                        g[d] = v / self.h[d]
                    st = np.zeros((N, N))
                    st[i, :] += 0.5 * g
                    st[:, i] += 0.5 * g
                    return np.sum(st * sig)
                l[i, j] = integrate_tensor(f, [1] * N)
        l *= np.prod(self.h)
        nodes, _ = self.element_dofs()
        F = np.zeros((self.num_nodes, N))
        for ei in range(self.num_elems):
            for j in range(len(loc)):
                F[nodes[ei, j]] += self.rho[ei] * l[:, j]
        return F

    def assemble(self):
        import scipy.sparse as sp
        _, dofs = self.element_dofs()
        ks = dofs.shape[1]
        if self.Ke is not None:
            vals = self.Ke.reshape(self.num_elems, ks, ks)
        else:
            vals = self.young()[:, None, None] * self.K0[None, :, :]
        rows = np.repeat(dofs[:, :, None], ks, axis=2)
        cols = np.repeat(dofs[:, None, :], ks, axis=1)
        n = self.num_nodes * self.N
        return sp.coo_matrix((vals.ravel(), (rows.ravel(), cols.ravel())), shape=(n, n)).tocsc()

    def solve(self, f):
        """TPS::solve (TPS.hh:834-865): remove fixed rows/cols, factorise, solve, fixed dofs = 0."""
        import scipy.sparse.linalg as spla
        if np.any(self.dvals[self.dmask != 0] != 0):
            raise RuntimeError("Nonzero Dirichlet constraints currently unsupported")
        free = np.flatnonzero(self.dmask.reshape(-1) == 0)
        if self._lu is None:
            K = self.assemble()
            self._lu = spla.splu(K[free][:, free].tocsc())
        x = np.zeros(self.num_nodes * self.N)
        x[free] = self._lu.solve(np.asarray(f, dtype=np.float64).reshape(-1)[free])
        return x.reshape(self.num_nodes, self.N)


# --------------------------------------------------------------------------------------
# MultigridSolver restatement
# --------------------------------------------------------------------------------------

class OracleMG:
    """MultigridSolver<1,..,1> (MG.hh:11-759)."""

    def __init__(self, fine, num_levels, nthreads=1):
        self.sims = [fine]
        self.nthreads = nthreads
        self.symmetric_gs = True                                    # MG.hh:758
        ne = fine.ne.copy()
        for l in range(1, num_levels + 1):
            if np.any(ne % 2 == 1):
                raise RuntimeError("Grid size currently must be divisible by 2^numCoarseningLevels "
                                   "(nonuniform coarsening not yet implemented)")
            ne = ne // 2
            c = OracleSim((fine.bbmin, fine.bbmax), ne, fine.lam_mu)
            self._coarsen_dirichlet(self.sims[-1], c)
            self.sims.append(c)
        self.x = [np.zeros((s.num_nodes, s.N)) for s in self.sims]
        self.b = [np.zeros((s.num_nodes, s.N)) for s in self.sims]

    @staticmethod
    def _coarsen_dirichlet(finer, coarser):
        """MG.hh:57-84: a fine Dirichlet node lying on a coarse element vertex/edge/face constrains
        every coarse node of that entity (values zero)."""
        N = finer.N
        fidx = np.stack(np.meshgrid(*[np.arange(n) for n in finer.nn], indexing="ij"), axis=-1).reshape(-1, N)
        cm = coarser.dmask.reshape(tuple(coarser.nn) + (N,))
        for nf in np.flatnonzero(finer.dmask.any(axis=1)):
            gi = fidx[nf]
            e = np.minimum(gi // 2, coarser.ne - 1)
            loc = gi - 2 * e                                        # 0, 1 (interior) or 2
            onb = np.where(loc == 0, 0, np.where(loc == 2, 1, -1))
            if np.all(onb < 0):
                raise RuntimeError("Dirichlet constraints on internal nodes are not supported")
            sl = tuple(slice(e[d], e[d] + 2) if onb[d] < 0 else slice(e[d] + onb[d], e[d] + onb[d] + 1)
                       for d in range(N))
            for c in range(N):
                if finer.dmask[nf, c]:
                    cm[sl + (c,)] = 1

    # --- small helpers ---
    def zero_dirichlet(self, l, u):
        u[self.sims[l].dmask != 0] = 0.0                            # MG.hh:364-378
        return u

    def enforce_dirichlet(self, l, u, zero):
        s = self.sims[l]                                            # MG.hh:386-398
        m = s.dmask != 0
        u[m] = 0.0 if zero else s.dvals[m]
        return u

    def apply_k(self, l, u):
        return self.sims[l].apply_k(u, self.nthreads)               # MG.hh:353-358

    def residual(self, l, u, b):
        return self.zero_dirichlet(l, b - self.apply_k(l, u))      # MG.hh:401-413

    def smoothing(self, l, u, b, forward=True):
        self.sims[l].gs_sweep(u, b, forward, self.nthreads)         # MG.hh:336-340

    def restriction(self, l, fine_vals):
        c = self.sims[l + 1]                                        # MG.hh:146-161
        out = np.empty((c.num_nodes, c.N))
        lib().ref_restrict(c.N, _dims(c.ne), c.N, _p(np.ascontiguousarray(fine_vals)), _p(out))
        return out

    def interpolation(self, l, coarse_vals, out=None, accumulate=False):
        f, c = self.sims[l], self.sims[l + 1]                       # MG.hh:116-141
        if out is None:
            out = np.zeros((f.num_nodes, f.N))
        lib().ref_prolong(c.N, _dims(c.ne), c.N, _p(np.ascontiguousarray(coarse_vals)), _p(out),
                          int(accumulate), self.nthreads)
        return out

    def update_element_stiffness(self):
        """updateElementStiffnessMatrices -> buildPESCoarse (MG.hh:415-425, 604-669)."""
        for l in range(1, len(self.sims)):
            finer, coarser = self.sims[l - 1], self.sims[l]
            ks = finer.K0.shape[0]
            Kec = np.empty((coarser.num_elems, ks * ks))
            if finer.Ke is None:
                E = finer.young()
                lib().ref_coarsen_ke(coarser.N, _dims(coarser.ne), 0, _p(np.ascontiguousarray(finer.K0)), _p(E),
                                     _p(Kec), self.nthreads)
            else:
                lib().ref_coarsen_ke(coarser.N, _dims(coarser.ne), 1, None, _p(finer.Ke), _p(Kec), self.nthreads)
            coarser.set_cached_ke(Kec)

    # --- cycles ---
    def vcycle(self, l, nsmooth, residual_system):
        """MG.hh:516-553."""
        coarsest = len(self.sims) - 1
        if l == coarsest:
            self.x[l] = self.sims[l].solve(self.b[l])
            return
        self.enforce_dirichlet(l, self.x[l], residual_system)
        for _ in range(nsmooth):
            self.smoothing(l, self.x[l], self.b[l], True)
        self.b[l + 1] = self.restriction(l, self.residual(l, self.x[l], self.b[l]))
        self.x[l + 1] = np.zeros_like(self.x[l + 1])
        self.vcycle(l + 1, nsmooth, True)
        self.interpolation(l, self.x[l + 1], out=self.x[l], accumulate=True)
        for _ in range(nsmooth):
            self.smoothing(l, self.x[l], self.b[l], not self.symmetric_gs)

    def full_multigrid(self, l, nsmooth, residual_system):
        """MG.hh:486-508."""
        coarsest = len(self.sims) - 1
        if l == coarsest:
            self.x[l] = self.sims[l].solve(self.b[l])
            return
        self.b[l + 1] = self.restriction(l, self.b[l])
        self.full_multigrid(l + 1, nsmooth, residual_system)
        self.x[l] = self.interpolation(l, self.x[l + 1])
        self.vcycle(l, nsmooth, residual_system)

    def solve(self, u, f, num_steps, nsmooth, stiffness_updated=False, zero_dirichlet=False, fmg=False):
        """MG.hh:447-472."""
        if not stiffness_updated:
            self.update_element_stiffness()
        if num_steps == 0:
            return u
        self.x[0] = np.array(u, dtype=np.float64, copy=True)
        self.b[0] = np.array(f, dtype=np.float64, copy=True)
        if fmg:
            self.full_multigrid(0, nsmooth, zero_dirichlet)
            for _ in range(1, num_steps):
                self.vcycle(0, nsmooth, zero_dirichlet)
        else:
            for _ in range(num_steps):
                self.vcycle(0, nsmooth, zero_dirichlet)
        return self.x[0]

    def apply_preconditioner_inv(self, r, num_steps, nsmooth, fmg):
        if nsmooth == 0:
            return r                                                # MG.hh:476-479
        return self.solve(np.zeros_like(r), r, num_steps, nsmooth, True, True, fmg).copy()

    def pcg(self, u, b, max_iter, tol, mg_iterations=1, mg_smoothing=1, fmg=False, callback=None):
        """preconditionedConjugateGradient (MG.hh:679-732).  The reference's loop counter is
        uninitialised (MG.hh:710); it is started at 0 here."""
        x = np.array(u, dtype=np.float64, copy=True)
        self.enforce_dirichlet(0, x, False)
        self.update_element_stiffness()
        b_norm_sq = float(np.sum(b * b))
        r = self.residual(0, x, b)
        rMr = 0.0
        d = None
        i = 0
        history = []
        while i < max_iter and float(np.sum(r * r)) > tol * tol * b_norm_sq:
            i += 1
            s = self.apply_preconditioner_inv(r, mg_iterations, mg_smoothing, fmg)
            self.zero_dirichlet(0, s)
            rMr_old = rMr
            rMr = float(np.sum(r * s))
            d = s if d is None else s + (rMr / rMr_old) * d
            Ad = self.zero_dirichlet(0, self.apply_k(0, d))
            alpha = rMr / float(np.sum(d * Ad))
            x += alpha * d
            r -= alpha * Ad
            history.append(float(np.sqrt(np.sum(r * r))))
            if callback:
                callback(i, x, r)
        self.last_iters = i
        self.last_history = history
        return x


# --------------------------------------------------------------------------------------
# L2: objective, filters, constraint, problem, OC (TopologyOptimization*.hh, OptimalityCriterion.hh)
# --------------------------------------------------------------------------------------

class OracleComplianceObjective:
    """ComplianceObjective (TopologyOptimizationObjective.hh:24-63): direct solve."""

    def __init__(self, sim, skip_solve=False):
        self.sim = sim
        self.f = sim.build_load_vector()
        self.u = np.zeros_like(self.f)
        if not skip_solve:
            self.update_cache(sim.rho)

    def compliance(self):
        return 0.5 * float(np.sum(self.f * self.u))

    def gradient(self):
        return self.sim.compliance_gradient(self.u)

    def update_cache(self, x_phys):
        self.sim.set_densities(x_phys)
        self.u = self.sim.solve(self.f)


class OracleMGComplianceObjective(OracleComplianceObjective):
    """MultigridComplianceObjective (TopologyOptimizationObjective.hh:67-102)."""

    def __init__(self, mg):
        self.mg = mg
        self.cgIter, self.tol = 100, 1e-5
        self.mgIterations, self.mgSmoothingIterations = 1, 2
        self.fullMultigrid, self.zeroInit = True, False
        super().__init__(mg.sims[0], skip_solve=True)
        self.update_cache(self.sim.rho)

    def update_cache(self, x_phys):
        self.sim.set_densities(x_phys)
        if self.zeroInit:
            self.u = np.zeros_like(self.u)
        self.u = self.mg.pcg(self.u, self.f, self.cgIter, self.tol, self.mgIterations,
                             self.mgSmoothingIterations, self.fullMultigrid)


def smoothing_matrix(ne, radius=1):
    """SmoothingFilter::updateMatrix (TopologyOptimizationFilter.hh:133-150): box (2r+1)^N
    neighbourhood clipped to the grid, each row = 1/numInStencil."""
    import scipy.sparse as sp
    ne = np.asarray(ne, dtype=np.int64)
    N = len(ne)
    n = int(np.prod(ne))
    idx = np.stack(np.meshgrid(*[np.arange(k) for k in ne], indexing="ij"), axis=-1).reshape(-1, N)
    estr = np.ones(N, dtype=np.int64)
    for d in range(N - 2, -1, -1):
        estr[d] = estr[d + 1] * ne[d + 1]
    rows, cols = [], []
    r = int(radius)
    for off in np.ndindex(*([2 * r + 1] * N)):
        o = np.array(off) - r
        nb = idx + o
        ok = np.all((nb >= 0) & (nb < ne), axis=1)
        rows.append(np.flatnonzero(ok))
        cols.append(nb[ok] @ estr)
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    A = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(n, n)).tocsr()
    cnt = np.asarray(A.sum(axis=1)).ravel()
    return sp.diags(1.0 / cnt) @ A


class OracleSmoothingFilter:
    def __init__(self, radius=1):
        self.radius = radius
        self.A = None

    def set_grid(self, ne):
        self.A = smoothing_matrix(ne, self.radius)

    def apply(self, x):
        return self.A @ x

    def backprop(self, g, vars_):
        return self.A.T @ g


class OracleProjectionFilter:
    """ProjectionFilter (TopologyOptimizationFilter.hh:55-79)."""

    def __init__(self, beta=1.0):
        self.beta = beta

    def set_grid(self, ne):
        pass

    def apply(self, x):
        b = self.beta
        return 0.5 * (np.tanh(0.5 * b) + np.tanh(b * (x - 0.5))) / np.tanh(0.5 * b)

    def backprop(self, g, vars_):
        b = self.beta
        t = np.tanh(b * (vars_ - 0.5))
        return g * 0.5 * b * (1.0 - t * t) / np.tanh(0.5 * b)


class OracleVolumeConstraint:
    """TotalVolumeConstraint (TopologyOptimizationConstraint.hh:21-34)."""

    def __init__(self, v):
        self.v = float(v)

    def evaluate(self, x):
        return 1.0 - float(np.mean(x)) / self.v

    def backprop(self, x):
        return np.full(x.size, -1.0 / (self.v * x.size))


class OracleProblem:
    """TopologyOptimizationProblem (TopologyOptimizationProblem.hh:18-204)."""

    def __init__(self, sim, objective, constraints, filters):
        self.sim, self.objective, self.constraints, self.filters = sim, objective, constraints, filters
        for f in filters:
            f.set_grid(sim.ne)
        self.cached = [np.zeros(sim.num_elems) for _ in range(len(filters) + 1)]
        self.vars_set = False

    def set_vars(self, x, force=False):
        x = np.asarray(x, dtype=np.float64)
        if not force and self.vars_set and np.linalg.norm(x - self.cached[0]) < 1e-16:
            return False
        self.cached[0] = x.copy()
        for i, f in enumerate(self.filters):
            self.cached[i + 1] = f.apply(self.cached[i])
        self.objective.update_cache(self.cached[-1])
        self.vars_set = True
        return True

    def evaluate_oc_constraint(self, x):
        for f in self.filters:                                      # Problem.hh:73-85
            x = f.apply(x)
        return self.constraints[0].evaluate(x)

    def evaluate_objective(self):
        return self.objective.compliance()

    def evaluate_objective_gradient(self):
        g = self.objective.gradient()                               # Problem.hh:98-113
        nf = len(self.filters)
        for i in range(nf):
            g = self.filters[nf - 1 - i].backprop(g, self.cached[nf - 1 - i])
        return g

    def evaluate_constraints(self):
        return np.array([c.evaluate(self.cached[-1]) for c in self.constraints])

    def evaluate_constraints_jacobian(self):
        nf = len(self.filters)
        rows = []
        for c in self.constraints:
            d = c.backprop(self.cached[-1])
            for i in range(nf):
                d = self.filters[nf - 1 - i].backprop(d, self.cached[nf - 1 - i])
            rows.append(d)
        return np.array(rows)


class OracleOC:
    """OCOptimizer (OptimalityCriterion.hh:30-81); the multiplier bracket persists across steps."""

    def __init__(self, problem):
        self.p = problem
        self.lmin, self.lmax = 1.0, 2.0

    def step(self, m=0.2, ctol=1e-6):
        p = self.p
        dJ = p.evaluate_objective_gradient()
        dc = p.evaluate_constraints_jacobian()[0]
        x0 = p.cached[0].copy()

        def stepped(lam):
            return np.minimum(np.minimum(np.maximum(np.maximum(x0 * np.sqrt(dJ / (dc * lam)), x0 - m), 0.0),
                                         x0 + m), 1.0)

        def ceval(lam):
            return p.evaluate_oc_constraint(stepped(lam))

        while ceval(self.lmin) > 0:
            self.lmax = self.lmin
            self.lmin /= 2
        while ceval(self.lmax) < 0:
            self.lmin = self.lmax
            self.lmax *= 2
        mid = 0.5 * (self.lmin + self.lmax)
        vol = ceval(mid)
        while abs(vol) > ctol:
            if vol < 0:
                self.lmin = mid
            if vol > 0:
                self.lmax = mid
            mid = 0.5 * (self.lmin + self.lmax)
            vol = ceval(mid)
        p.set_vars(stepped(mid))
        return p.evaluate_objective(), p.evaluate_constraints()[0], mid


# --------------------------------------------------------------------------------------
# Fourier-feature MLP (networks.py:128-185), float32
# --------------------------------------------------------------------------------------

def mlp_forward(coords, B, weights, biases, sigmoid_out=False):
    """networks.MLP.forward (networks.py:181-185): x -> [sin(2 pi x B^T), cos(2 pi x B^T)] ->
    (Linear+ReLU)* -> Linear [-> Sigmoid]; all float32 like torch CPU."""
    x = np.asarray(coords, dtype=np.float32).reshape(-1, coords.shape[-1])
    proj = (np.float32(2.0 * np.pi) * x) @ B.T.astype(np.float32)
    h = np.concatenate([np.sin(proj), np.cos(proj)], axis=-1).astype(np.float32)
    nl = len(weights)
    for i, (W, b) in enumerate(zip(weights, biases)):
        h = h @ W.T.astype(np.float32) + b.astype(np.float32)
        if i < nl - 1:
            h = np.maximum(h, np.float32(0))
    if sigmoid_out:
        h = (1.0 / (1.0 + np.exp(-h.astype(np.float64)))).astype(np.float32)
    return h


def get_mgrid(sidelen, domain=None):
    """utils.get_mgrid (utils.py:35-53): linspace including both ends per axis, 'ij' meshgrid,
    float32, shape [1, n0, n1, (n2), N]."""
    N = len(sidelen)
    if domain is None:
        domain = [[0.0, 1.0]] * N
    axes = [np.linspace(domain[d][0], domain[d][1], sidelen[d], dtype=np.float32) for d in range(N)]
    g = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1)
    return g[None].astype(np.float32)


# ======================================================================================================
# Degree-2 elements: TensorProductSimulator<2,2,2> (TPS.hh:97-110 with Degrees = 2,2,2).  The reference's python
# bindings leave this instantiation out (VoxelFEM.cc:226-229), so it holds no golden numbers for it; this restatement
# is pinned by the analytic properties of the element (rigid-body null space, linear patch test, constant-strain
# energy) in tests/test_oracle_kats.py.
# ======================================================================================================
def _lagrange2(x):
    """values and derivatives of the three degree-2 Lagrange polynomials on nodes 0, 1/2, 1 (TensorProductPolynomialInterpolant.hh)."""
    x = np.asarray(x, dtype=np.float64)
    N = np.stack([2 * (x - 0.5) * (x - 1), -4 * x * (x - 1), 2 * x * (x - 0.5)])
    dN = np.stack([4 * x - 3, -8 * x + 4, 4 * x - 1])
    return N, dN


def q2_reference_stiffness(h, lam, mu):
    """81 x 81 full-density element stiffness of the 27-node hexahedron with edge lengths h, isotropic (lam, mu);
    local node 9a+3b+c, dof 3*node+component; Gauss-Legendre 3 points per axis (exact for the degree-4 integrand)."""
    xg, wg = np.polynomial.legendre.leggauss(3)
    xg, wg = 0.5 * (xg + 1.0), 0.5 * wg
    N, dN = _lagrange2(xg)                                  # [3 basis, 3 points]
    K = np.zeros((81, 81))
    vol = float(np.prod(h))
    for qa in range(3):
        for qb in range(3):
            for qc in range(3):
                w = wg[qa] * wg[qb] * wg[qc] * vol
                G = np.zeros((27, 3))
                for a in range(3):
                    for b in range(3):
                        for c in range(3):
                            n = 9 * a + 3 * b + c
                            G[n, 0] = dN[a, qa] * N[b, qb] * N[c, qc] / h[0]
                            G[n, 1] = N[a, qa] * dN[b, qb] * N[c, qc] / h[1]
                            G[n, 2] = N[a, qa] * N[b, qb] * dN[c, qc] / h[2]
                # strain-displacement in Voigt-free form: K[(n,i),(m,j)] = lam G[n,i] G[m,j] + mu G[n,j] G[m,i] + mu delta_ij G[n].G[m]
                GG = G @ G.T
                blk = lam * np.einsum("ni,mj->nimj", G, G) + mu * np.einsum("nj,mi->nimj", G, G) \
                    + mu * np.einsum("nm,ij->nimj", GG, np.eye(3))
                K += w * blk.reshape(81, 81)
    return K


class OracleSimQ2:
    """applyK / complianceGradient of the degree-2 simulator (TPS.hh:905-952, 730-751), numpy element loop."""

    def __init__(self, ne, domain=([0, 0, 0], [1, 1, 1]), young=1.0, poisson=0.0):
        self.ne = np.asarray(ne, dtype=np.int64)
        self.nn = 2 * self.ne + 1
        lo, hi = np.asarray(domain[0], float), np.asarray(domain[1], float)
        self.h = (hi - lo) / self.ne
        self.E0, self.Emin, self.gamma = 1.0, 1e-9, 3.0
        self.set_isotropic(young, poisson)
        self.num_elems = int(np.prod(self.ne))
        self.num_nodes = int(np.prod(self.nn))
        self.rho = np.zeros(self.num_elems)
        ei = np.stack(np.meshgrid(*[np.arange(n) for n in self.ne], indexing="ij"), axis=-1).reshape(-1, 3)
        loc = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), axis=-1).reshape(-1, 3)
        nd = 2 * ei[:, None, :] + loc[None, :, :]                                    # [ne, 27, 3]
        self.enodes = (nd[..., 0] * self.nn[1] + nd[..., 1]) * self.nn[2] + nd[..., 2]

    def set_isotropic(self, young, poisson):
        self.lam = poisson * young / ((1 + poisson) * (1 - 2 * poisson))
        self.mu = young / (2 + 2 * poisson)
        self.K0 = q2_reference_stiffness(self.h, self.lam, self.mu)

    def young(self):
        return self.Emin + self.rho ** self.gamma * (self.E0 - self.Emin)

    def apply_k(self, u):
        u = np.asarray(u, dtype=np.float64).reshape(self.num_nodes, 3)
        ue = u[self.enodes].reshape(self.num_elems, 81)
        fe = (ue @ self.K0.T) * self.young()[:, None]
        out = np.zeros((self.num_nodes, 3))
        np.add.at(out, self.enodes.reshape(-1), fe.reshape(-1, 3))
        return out

    def compliance_gradient(self, u):
        u = np.asarray(u, dtype=np.float64).reshape(self.num_nodes, 3)
        ue = u[self.enodes].reshape(self.num_elems, 81)
        en = np.einsum("ei,ij,ej->e", ue, self.K0, ue)
        return -0.5 * self.gamma * self.rho ** (self.gamma - 1) * (self.E0 - self.Emin) * en
