"""Host-side mirror of the reference's ``pyVoxelFEM`` pybind11 module for the MI355X path.

Same names, argument meaning and error behaviour as
``VoxelFEM/python_bindings/VoxelFEM.cc:41-289`` of the reference; arrays cross this boundary as
float64 numpy arrays (copied, like the Eigen conversions of the reference), the numerics run in
``libvfem.so`` (HIP kernels, include/vfem.h) on device memory held in torch tensors.
``*_device`` methods take/return torch CUDA tensors without copies.

Host-side pieces (BC/material parsing, filters, volume constraint, optimality-criterion update,
problem cache) restate ``TopologyOptimization{Problem,Filter,Constraint}.hh`` and
``OptimalityCriterion.hh``; SURVEY 8(f)-1 ranks moving them to the device as the next step.
"""
import ctypes
import json
import os
import sys
import types

import numpy as np
import torch

from . import _lib

__all__ = [
    "TensorProductSimulator", "TopologyOptimizationProblem", "ComplianceObjective",
    "MultigridComplianceObjective", "OCOptimizer", "PythonFilter", "ProjectionFilter",
    "SmoothingFilter", "LangelaarFilter", "applyFilter", "TotalVolumeConstraint",
    "benchmark_reset", "benchmark_report", "benchmark_start_timer_section",
    "benchmark_stop_timer_section", "benchmark_start_timer", "benchmark_stop_timer", "detail",
]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _to_dev(a, shape=None):
    """float64 contiguous device tensor from numpy / torch / sequence."""
    if isinstance(a, torch.Tensor):
        t = a.to(device=_dev(), dtype=torch.float64).contiguous()
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(_dev())
    if shape is not None:
        if t.numel() != int(np.prod(shape)):
            raise RuntimeError("Invalid input size")
        t = t.reshape(shape)
    return t


def _to_np(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------------------------
# material / boundary-condition files (MeshFEM Materials.cc, BoundaryConditions.cc:255-370)
# ----------------------------------------------------------------------------------------------

def _read_isotropic_material(path):
    with open(path) as fh:
        m = json.load(fh)
    if m.get("type", "isotropic_material") != "isotropic_material":
        raise RuntimeError("unsupported material type: " + str(m.get("type")))
    return float(m["young"]), float(m["poisson"])


def _parse_regions(path):
    with open(path) as fh:
        cfg = json.load(fh)
    regions = []
    for reg in cfg["regions"]:
        t = reg["type"]
        comps = "xyz"
        if t.startswith("dirichlet"):
            tail = t[len("dirichlet"):]
            n = 0
            while n < len(tail) and "x" <= tail[n] <= "z":
                n += 1
            if n > 3:
                raise RuntimeError("invalid mask")
            if n > 0:
                comps = tail[:n]
                if len(set(comps)) != len(comps):
                    raise RuntimeError("invalid component specifier: '%s'" % comps)
            if tail[n:] != "":
                raise RuntimeError('Illegal constraint type, only "dirichlet" and "force" accepted')
            kind = "dirichlet"
        elif t == "force":
            kind = "force"
        else:
            raise RuntimeError('Illegal constraint type, only "dirichlet" and "force" accepted')
        if "box%" in reg:
            relative, box = True, reg["box%"]
        elif "box" in reg:
            relative, box = False, reg["box"]
        else:
            raise RuntimeError("only box / box% regions are supported")
        regions.append((kind, comps, [float(v) for v in reg["value"]],
                        [float(v) for v in box["minCorner"]], [float(v) for v in box["maxCorner"]], relative))
    return regions


class SuiteSparseMatrix:
    """What ``TensorProductSimulator.getK()`` returns (TPS.hh:966-973): the assembled stiffness matrix in compressed-column
    form, upper triangle only (TPS.hh:590-612 accumulates ``di <= dj``), with the attribute names of MeshFEM's binding
    (python_bindings/sparse_matrices.cc:67-125): m, n, nz, Ap, Ai, Ax, symmetry_mode, trace(), apply(), toSciPy().
    Host-side and assembled with numpy: a diagnostic of small problems, not part of the device path."""

    def __init__(self, upper_csc):
        self._A = upper_csc
        self.m, self.n = upper_csc.shape
        self.nz = int(upper_csc.nnz)
        self.Ap = upper_csc.indptr.astype(np.int64)
        self.Ai = upper_csc.indices.astype(np.int64)
        self.Ax = upper_csc.data
        self.symmetry_mode = "UPPER_TRIANGLE"

    def trace(self):
        return float(self._A.diagonal().sum())

    def toSciPy(self):
        return self._A.copy()

    def full(self):
        """the symmetric matrix the upper triangle stands for"""
        import scipy.sparse as sp
        return (self._A + sp.triu(self._A, 1).T).tocsc()

    def apply(self, vec, transpose=False):
        return self.full() @ np.asarray(vec, dtype=np.float64).reshape(-1)


def _assemble_upper(elem_nodes, K0, young, ndof_per_node, num_nodes):
    """sum_e E_e G_e^T K0 G_e, entries with row <= column (TPS.hh:590-625)"""
    import scipy.sparse as sp
    nd = ndof_per_node
    dofs = (elem_nodes[:, :, None] * nd + np.arange(nd)[None, None, :]).reshape(elem_nodes.shape[0], -1)
    ke = K0.shape[0]
    n = num_nodes * nd
    A = sp.csc_matrix((n, n))
    step = max(1, (1 << 22) // (ke * ke))                  # bounded temporary: ~4 M triplets at a time
    for a in range(0, dofs.shape[0], step):
        d = dofs[a:a + step]
        rows = np.repeat(d, ke, axis=1).reshape(-1)
        cols = np.tile(d, (1, ke)).reshape(-1)
        vals = (young[a:a + step, None] * K0.reshape(1, -1)).reshape(-1)
        keep = rows <= cols
        A = A + sp.coo_matrix((vals[keep], (rows[keep], cols[keep])), shape=(n, n)).tocsc()
    A.sum_duplicates()
    A.sort_indices()
    return SuiteSparseMatrix(A)


def _constant_strain_load(eps, lam, mu, h, degree, rho_grid):
    """TPS::constantStrainLoad (TPS.hh:792-821 with Element::constantStrainLoad, :145-173): nodal load of the unit strain
    ``eps`` (N x N symmetric): every element contributes  rho_e * vol * (C : eps) . int grad(phi_j)  to its node j -- the
    RAW density scales the load (no SIMP law, TPS.hh:813).  int grad(phi_j) factorises over the axes of the tensor-product
    Lagrange basis: int phi' = phi(1) - phi(0), int phi = the Newton-Cotes weights."""
    N = len(h)
    eps = np.asarray(eps, dtype=np.float64).reshape(N, N)
    sigma = lam * np.trace(eps) * np.eye(N) + 2.0 * mu * 0.5 * (eps + eps.T)            # isotropic C : eps
    I = {1: np.array([0.5, 0.5]), 2: np.array([1.0, 4.0, 1.0]) / 6.0}[degree]
    dI = {1: np.array([-1.0, 1.0]), 2: np.array([-1.0, 0.0, 1.0])}[degree]
    vol = float(np.prod(h))
    ne = rho_grid.shape
    nn = tuple(n * degree + 1 for n in ne)
    F = torch.zeros(nn + (N,), dtype=torch.float64, device=rho_grid.device)
    for loc in np.ndindex(*([degree + 1] * N)):
        g = np.array([dI[loc[k]] / h[k] * np.prod([I[loc[m]] for m in range(N) if m != k]) for k in range(N)])
        load = torch.as_tensor(vol * (sigma @ g), dtype=torch.float64, device=rho_grid.device)
        sl = tuple(slice(loc[d], loc[d] + degree * ne[d], degree) for d in range(N))
        F[sl] += rho_grid[..., None] * load
    return F.reshape(-1, N)


def _densities_from_msh(path, field, ne, N):
    """TPS::readDensities (TPS.hh:470-509): per-element scalar field of a Gmsh file; an element's grid cell is found from
    its vertex centroid relative to the bounding box of the mesh vertices"""
    if not str(path).endswith(".msh"):
        raise RuntimeError("Material file extension" + os.path.splitext(str(path))[1] + " is not supported")
    from . import io
    parser = io.MSHFieldParser3(path)
    values = np.asarray(parser.scalarField(field), dtype=np.float64).reshape(-1)
    V, E = np.asarray(parser.vertices(), dtype=np.float64), np.asarray(parser.elements())
    nel = int(np.prod(ne))
    if E.shape[0] != nel:
        raise RuntimeError("The number of elements in the mesh : %d and the number of elements of the simulator : %d must be equal."
                           % (E.shape[0], nel))
    lo, hi = V.min(axis=0), V.max(axis=0)
    rel = (V[E].mean(axis=1) - lo) / np.where(hi > lo, hi - lo, 1.0)
    cell = np.minimum(np.floor(rel[:, :N] * np.asarray(ne)).astype(np.int64), np.asarray(ne) - 1)
    rho = np.zeros(nel)
    rho[np.ravel_multi_index(tuple(cell.T), tuple(int(n) for n in ne))] = values
    return rho


# ----------------------------------------------------------------------------------------------
# TensorProductSimulator<1,1,1>
# ----------------------------------------------------------------------------------------------

class TensorProductSimulator1_1_1:
    """``pyVoxelFEM.detail.TensorProductSimulator1_1_1`` (VoxelFEM.cc:48-92; TPS.hh:219-1419)."""

    N = 3

    def __init__(self, domainBoundingBox, numElemg):
        _lib.require_gpu()
        self._lib = _lib.load()
        lo = np.asarray(domainBoundingBox[0], dtype=np.float64).reshape(-1)
        hi = np.asarray(domainBoundingBox[1], dtype=np.float64).reshape(-1)
        ne = [int(v) for v in numElemg]
        if len(ne) != 3 or lo.size != 3 or hi.size != 3:
            raise RuntimeError("Dimension mismatch: %d vs 3" % len(ne))
        self._bbmin, self._bbmax = lo.copy(), hi.copy()
        self._ne = np.array(ne, dtype=np.int64)
        self._nn = self._ne + 1
        h = ctypes.c_void_p()
        _lib.check(self._lib.vfem_sim_create(
            ctypes.byref(h), lo.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            hi.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            self._ne.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        self._h = h
        self._E0, self._Emin, self._gamma = 1.0, 1e-9, 3.0           # TPS.hh:1392-1394
        self._mask = np.zeros((self.numNodes(), 3), dtype=bool)
        self._dvals = np.zeros((self.numNodes(), 3))
        self._loads = torch.zeros((self.numNodes(), 3), dtype=torch.float64, device=_dev())
        self._has_force = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.vfem_sim_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- sizes / geometry ----
    def numNodes(self):
        return int(np.prod(self._nn))

    def numElements(self):
        return int(np.prod(self._ne))

    def NbElementsPerDimension(self):
        return self._ne.copy()

    def nodePosition(self, ni):
        idx = np.array(np.unravel_index(int(ni), tuple(self._nn)), dtype=np.float64)
        return self._bbmin + idx * (self._bbmax - self._bbmin) / (self._nn - 1.0)       # TPS.hh:275-278

    def elementIndexForGridCell(self, cellIdxs):
        return int(np.ravel_multi_index(tuple(int(c) for c in cellIdxs), tuple(self._ne)))

    def elementNodes(self, ei):
        e = np.array(np.unravel_index(int(ei), tuple(self._ne)))
        out = []
        for m in range(8):
            loc = np.array([(m >> 2) & 1, (m >> 1) & 1, m & 1])
            out.append(int(np.ravel_multi_index(tuple(e + loc), tuple(self._nn))))
        return np.array(out, dtype=np.uint64)

    def elemNodeGlobalIndex(self, ei, n):
        return int(self.elementNodes(ei)[int(n)])

    def getMesh(self):
        """(V, F): node positions and 8-node hexahedra in Gmsh ordering (TPS.hh:531-565)."""
        idx = np.stack(np.meshgrid(*[np.arange(n) for n in self._nn], indexing="ij"), -1).reshape(-1, 3)
        V = self._bbmin + idx * (self._bbmax - self._bbmin) / (self._nn - 1.0)
        eidx = np.stack(np.meshgrid(*[np.arange(n) for n in self._ne], indexing="ij"), -1).reshape(-1, 3)
        nstr = np.array([self._nn[1] * self._nn[2], self._nn[2], 1])
        first = eidx @ nstr
        loc = [np.array([(m >> 2) & 1, (m >> 1) & 1, m & 1]) @ nstr for m in range(8)]
        order = [0, 1, 3, 2, 4, 5, 7, 6]
        F = np.stack([first + loc[m] for m in order], axis=1)
        return V, F

    # ---- material / SIMP ----
    def readMaterial(self, materialPath):
        young, poisson = _read_isotropic_material(materialPath)
        self._young, self._poisson = young, poisson
        _lib.check(self._lib.vfem_sim_set_isotropic(self._h, young, poisson))
        self._direct_mg = None                    # the reference resets its solver when the operator changes (TPS.hh:404)

    def _push_simp(self):
        _lib.check(self._lib.vfem_sim_set_simp(self._h, self._E0, self._Emin, self._gamma))

    def _lame(self):
        E, nu = getattr(self, "_young", 1.0), getattr(self, "_poisson", 0.0)         # ETensor(1, 0) default, TPS.hh:1379
        return nu * E / ((1.0 + nu) * (1.0 - 2.0 * nu)), E / (2.0 + 2.0 * nu)

    def readDensities(self, materialPath, fieldName="density"):
        """TPS::readDensities (VoxelFEM.cc:54)"""
        self.setElementDensities(_densities_from_msh(materialPath, fieldName, self._ne, 3))

    def constantStrainLoad(self, eps):
        """TPS::constantStrainLoad (VoxelFEM.cc:66), evaluated on the device"""
        lam, mu = self._lame()
        rho = self.getDensities_device()[:self.numElements()].reshape(tuple(int(n) for n in self._ne))
        return _to_np(_constant_strain_load(eps, lam, mu, (self._bbmax - self._bbmin) / self._ne, 1, rho))

    def getK(self):
        """TPS::getK (VoxelFEM.cc:62): assembled stiffness matrix, upper triangle, compressed columns (host, small grids)"""
        if self.numElements() > (1 << 21):
            raise RuntimeError("getK assembles on the host; use applyK for grids of this size")
        rho = self.getDensities()[:self.numElements()]
        young = self._Emin + rho ** self._gamma * (self._E0 - self._Emin)
        nodes = np.stack([self.elementNodes(0) - 0], 0).astype(np.int64)
        eidx = np.stack(np.meshgrid(*[np.arange(n) for n in self._ne], indexing="ij"), -1).reshape(-1, 3)
        nstr = np.array([self._nn[1] * self._nn[2], self._nn[2], 1])
        nodes = (eidx @ nstr)[:, None] + nodes
        return _assemble_upper(nodes, self.fullDensityElementStiffnessMatrix(), young, 3, self.numNodes())

    E_0 = property(lambda s: s._E0, lambda s, v: (setattr(s, "_E0", float(v)), s._push_simp())[0])
    E_min = property(lambda s: s._Emin, lambda s, v: (setattr(s, "_Emin", float(v)), s._push_simp())[0])
    gamma = property(lambda s: s._gamma, lambda s, v: (setattr(s, "_gamma", float(v)), s._push_simp())[0])

    @property
    def ETensor(self):
        raise RuntimeError("ETensor objects are not exposed; use readMaterial (isotropic materials)")

    def fullDensityElementStiffnessMatrix(self):
        K0 = np.empty((24, 24))
        _lib.check(self._lib.vfem_sim_k0(self._h, K0.ctypes.data_as(ctypes.c_void_p)))
        return K0

    def elementStiffnessMatrix(self, ei):
        rho = self.elementDensity(ei)
        return (self._Emin + rho ** self._gamma * (self._E0 - self._Emin)) * self.fullDensityElementStiffnessMatrix()

    def clearCachedElementStiffness(self):
        pass

    # ---- densities ----
    def setUniformDensities(self, density):
        _lib.check(self._lib.vfem_sim_set_uniform_density(self._h, float(density), _stream()))

    def setElementDensities(self, rho):
        if int(self._lib.vfem_sim_num_stored_elements(self._h)) != self.numElements():
            raise RuntimeError("padded slab simulator: use setElementDensities_padded")
        t = _to_dev(rho, (self.numElements(),))
        _lib.check(self._lib.vfem_sim_set_densities(self._h, _ptr(t), _stream()))

    def setElementDensities_padded(self, rho):
        """slab simulators: densities for every stored element layer (node grid + padding)"""
        n = int(self._lib.vfem_sim_num_stored_elements(self._h))
        t = _to_dev(rho, (n,))
        _lib.check(self._lib.vfem_sim_set_densities(self._h, _ptr(t), _stream()))

    def getDensities_device(self):
        t = torch.empty(int(self._lib.vfem_sim_num_stored_elements(self._h)), dtype=torch.float64, device=_dev())
        _lib.check(self._lib.vfem_sim_get_densities(self._h, _ptr(t), _stream()))
        return t

    def getDensities(self):
        return _to_np(self.getDensities_device())

    def setElementDensity(self, ei, value):
        t = self.getDensities_device()
        t[int(ei)] = float(value)
        _lib.check(self._lib.vfem_sim_set_densities(self._h, _ptr(t), _stream()))

    def elementDensity(self, ei):
        return float(self.getDensities_device()[int(ei)].item())

    # ---- boundary conditions ----
    def _push_dirichlet(self):
        m = (self._mask[:, 0].astype(np.uint8) | (self._mask[:, 1].astype(np.uint8) << 1) |
             (self._mask[:, 2].astype(np.uint8) << 2))
        m = np.ascontiguousarray(m)
        vals = np.ascontiguousarray(self._dvals)
        _lib.check(self._lib.vfem_sim_set_dirichlet(self._h, m.ctypes.data_as(ctypes.c_void_p),
                                                   vals.ctypes.data_as(ctypes.c_void_p)))
        self._direct_mg = None                    # coarse Dirichlet masks of a cached hierarchy would be stale (TPS.hh:404)

    def _push_loads(self):
        _lib.check(self._lib.vfem_sim_set_loads(self._h, _ptr(self._loads), _stream()))

    def applyDisplacementsAndLoadsFromFile(self, bcPath):
        """applyDisplacementsAndLoads (TPS.hh:358-409): inclusive box test on node coordinates; forces are
        split evenly over the matched nodes; Dirichlet components merge, conflicting values throw."""
        size = self._bbmax - self._bbmin
        spacing = size / (self._nn - 1.0)
        coords = [self._bbmin[d] + np.arange(self._nn[d]) * spacing[d] for d in range(3)]
        shape = tuple(self._nn)
        mask3 = self._mask.reshape(shape + (3,))
        vals3 = self._dvals.reshape(shape + (3,))
        loads = self._loads.reshape(shape + (3,))
        for kind, comps, value, lo, hi, relative in _parse_regions(bcPath):
            lo, hi = np.array(lo[:3]), np.array(hi[:3])
            if relative:
                lo, hi = self._bbmin + lo * size, self._bbmin + hi * size
            sel = [np.flatnonzero((coords[d] >= lo[d]) & (coords[d] <= hi[d])) for d in range(3)]
            count = int(np.prod([s.size for s in sel]))
            if kind == "force":
                if count == 0:
                    raise RuntimeError("Force constraint region unmatched")
                ix = torch.from_numpy(sel[0]).to(_dev())
                iy = torch.from_numpy(sel[1]).to(_dev())
                iz = torch.from_numpy(sel[2]).to(_dev())
                v = torch.tensor(value[:3], dtype=torch.float64, device=_dev()) / count
                loads[ix[:, None, None], iy[None, :, None], iz[None, None, :]] = v
            else:
                if count == 0:
                    raise RuntimeError("Dirichlet region unmatched")
                blk = np.ix_(sel[0], sel[1], sel[2])
                for c, name in enumerate("xyz"):
                    if name not in comps:
                        continue
                    already = mask3[..., c][blk]
                    if np.any(already & (np.abs(vals3[..., c][blk] - value[c]) > 1e-10)):
                        raise RuntimeError("Conflicting dirichlet displacements.")
                    vc = vals3[..., c]
                    mc = mask3[..., c]
                    vc[blk] = np.where(already, vc[blk], value[c])
                    mc[blk] = True
        self._push_dirichlet()
        self._push_loads()

    def _get_mask(self):
        return self._mask.copy()

    def _set_mask(self, mask):
        mask = np.asarray(mask, dtype=bool)
        if mask.shape != (self.numNodes(), 3):
            raise RuntimeError("Size mismatch")
        self._mask = mask.copy()
        self._push_dirichlet()

    def _get_dvals(self):
        return self._dvals.copy()

    def _set_dvals(self, values):
        values = np.asarray(values, dtype=np.float64)
        if values.shape != (self.numNodes(), 3):
            raise RuntimeError("Size mismatch")
        self._dvals = values.copy()
        self._push_dirichlet()

    dirichletMask = property(_get_mask, _set_mask)
    dirichletValues = property(_get_dvals, _set_dvals)

    def getDirichletVarsAndValues(self):
        idx = np.flatnonzero(self._mask.reshape(-1))
        return list(idx), list(self._dvals.reshape(-1)[idx])

    def getForceMask(self):
        return (_to_np(self._loads) != 0)

    def setLoads_device(self, f):
        self._loads = _to_dev(f, (self.numNodes(), 3)).clone()
        self._push_loads()

    def buildLoadVector_device(self):
        return self._loads.clone()

    def buildLoadVector(self):
        return _to_np(self._loads)

    # ---- operators ----
    def applyK_device(self, u, variant=0):
        u = _to_dev(u, (self.numNodes(), 3))
        out = torch.empty_like(u)
        _lib.check(self._lib.vfem_sim_apply_k(self._h, _ptr(u), _ptr(out), int(variant), _stream()))
        return out

    def applyK(self, u):
        return _to_np(self.applyK_device(u))

    def complianceGradient_device(self, u):
        u = _to_dev(u, (self.numNodes(), 3))
        g = torch.empty(self.numElements(), dtype=torch.float64, device=_dev())
        _lib.check(self._lib.vfem_sim_compliance_gradient(self._h, _ptr(u), _ptr(g), _stream()))
        return g

    def _compliance(self, f, u):
        v = ctypes.c_double(0.0)
        _lib.check(self._lib.vfem_compliance(self._h, _ptr(f), _ptr(u), ctypes.byref(v), _stream()))
        return v.value

    def multigridSolver(self, numCoarseningLevels):
        return MultigridSolver1_1_1(self, int(numCoarseningLevels))

    def _direct_levels(self):
        lv, ne = 0, self._ne.copy()
        while np.all(ne % 2 == 0) and np.prod(ne + 1) * 3 > 3000 and lv < 12:
            ne //= 2
            lv += 1
        return lv

    def solve(self, f):
        """TPS::solve (TPS.hh:834-865).  The reference factorises with CHOLMOD; here the same system is
        solved by multigrid-preconditioned CG driven to a relative residual of 1e-11."""
        if np.any(self._dvals[self._mask] != 0):
            raise RuntimeError("Nonzero Dirichlet constraints currently unsupported")
        mg = getattr(self, "_direct_mg", None)
        if mg is None:
            mg = self.multigridSolver(self._direct_levels())
            self._direct_mg = mg
        u = mg.preconditionedConjugateGradient_device(torch.zeros((self.numNodes(), 3), dtype=torch.float64,
                                                                  device=_dev()),
                                                      _to_dev(f, (self.numNodes(), 3)), 500, 1e-11, None, 1, 2, True)
        if not mg.last_relative_residual <= 1e-11:
            raise RuntimeError("TensorProductSimulator.solve: the iterative solve that stands in for the direct factorisation "
                               "did not converge (relative residual %.3e after %d iterations)"
                               % (mg.last_relative_residual, mg.last_iterations))
        return _to_np(u)

    def solveWithImposedLoads(self):
        return self.solve(self.buildLoadVector())


# ----------------------------------------------------------------------------------------------
# MultigridSolver<1,1,1>
# ----------------------------------------------------------------------------------------------

class _LevelView:
    """What ``MultigridSolver.getSimulator(l)`` exposes for coarse levels (sizes + Dirichlet mask)."""

    def __init__(self, mg, l):
        self._mg, self._l = mg, l

    def numNodes(self):
        return int(self._mg._lib.vfem_mg_level_num_nodes(self._mg._h, self._l))

    def NbElementsPerDimension(self):
        ne = (ctypes.c_int64 * 3)()
        _lib.check(self._mg._lib.vfem_mg_level_dims(self._mg._h, self._l, ne))
        return np.array(list(ne), dtype=np.int64)

    def numElements(self):
        return int(np.prod(self.NbElementsPerDimension()))

    @property
    def dirichletMask(self):
        m = np.empty(self.numNodes(), dtype=np.uint8)
        _lib.check(self._mg._lib.vfem_mg_level_dirichlet_mask(self._mg._h, self._l, m.ctypes.data_as(ctypes.c_void_p)))
        return np.stack([(m >> c) & 1 for c in range(3)], axis=1).astype(bool)


class MultigridSolver1_1_1:
    """``pyVoxelFEM.detail.MultigridSolver1_1_1`` (VoxelFEM.cc:94-131; MG.hh:11-759)."""

    def __init__(self, tps, numCoarseningLevels):
        self._lib = _lib.load()
        self._tps = tps                      # keeps the fine simulator alive (MG.hh:32,88)
        h = ctypes.c_void_p()
        _lib.check(self._lib.vfem_mg_create(ctypes.byref(h), tps._h, int(numCoarseningLevels)))
        self._h = h
        self.L = int(numCoarseningLevels)
        self.buildBlockStiffnessMatrices = True
        self.buildFinestBlockStiffnessMatrix = False
        self.last_iterations = 0
        self.last_relative_residual = 0.0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.vfem_mg_destroy(h)
            except Exception:
                pass
            self._h = None

    def _nn(self, l):
        n = int(self._lib.vfem_mg_level_num_nodes(self._h, int(l)))
        if n < 0:
            raise IndexError("vector::_M_range_check")      # m_sims.at(l)
        return n

    def getSimulator(self, l):
        self._nn(l)
        return self._tps if int(l) == 0 else _LevelView(self, int(l))

    def setSymmetricGaussSeidel(self, symmetric):
        self._lib.vfem_mg_set_symmetric_gauss_seidel(self._h, int(bool(symmetric)))

    def updateElementStiffnessMatrices(self):
        _lib.check(self._lib.vfem_mg_update_operators(self._h, _stream()))

    def updateBlockKs(self):
        _lib.check(self._lib.vfem_mg_update_operators(self._h, _stream()))

    # ---- per-operator entry points (device variants first) ----
    def applyK_device(self, l, u):
        u = _to_dev(u, (self._nn(l), 3))
        out = torch.empty_like(u)
        _lib.check(self._lib.vfem_mg_apply_k(self._h, int(l), _ptr(u), _ptr(out), _stream()))
        return out

    def applyK(self, l, u):
        return _to_np(self.applyK_device(l, u))

    def computeResidual_device(self, l, u, b):
        u = _to_dev(u, (self._nn(l), 3))
        b = _to_dev(b, (self._nn(l), 3))
        r = torch.empty_like(u)
        _lib.check(self._lib.vfem_mg_residual(self._h, int(l), _ptr(u), _ptr(b), _ptr(r), _stream()))
        return r

    def computeResidual(self, l, u, b):
        return _to_np(self.computeResidual_device(l, u, b))

    def smoothing_device(self, l, u, b, forward=True):
        u = _to_dev(u, (self._nn(l), 3)).clone()
        b = _to_dev(b, (self._nn(l), 3))
        if u.shape != b.shape:
            raise RuntimeError("Invalid input size")
        _lib.check(self._lib.vfem_mg_smooth(self._h, int(l), _ptr(u), _ptr(b), int(bool(forward)), _stream()))
        return u

    def smoothing(self, l, u, b):
        """Bound as updateElementStiffnessMatrices + one forward sweep on a copy (VoxelFEM.cc:99-104).  The
        reference binding runs the sequential sweep (MG.hh:269-282); the solver itself uses the multicoloured
        sweep (MG.hh:336-340), which is what this returns."""
        self.updateElementStiffnessMatrices()
        return _to_np(self.smoothing_device(l, u, b, True))

    def zeroOutDirichletComponents(self, l, u):
        t = _to_dev(u, (self._nn(l), 3)).clone()
        _lib.check(self._lib.vfem_mg_zero_dirichlet(self._h, int(l), _ptr(t), _stream()))
        return _to_np(t)

    def restriction_device(self, fine_level, values):
        v = _to_dev(values, (self._nn(fine_level), 3))
        out = torch.empty((self._nn(fine_level + 1), 3), dtype=torch.float64, device=_dev())
        _lib.check(self._lib.vfem_mg_restrict(self._h, int(fine_level), _ptr(v), _ptr(out), _stream()))
        return out

    def interpolation_device(self, fine_level, values, out=None):
        v = _to_dev(values, (self._nn(fine_level + 1), 3))
        acc = out is not None
        if out is None:
            out = torch.empty((self._nn(fine_level), 3), dtype=torch.float64, device=_dev())
        _lib.check(self._lib.vfem_mg_interpolate(self._h, int(fine_level), _ptr(v), _ptr(out), int(acc), _stream()))
        return out

    def coarsestSolve_device(self, b):
        b = _to_dev(b, (self._nn(self.L), 3))
        x = torch.empty_like(b)
        _lib.check(self._lib.vfem_mg_coarsest_solve(self._h, _ptr(b), _ptr(x), _stream()))
        return x

    def _field(self, which, l):
        n = self._nn(l) if which != 2 else self._nn(0)
        p = self._lib.vfem_mg_field_ptr(self._h, which, int(l))
        out = torch.empty((n, 3), dtype=torch.float64, device=_dev())
        _lib.check(self._lib.vfem_copy_d2d(_ptr(out), ctypes.c_void_p(p), n * 3 * 8, _stream()))
        return out

    def debug_get_x(self, l):
        return _to_np(self._field(0, l))

    def debug_get_b(self, l):
        return _to_np(self._field(1, l))

    def debugMulticolorVisit(self):
        """Visit order of the multicoloured sweep on level 0 (MG.hh:328-334)."""
        nn = tuple(self._tps._nn)
        result = np.zeros(nn, dtype=np.int32)
        counter = 0
        for lni in range(8):
            c = ((lni >> 2) & 1, (lni >> 1) & 1, lni & 1)
            sub = result[c[0]::2, c[1]::2, c[2]::2]
            sub[...] = counter + np.arange(sub.size).reshape(sub.shape)
            counter += sub.size
        return result.reshape(-1)

    # ---- solvers ----
    def solve_device(self, u, f, numSteps, numSmoothingSteps, stiffnessUpdated=False, zeroDirichlet=False,
                     it_callback=None, fullMultigrid=False):
        x = _to_dev(u, (self._nn(0), 3)).clone()
        f = _to_dev(f, (self._nn(0), 3))
        if it_callback is None:
            _lib.check(self._lib.vfem_mg_solve(self._h, _ptr(x), _ptr(f), int(numSteps), int(numSmoothingSteps),
                                               int(bool(stiffnessUpdated)), int(bool(zeroDirichlet)),
                                               int(bool(fullMultigrid)), _stream()))
            return x
        for i in range(int(numSteps)):
            _lib.check(self._lib.vfem_mg_solve(self._h, _ptr(x), _ptr(f), 1, int(numSmoothingSteps),
                                               int(bool(stiffnessUpdated) or i > 0), int(bool(zeroDirichlet)),
                                               int(bool(fullMultigrid) and i == 0), _stream()))
            it_callback(i, _to_np(x))
        return x

    def solve(self, u, f, numSteps, numSmoothingSteps, stiffnessUpdated=False, zeroDirichlet=False,
              it_callback=None, fullMultigrid=False):
        return _to_np(self.solve_device(u, f, numSteps, numSmoothingSteps, stiffnessUpdated, zeroDirichlet,
                                        it_callback, fullMultigrid))

    def preconditionedConjugateGradient_device(self, u, b, maxIter, tol, it_callback=None, mgIterations=1,
                                               mgSmoothingIterations=1, fullMultigrid=False, residual_cb=None):
        if isinstance(u, torch.Tensor) and isinstance(b, torch.Tensor) and u.shape != b.shape:
            raise RuntimeError("x and b should have the same size")
        x = _to_dev(u).reshape(-1, 3).clone()
        b = _to_dev(b).reshape(-1, 3)
        if x.shape[0] != b.shape[0]:
            raise RuntimeError("x and b should have the same size")
        if x.shape[0] != self._nn(0):
            raise RuntimeError("size of input and number of nodes don't correspond")
        its = ctypes.c_int(0)
        rel = ctypes.c_double(0.0)

        def _cb(_user, it, rnorm):
            if residual_cb is not None:
                residual_cb(it, rnorm)
            if it_callback is not None:
                it_callback(it, _to_np(x), _to_np(self._field(2, 0)))

        cb = _lib.RESIDUAL_CB(_cb) if (residual_cb is not None or it_callback is not None) else _lib.RESIDUAL_CB()
        _lib.check(self._lib.vfem_mg_pcg(self._h, _ptr(x), _ptr(b), int(maxIter), float(tol), int(mgIterations),
                                         int(mgSmoothingIterations), int(bool(fullMultigrid)), cb, None,
                                         ctypes.byref(its), ctypes.byref(rel), _stream()))
        self.last_iterations = its.value
        self.last_relative_residual = rel.value
        return x

    def preconditionedConjugateGradient(self, u, b, maxIter, tol, it_callback=None, mgIterations=1,
                                        mgSmoothingIterations=1, fullMultigrid=False):
        return _to_np(self.preconditionedConjugateGradient_device(u, b, maxIter, tol, it_callback, mgIterations,
                                                                  mgSmoothingIterations, fullMultigrid))


# ----------------------------------------------------------------------------------------------
# objectives (TopologyOptimizationObjective.hh)
# ----------------------------------------------------------------------------------------------

class ComplianceObjective1_1_1:
    """ComplianceObjective (TopologyOptimizationObjective.hh:24-63): 1/2 f.u with a direct solve."""

    def __init__(self, simulator, skipSolve=False):
        self._sim = simulator
        self._f = simulator.buildLoadVector_device()
        self._u = torch.zeros_like(self._f)
        if not skipSolve:
            self.updateCache(None)

    def compliance(self):
        return self._sim._compliance(self._f, self._u)

    def evaluate(self, xPhys=None):
        return self.compliance()

    def gradient_device(self):
        return self._sim.complianceGradient_device(self._u)

    def gradient(self):
        return _to_np(self.gradient_device())

    def updateCache(self, xPhys):
        if xPhys is not None:
            self._sim.setElementDensities(xPhys)
        self._u = _to_dev(self._sim.solve(self._f), self._f.shape)

    def u(self):
        return _to_np(self._u)

    def f(self):
        return _to_np(self._f)


class MultigridComplianceObjective1_1_1(ComplianceObjective1_1_1):
    """MultigridComplianceObjective (TopologyOptimizationObjective.hh:67-102)."""

    def __init__(self, mg_solver):
        self._mg = mg_solver
        self.cgIter = 100
        self.tol = 1e-5
        self.mgIterations = 1
        self.mgSmoothingIterations = 2
        self.fullMultigrid = True
        self.zeroInit = False
        self.residual_cb = None
        super().__init__(mg_solver.getSimulator(0), skipSolve=True)
        self.updateCache(None)          # the reference constructor solves once (:78-82)

    @property
    def mg(self):
        return self._mg

    def updateCache(self, xPhys):
        if xPhys is not None:
            self._sim.setElementDensities(xPhys)
        if self.zeroInit:
            self._u = torch.zeros_like(self._u)
        self._u = self._mg.preconditionedConjugateGradient_device(
            self._u, self._f, int(self.cgIter), float(self.tol), None, int(self.mgIterations),
            int(self.mgSmoothingIterations), bool(self.fullMultigrid), residual_cb=self.residual_cb)


# ----------------------------------------------------------------------------------------------
# filters / constraint (TopologyOptimizationFilter.hh, TopologyOptimizationConstraint.hh) -- device kernels
# (vfem_box_filter, vfem_projection*, vfem_mean); the public methods take/return numpy like the bound C++ classes,
# the *_dev variants work on float64 CUDA tensors.
# ----------------------------------------------------------------------------------------------

class _Filter:
    def __init__(self):
        self._grid = None

    def _set_grid(self, dims):
        self._grid = tuple(int(d) for d in dims)

    def _check(self):
        if self._grid is None:
            raise RuntimeError("Filter grid dimensions not set. Initialize a TopologyOpimizationProblem object "
                               "with this filter before using it.")

    def getGridDimensions(self):
        return np.array(self._grid)

    def _n3(self):
        g = list(self._grid) + [1] * (3 - len(self._grid))
        return (ctypes.c_int64 * 3)(*g)

    def apply(self, x):
        return _to_np(self.apply_dev(_to_dev(x).reshape(-1)))

    def backprop(self, g, x):
        return _to_np(self.backprop_dev(_to_dev(g).reshape(-1), _to_dev(x).reshape(-1)))


class ProjectionFilter(_Filter):
    """tanh projection (TopologyOptimizationFilter.hh:55-79)."""

    def __init__(self):
        super().__init__()
        self._beta = 1.0

    def _get_beta(self):
        return self._beta

    def _set_beta(self, beta):
        if beta <= 0:
            raise RuntimeError("Beta parameter has to be positive (received beta = %f)" % beta)
        self._beta = float(beta)

    beta = property(_get_beta, _set_beta)

    def apply_dev(self, x):
        out = torch.empty_like(x)
        _lib.check(_lib.load().vfem_projection(x.numel(), self._beta, _ptr(x), _ptr(out), _stream()))
        return out

    def backprop_dev(self, g, x):
        out = torch.empty_like(g)
        _lib.check(_lib.load().vfem_projection_backprop(g.numel(), self._beta, _ptr(g), _ptr(x), _ptr(out), _stream()))
        return out


class SmoothingFilter(_Filter):
    """Box filter, each row normalised by its in-bounds neighbour count (TopologyOptimizationFilter.hh:105-162);
    the sparse matrix of the reference is replaced by a direct stencil kernel."""

    def __init__(self):
        super().__init__()
        self._radius = 1

    def _get_radius(self):
        return self._radius

    def _set_radius(self, r):
        self._radius = int(r)

    radius = property(_get_radius, _set_radius)

    def _run(self, x, transpose):
        self._check()
        out = torch.empty_like(x)
        _lib.check(_lib.load().vfem_box_filter(self._n3(), self._radius, _ptr(x), _ptr(out), int(transpose), _stream()))
        return out

    def apply_dev(self, x):
        return self._run(x, 0)

    def backprop_dev(self, g, x):
        return self._run(g, 1)


class PythonFilter(_Filter):
    """Callback filter (TopologyOptimizationFilter.hh:81-103): apply_cb(in, out), backprop_cb(in, vars, out) on numpy."""

    def __init__(self):
        super().__init__()
        self.apply_cb = None
        self.backprop_cb = None

    def apply_dev(self, x):
        if self.apply_cb is None:
            raise RuntimeError("Apply callback must be configured")
        xin = _to_np(x)
        out = np.zeros_like(xin)
        self.apply_cb(xin, out)
        return _to_dev(out)

    def backprop_dev(self, g, x):
        if self.backprop_cb is None:
            raise RuntimeError("Backprop callback must be configured")
        gin = _to_np(g)
        out = np.zeros_like(gin)
        self.backprop_cb(gin, _to_np(x), out)
        return _to_dev(out)


class LangelaarFilter(_Filter):
    def apply_dev(self, x):
        raise RuntimeError("LangelaarFilter is not part of the accelerated path (never used by the drivers)")

    backprop_dev = apply_dev


def applyFilter(filter, x):
    filter._check()
    return filter.apply(np.asarray(x, dtype=np.float64))


class TotalVolumeConstraint:
    """1 - mean(x)/v (TopologyOptimizationConstraint.hh:21-34)."""

    def __init__(self, volumeFraction):
        self.volumeFraction = float(volumeFraction)

    def evaluate_dev(self, x):
        m = ctypes.c_double(0.0)
        _lib.check(_lib.load().vfem_mean(x.numel(), _ptr(x), ctypes.byref(m), _stream()))
        return 1.0 - m.value / self.volumeFraction

    def evaluate(self, x):
        return self.evaluate_dev(_to_dev(x).reshape(-1))

    def backprop_dev(self, x):
        return torch.full((x.numel(),), -1.0 / (self.volumeFraction * x.numel()), dtype=torch.float64, device=_dev())

    def backprop(self, x):
        return np.full(np.size(x), -1.0 / (self.volumeFraction * np.size(x)))


# ----------------------------------------------------------------------------------------------
# problem + optimality criterion (TopologyOptimizationProblem.hh, OptimalityCriterion.hh); the design variables,
# every filtered stage and the sensitivities stay on the device between calls
# ----------------------------------------------------------------------------------------------

class TopologyOptimizationProblem1_1_1:
    def __init__(self, simulator, objective, constraints, filters=()):
        self._sim = simulator
        self._objective = objective
        self._constraints = list(constraints)
        self._filters = list(filters)
        self._nvars = simulator.numElements()
        for f in self._filters:
            f._set_grid(simulator.NbElementsPerDimension())
        z = lambda: torch.zeros(self._nvars, dtype=torch.float64, device=_dev())
        self._cached = [z() for _ in range(len(self._filters) + 1)]
        self._vars_set = False

    def numVars(self):
        return self._nvars

    def getVars(self):
        return _to_np(self._cached[0])

    def getVars_device(self):
        return self._cached[0]

    def setVars(self, x, forceUpdate=False):
        x = _to_dev(x, (self._nvars,))
        if (not forceUpdate) and self._vars_set and float((x - self._cached[0]).norm()) < 1e-16:
            return False                                             # Problem.hh:50-51
        self._cached[0] = x.clone()
        for i, f in enumerate(self._filters):
            self._cached[i + 1] = f.apply_dev(self._cached[i])
        self._objective.updateCache(self._cached[-1])
        self._vars_set = True
        return True

    def _need_vars(self):
        if not self._vars_set:
            raise RuntimeError("Must call setVars first!")

    def evaluateOCConstraintAtVars_dev(self, x):
        if len(self._constraints) != 1 or not isinstance(self._constraints[0], TotalVolumeConstraint):
            raise RuntimeError("Applicable only for a topology optimization with a single (volume) constraint")
        for f in self._filters:
            x = f.apply_dev(x)
        return self._constraints[0].evaluate_dev(x)

    def evaluateOCConstraintAtVars(self, x):
        return self.evaluateOCConstraintAtVars_dev(_to_dev(x, (self._nvars,)))

    def evaluateObjective(self):
        self._need_vars()
        return self._objective.evaluate(None)

    def evaluateObjectiveGradient_device(self):
        self._need_vars()
        g = self._objective.gradient_device()
        nf = len(self._filters)
        for i in range(nf):
            g = self._filters[nf - 1 - i].backprop_dev(g, self._cached[nf - 1 - i])
        return g

    def evaluateObjectiveGradient(self):
        return _to_np(self.evaluateObjectiveGradient_device())

    def evaluateConstraints(self):
        self._need_vars()
        return np.array([c.evaluate_dev(self._cached[-1]) for c in self._constraints])

    def evaluateConstraintsJacobian_device(self):
        self._need_vars()
        nf = len(self._filters)
        rows = []
        for c in self._constraints:
            d = c.backprop_dev(self._cached[-1])
            for i in range(nf):
                d = self._filters[nf - 1 - i].backprop_dev(d, self._cached[nf - 1 - i])
            rows.append(d)
        return rows

    def evaluateConstraintsJacobian(self):
        return np.stack([_to_np(r) for r in self.evaluateConstraintsJacobian_device()]).reshape(len(self._constraints), self._nvars)

    def getDensities(self):
        return self._sim.getDensities()

    def getSimulator(self):
        return self._sim

    objective = property(lambda s: s._objective, lambda s, o: setattr(s, "_objective", o))
    constraints = property(lambda s: list(s._constraints), lambda s, c: setattr(s, "_constraints", list(c)))

    def _set_filters(self, filters):
        self._filters = list(filters)
        for f in self._filters:
            f._set_grid(self._sim.NbElementsPerDimension())
        self._cached = [self._cached[0]] + [torch.zeros(self._nvars, dtype=torch.float64, device=_dev()) for _ in self._filters]

    filters = property(lambda s: list(s._filters), _set_filters)


class OCOptimizer1_1_1:
    """OCOptimizer (OptimalityCriterion.hh:30-81); the multiplier bracket persists across steps.  Each bisection
    probe is three small kernels (candidate step, filters, mean) and one scalar read-back."""

    def __init__(self, problem):
        self._p = problem
        self._lmin, self._lmax = 1.0, 2.0

    def step(self, m=0.2, ctol=1e-6):
        p = self._p
        lib = _lib.load()
        dJ = p.evaluateObjectiveGradient_device()
        dc = p.evaluateConstraintsJacobian_device()[0]
        x0 = p.getVars_device().clone()
        cand = torch.empty_like(x0)

        def stepped(lam):
            _lib.check(lib.vfem_oc_candidate(x0.numel(), _ptr(x0), _ptr(dJ), _ptr(dc), float(lam), float(m), _ptr(cand), _stream()))
            return cand

        def ceval(lam):
            return p.evaluateOCConstraintAtVars_dev(stepped(lam))

        while ceval(self._lmin) > 0:
            self._lmax = self._lmin
            self._lmin /= 2
        while ceval(self._lmax) < 0:
            self._lmin = self._lmax
            self._lmax *= 2
        mid = 0.5 * (self._lmin + self._lmax)
        vol = ceval(mid)
        while abs(vol) > ctol:
            if vol < 0:
                self._lmin = mid
            if vol > 0:
                self._lmax = mid
            mid = 0.5 * (self._lmin + self._lmax)
            vol = ceval(mid)
        p.setVars(stepped(mid).clone())
        print("objective, constraint, lambda estimate: %g\t%g\t%g" % (p.evaluateObjective(),
                                                                      p.evaluateConstraints()[0], mid))


# ----------------------------------------------------------------------------------------------
# module-level factories (VoxelFEM.cc:136-216, 234-240)
# ----------------------------------------------------------------------------------------------

class _GenericSimulator:
    """Generic path (``libvfem`` ``vfem_gsim_*``): TensorProductSimulator<p,..,p> for N = 2, 3 and p = 1, 2 other than
    the tuned <1,1,1> instantiation -- the 2-D simulators the reference binds (VoxelFEM.cc:226, plane stress) and
    the degree-2 elements its templates support (TPS.hh:97-110).  Same surface as ``TensorProductSimulator1_1_1``."""

    N = 3
    P = 1

    def __init__(self, domainBoundingBox, numElemg, _element_padding=(0, 0)):
        """``_element_padding`` (slab decomposition, ndr_amd/distributed_q2.py): element layers kept below / above the node
        grid along x in the density array only (``setElementDensities_padded``)"""
        _lib.require_gpu()
        self._lib = _lib.load()
        N, p = self.N, self.P
        lo = np.asarray(domainBoundingBox[0], dtype=np.float64).reshape(-1)
        hi = np.asarray(domainBoundingBox[1], dtype=np.float64).reshape(-1)
        ne = [int(v) for v in numElemg]
        if len(ne) != N or lo.size != N or hi.size != N:
            raise RuntimeError("Dimension mismatch: %d vs %d" % (len(ne), N))
        self._bbmin, self._bbmax = lo.copy(), hi.copy()
        self._ne = np.array(ne, dtype=np.int64)
        self._nn = p * self._ne + 1
        h = ctypes.c_void_p()
        self._pad = (int(_element_padding[0]), int(_element_padding[1]))
        _lib.check(self._lib.vfem_gsim_create_padded(
            ctypes.byref(h), N, p, lo.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            hi.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), self._ne.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
            self._pad[0], self._pad[1]))
        self._h = h
        self._E0, self._Emin, self._gamma = 1.0, 1e-9, 3.0           # TPS.hh:1392-1394
        self._mask = np.zeros((self.numNodes(), N), dtype=bool)
        self._dvals = np.zeros((self.numNodes(), N))
        self._loads = torch.zeros((self.numNodes(), N), dtype=torch.float64, device=_dev())

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.vfem_gsim_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- sizes / geometry ----
    def numNodes(self):
        return int(np.prod(self._nn))

    def numElements(self):
        return int(np.prod(self._ne))

    def NbElementsPerDimension(self):
        return self._ne.copy()

    def nodePosition(self, ni):
        idx = np.array(np.unravel_index(int(ni), tuple(self._nn)), dtype=np.float64)
        return self._bbmin + idx * (self._bbmax - self._bbmin) / (self._nn - 1.0)

    def elementIndexForGridCell(self, cellIdxs):
        return int(np.ravel_multi_index(tuple(int(c) for c in cellIdxs), tuple(self._ne)))

    def elementNodes(self, ei):
        e = np.array(np.unravel_index(int(ei), tuple(self._ne)))
        out = []
        for loc in np.ndindex(*([self.P + 1] * self.N)):
            out.append(int(np.ravel_multi_index(tuple(self.P * e + np.array(loc)), tuple(self._nn))))
        return np.array(out, dtype=np.uint64)

    def elemNodeGlobalIndex(self, ei, n):
        return int(self.elementNodes(ei)[int(n)])

    def getMesh(self):
        """(V, F): all node positions; per element its corner nodes in Gmsh quad / hexahedron order (TPS.hh:531-565)"""
        N, p = self.N, self.P
        idx = np.stack(np.meshgrid(*[np.arange(n) for n in self._nn], indexing="ij"), -1).reshape(-1, N)
        V = self._bbmin + idx * (self._bbmax - self._bbmin) / (self._nn - 1.0)
        eidx = np.stack(np.meshgrid(*[np.arange(n) for n in self._ne], indexing="ij"), -1).reshape(-1, N)
        nstr = np.array([int(np.prod(self._nn[d + 1:])) for d in range(N)])
        first = (p * eidx) @ nstr
        corners = [(0, 0), (1, 0), (1, 1), (0, 1)] if N == 2 else \
            [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]
        F = np.stack([first + (p * np.array(c)) @ nstr for c in corners], axis=1)
        return V, F

    # ---- material / SIMP ----
    def readMaterial(self, materialPath):
        young, poisson = _read_isotropic_material(materialPath)
        self._young, self._poisson = young, poisson
        _lib.check(self._lib.vfem_gsim_set_isotropic(self._h, young, poisson))
        self._direct_mg = None                    # the reference resets its solver when the operator changes (TPS.hh:404)

    def _lame(self):
        """ElasticityTensor::setIsotropic (ElasticityTensor.hh:100-133): 3-D Lame parameters, plane stress in 2-D"""
        E, nu = getattr(self, "_young", 1.0), getattr(self, "_poisson", 0.0)
        lam = nu * E / ((1.0 + nu) * (1.0 - 2.0 * nu)) if self.N == 3 else nu * E / (1.0 - nu * nu)
        return lam, E / (2.0 + 2.0 * nu)

    def readDensities(self, materialPath, fieldName="density"):
        """TPS::readDensities (VoxelFEM.cc:54)"""
        self.setElementDensities(_densities_from_msh(materialPath, fieldName, self._ne, self.N))

    def constantStrainLoad(self, eps):
        """TPS::constantStrainLoad (VoxelFEM.cc:66), evaluated on the device"""
        lam, mu = self._lame()
        rho = self.getDensities_device().reshape(tuple(int(n) for n in self._ne))
        return _to_np(_constant_strain_load(eps, lam, mu, (self._bbmax - self._bbmin) / self._ne, self.P, rho))

    def getK(self):
        """TPS::getK (VoxelFEM.cc:62): assembled stiffness matrix, upper triangle, compressed columns (host, small grids)"""
        if self.numElements() * (self.P + 1) ** (2 * self.N) > (1 << 30):
            raise RuntimeError("getK assembles on the host; use applyK for grids of this size")
        rho = self.getDensities()
        young = self._Emin + rho ** self._gamma * (self._E0 - self._Emin)
        nodes = np.stack([np.asarray(self.elementNodes(e), dtype=np.int64) for e in range(self.numElements())])
        return _assemble_upper(nodes, self.fullDensityElementStiffnessMatrix(), young, self.N, self.numNodes())

    def _push_simp(self):
        _lib.check(self._lib.vfem_gsim_set_simp(self._h, self._E0, self._Emin, self._gamma))

    E_0 = property(lambda s: s._E0, lambda s, v: (setattr(s, "_E0", float(v)), s._push_simp())[0])
    E_min = property(lambda s: s._Emin, lambda s, v: (setattr(s, "_Emin", float(v)), s._push_simp())[0])
    gamma = property(lambda s: s._gamma, lambda s, v: (setattr(s, "_gamma", float(v)), s._push_simp())[0])

    def fullDensityElementStiffnessMatrix(self):
        ke = int(self._lib.vfem_gsim_ke_size(self._h))
        K0 = np.empty((ke, ke))
        _lib.check(self._lib.vfem_gsim_k0(self._h, K0.ctypes.data_as(ctypes.c_void_p)))
        return K0

    def elementStiffnessMatrix(self, ei):
        rho = self.elementDensity(ei)
        return (self._Emin + rho ** self._gamma * (self._E0 - self._Emin)) * self.fullDensityElementStiffnessMatrix()

    def clearCachedElementStiffness(self):
        pass

    # ---- densities ----
    def setElementDensities(self, rho):
        if self._pad != (0, 0):
            raise RuntimeError("this simulator stores padding element layers: use setElementDensities_padded")
        t = _to_dev(rho, (self.numElements(),))
        _lib.check(self._lib.vfem_gsim_set_densities(self._h, _ptr(t), _stream()))

    def setElementDensities_padded(self, rho):
        """all stored element layers (padding below, the node grid's elements, padding above), x slowest"""
        t = _to_dev(rho, (int(self._lib.vfem_gsim_num_stored_elements(self._h)),))
        _lib.check(self._lib.vfem_gsim_set_densities(self._h, _ptr(t), _stream()))

    def setUniformDensities(self, density):
        if density > 1.0 or density < 0:
            raise RuntimeError("Density value (%f) has to be in between 0 and 1" % density)
        n = int(self._lib.vfem_gsim_num_stored_elements(self._h))
        self.setElementDensities_padded(torch.full((n,), float(density), dtype=torch.float64, device=_dev()))

    def getDensities_device(self):
        t = torch.empty(self.numElements(), dtype=torch.float64, device=_dev())
        _lib.check(self._lib.vfem_gsim_get_densities(self._h, _ptr(t), _stream()))
        return t

    def getDensities(self):
        return _to_np(self.getDensities_device())

    def setElementDensity(self, ei, value):
        t = self.getDensities_device()
        t[int(ei)] = float(value)
        self.setElementDensities(t)

    def elementDensity(self, ei):
        return float(self.getDensities_device()[int(ei)].item())

    # ---- boundary conditions ----
    def _push_dirichlet(self):
        m = np.zeros(self.numNodes(), dtype=np.uint8)
        for c in range(self.N):
            m |= self._mask[:, c].astype(np.uint8) << c
        vals = np.ascontiguousarray(self._dvals)
        _lib.check(self._lib.vfem_gsim_set_dirichlet(self._h, m.ctypes.data_as(ctypes.c_void_p),
                                                    vals.ctypes.data_as(ctypes.c_void_p)))
        self._direct_mg = None                    # coarse Dirichlet masks of a cached hierarchy would be stale (TPS.hh:404)

    def applyDisplacementsAndLoadsFromFile(self, bcPath):
        """applyDisplacementsAndLoads (TPS.hh:358-409), see ``TensorProductSimulator1_1_1``."""
        N = self.N
        size = self._bbmax - self._bbmin
        spacing = size / (self._nn - 1.0)
        coords = [self._bbmin[d] + np.arange(self._nn[d]) * spacing[d] for d in range(N)]
        shape = tuple(self._nn)
        mask3 = self._mask.reshape(shape + (N,))
        vals3 = self._dvals.reshape(shape + (N,))
        loads = np.zeros(shape + (N,))
        loads[...] = _to_np(self._loads).reshape(shape + (N,))
        for kind, comps, value, lo, hi, relative in _parse_regions(bcPath):
            lo, hi = np.array(lo[:N]), np.array(hi[:N])
            if relative:
                lo, hi = self._bbmin + lo * size, self._bbmin + hi * size
            sel = [np.flatnonzero((coords[d] >= lo[d]) & (coords[d] <= hi[d])) for d in range(N)]
            count = int(np.prod([s.size for s in sel]))
            blk = np.ix_(*sel)
            if kind == "force":
                if count == 0:
                    raise RuntimeError("Force constraint region unmatched")
                for c in range(N):
                    loads[..., c][blk] = value[c] / count
            else:
                if count == 0:
                    raise RuntimeError("Dirichlet region unmatched")
                for c, name in enumerate("xyz"[:N]):
                    if name not in comps:
                        continue
                    already = mask3[..., c][blk]
                    if np.any(already & (np.abs(vals3[..., c][blk] - value[c]) > 1e-10)):
                        raise RuntimeError("Conflicting dirichlet displacements.")
                    vc = vals3[..., c]
                    mc = mask3[..., c]
                    vc[blk] = np.where(already, vc[blk], value[c])
                    mc[blk] = True
        self._loads = _to_dev(loads.reshape(-1, N))
        self._push_dirichlet()

    def _get_mask(self):
        return self._mask.copy()

    def _set_mask(self, mask):
        mask = np.asarray(mask, dtype=bool)
        if mask.shape != (self.numNodes(), self.N):
            raise RuntimeError("Size mismatch")
        self._mask = mask.copy()
        self._push_dirichlet()

    def _get_dvals(self):
        return self._dvals.copy()

    def _set_dvals(self, values):
        values = np.asarray(values, dtype=np.float64)
        if values.shape != (self.numNodes(), self.N):
            raise RuntimeError("Size mismatch")
        self._dvals = values.copy()
        self._push_dirichlet()

    dirichletMask = property(_get_mask, _set_mask)
    dirichletValues = property(_get_dvals, _set_dvals)

    def getDirichletVarsAndValues(self):
        idx = np.flatnonzero(self._mask.reshape(-1))
        return list(idx), list(self._dvals.reshape(-1)[idx])

    def getForceMask(self):
        return (_to_np(self._loads) != 0)

    def setLoads_device(self, f):
        self._loads = _to_dev(f, (self.numNodes(), self.N)).clone()

    def buildLoadVector_device(self):
        return self._loads.clone()

    def buildLoadVector(self):
        return _to_np(self._loads)

    # ---- operators ----
    def applyK_device(self, u):
        u = _to_dev(u, (self.numNodes(), self.N))
        out = torch.empty_like(u)
        _lib.check(self._lib.vfem_gsim_apply_k(self._h, _ptr(u), _ptr(out), _stream()))
        return out

    def applyK(self, u):
        return _to_np(self.applyK_device(u))

    def complianceGradient_device(self, u):
        u = _to_dev(u, (self.numNodes(), self.N))
        g = torch.empty(self.numElements(), dtype=torch.float64, device=_dev())
        _lib.check(self._lib.vfem_gsim_compliance_gradient(self._h, _ptr(u), _ptr(g), _stream()))
        return g

    def _compliance(self, f, u):
        v = ctypes.c_double(0.0)
        _lib.check(self._lib.vfem_gsim_compliance(self._h, _ptr(f), _ptr(u), ctypes.byref(v), _stream()))
        return v.value

    def multigridSolver(self, numCoarseningLevels):
        return _GenericMultigridSolver(self, int(numCoarseningLevels))

    def _direct_levels(self):
        lv, ne = 0, self._ne.copy()
        while np.all(ne % 2 == 0) and np.prod(self.P * ne + 1) * self.N > 3000 and lv < 12:
            ne //= 2
            lv += 1
        return lv

    def solve(self, f):
        """TPS::solve (TPS.hh:834-865).  The reference factorises with CHOLMOD; here the same system is solved by
        multigrid-preconditioned CG driven to a relative residual of 1e-11 (the coarsest level is a dense Cholesky)."""
        if np.any(self._dvals[self._mask] != 0):
            raise RuntimeError("Nonzero Dirichlet constraints currently unsupported")
        mg = getattr(self, "_direct_mg", None)
        if mg is None:
            mg = self.multigridSolver(self._direct_levels())
            self._direct_mg = mg
        # a grid with odd element counts cannot be coarsened; beyond the dense-factorisation size it is solved by plain CG
        plain = mg.L == 0 and self.numNodes() * self.N > 40000
        u = mg.preconditionedConjugateGradient_device(
            torch.zeros((self.numNodes(), self.N), dtype=torch.float64, device=_dev()),
            _to_dev(f, (self.numNodes(), self.N)), 500000 if plain else 2000, 1e-11, None, 1, 0 if plain else 2, True)
        if mg.last_relative_residual > 1e-10:
            raise RuntimeError("direct-solve replacement did not converge (relative residual %g)" % mg.last_relative_residual)
        return _to_np(u)

    def solveWithImposedLoads(self):
        return self.solve(self.buildLoadVector())


class TensorProductSimulator1_1(_GenericSimulator):
    """``pyVoxelFEM.detail.TensorProductSimulator1_1`` (VoxelFEM.cc:226): bilinear quadrilaterals, plane stress."""
    N, P = 2, 1


class TensorProductSimulator2_2(_GenericSimulator):
    N, P = 2, 2


class TensorProductSimulator2_2_2(_GenericSimulator):
    """27-node hexahedra; supported by the reference templates (TPS.hh:97-110), unbound there (VoxelFEM.cc:226-229)."""
    N, P = 3, 2


class _GenericLevelView:
    def __init__(self, mg, l):
        self._mg, self._l = mg, l

    def numNodes(self):
        return self._mg._nn(self._l)

    def NbElementsPerDimension(self):
        ne = (ctypes.c_int64 * 3)()
        _lib.check(self._mg._lib.vfem_gmg_level_dims(self._mg._h, self._l, ne))
        return np.array(list(ne)[:self._mg.N], dtype=np.int64)

    def numElements(self):
        return int(np.prod(self.NbElementsPerDimension()))

    @property
    def dirichletMask(self):
        m = np.empty(self.numNodes(), dtype=np.uint8)
        _lib.check(self._mg._lib.vfem_gmg_level_dirichlet_mask(self._mg._h, self._l, m.ctypes.data_as(ctypes.c_void_p)))
        return np.stack([(m >> c) & 1 for c in range(self._mg.N)], axis=1).astype(bool)


class _GenericMultigridSolver:
    """MultigridSolver<p,..,p> on the generic path (MG.hh); same surface as ``MultigridSolver1_1_1``."""

    def __init__(self, tps, numCoarseningLevels):
        self._lib = _lib.load()
        self._tps = tps
        self.N = tps.N
        h = ctypes.c_void_p()
        _lib.check(self._lib.vfem_gmg_create(ctypes.byref(h), tps._h, int(numCoarseningLevels)))
        self._h = h
        self.L = int(numCoarseningLevels)
        self.buildBlockStiffnessMatrices = True
        self.buildFinestBlockStiffnessMatrix = False
        self.last_iterations = 0
        self.last_relative_residual = 0.0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.vfem_gmg_destroy(h)
            except Exception:
                pass
            self._h = None

    def _nn(self, l):
        n = int(self._lib.vfem_gmg_level_num_nodes(self._h, int(l)))
        if n < 0:
            raise IndexError("vector::_M_range_check")
        return n

    def getSimulator(self, l):
        self._nn(l)
        return self._tps if int(l) == 0 else _GenericLevelView(self, int(l))

    def setSymmetricGaussSeidel(self, symmetric):
        self._lib.vfem_gmg_set_symmetric_gauss_seidel(self._h, int(bool(symmetric)))

    def updateElementStiffnessMatrices(self):
        _lib.check(self._lib.vfem_gmg_update_operators(self._h, _stream()))

    updateBlockKs = updateElementStiffnessMatrices

    def applyK_device(self, l, u):
        u = _to_dev(u, (self._nn(l), self.N))
        out = torch.empty_like(u)
        _lib.check(self._lib.vfem_gmg_apply_k(self._h, int(l), _ptr(u), _ptr(out), _stream()))
        return out

    def applyK(self, l, u):
        return _to_np(self.applyK_device(l, u))

    def computeResidual_device(self, l, u, b):
        u = _to_dev(u, (self._nn(l), self.N))
        b = _to_dev(b, (self._nn(l), self.N))
        r = torch.empty_like(u)
        _lib.check(self._lib.vfem_gmg_residual(self._h, int(l), _ptr(u), _ptr(b), _ptr(r), _stream()))
        return r

    def computeResidual(self, l, u, b):
        return _to_np(self.computeResidual_device(l, u, b))

    def smoothing_device(self, l, u, b, forward=True):
        u = _to_dev(u, (self._nn(l), self.N)).clone()
        b = _to_dev(b, (self._nn(l), self.N))
        _lib.check(self._lib.vfem_gmg_smooth(self._h, int(l), _ptr(u), _ptr(b), int(bool(forward)), _stream()))
        return u

    def smoothing(self, l, u, b):
        self.updateElementStiffnessMatrices()
        return _to_np(self.smoothing_device(l, u, b, True))

    def zeroOutDirichletComponents(self, l, u):
        t = _to_dev(u, (self._nn(l), self.N)).clone()
        _lib.check(self._lib.vfem_gmg_zero_dirichlet(self._h, int(l), _ptr(t), _stream()))
        return _to_np(t)

    def restriction_device(self, fine_level, values):
        v = _to_dev(values, (self._nn(fine_level), self.N))
        out = torch.empty((self._nn(fine_level + 1), self.N), dtype=torch.float64, device=_dev())
        _lib.check(self._lib.vfem_gmg_restrict(self._h, int(fine_level), _ptr(v), _ptr(out), _stream()))
        return out

    def interpolation_device(self, fine_level, values, out=None):
        v = _to_dev(values, (self._nn(fine_level + 1), self.N))
        acc = out is not None
        if out is None:
            out = torch.empty((self._nn(fine_level), self.N), dtype=torch.float64, device=_dev())
        _lib.check(self._lib.vfem_gmg_interpolate(self._h, int(fine_level), _ptr(v), _ptr(out), int(acc), _stream()))
        return out

    def debugMulticolorVisit(self):
        """Visit order of the multicoloured sweep on level 0 (MG.hh:285-334)."""
        nn, p, N = tuple(self._tps._nn), self._tps.P, self.N
        result = np.zeros(nn, dtype=np.int32)
        counter = 0
        for lni in np.ndindex(*([p + 1] * N)):
            sl = tuple(slice(lni[d], None, (2 if lni[d] in (0, p) else 1) * p) for d in range(N))
            sub = result[sl]
            sub[...] = counter + np.arange(sub.size).reshape(sub.shape)
            counter += sub.size
        return result.reshape(-1)

    def solve_device(self, u, f, numSteps, numSmoothingSteps, stiffnessUpdated=False, zeroDirichlet=False,
                     it_callback=None, fullMultigrid=False):
        x = _to_dev(u, (self._nn(0), self.N)).clone()
        f = _to_dev(f, (self._nn(0), self.N))
        steps = [(int(numSteps), bool(stiffnessUpdated), bool(fullMultigrid))] if it_callback is None else \
            [(1, bool(stiffnessUpdated) or i > 0, bool(fullMultigrid) and i == 0) for i in range(int(numSteps))]
        for i, (n, upd, fmg) in enumerate(steps):
            _lib.check(self._lib.vfem_gmg_solve(self._h, _ptr(x), _ptr(f), n, int(numSmoothingSteps), int(upd),
                                                int(bool(zeroDirichlet)), int(fmg), _stream()))
            if it_callback is not None:
                it_callback(i, _to_np(x))
        return x

    def solve(self, u, f, numSteps, numSmoothingSteps, stiffnessUpdated=False, zeroDirichlet=False,
              it_callback=None, fullMultigrid=False):
        return _to_np(self.solve_device(u, f, numSteps, numSmoothingSteps, stiffnessUpdated, zeroDirichlet,
                                        it_callback, fullMultigrid))

    def preconditionedConjugateGradient_device(self, u, b, maxIter, tol, it_callback=None, mgIterations=1,
                                               mgSmoothingIterations=1, fullMultigrid=False, residual_cb=None):
        x = _to_dev(u).reshape(-1, self.N).clone()
        b = _to_dev(b).reshape(-1, self.N)
        if x.shape[0] != b.shape[0]:
            raise RuntimeError("x and b should have the same size")
        if x.shape[0] != self._nn(0):
            raise RuntimeError("size of input and number of nodes don't correspond")
        its, rel = ctypes.c_int(0), ctypes.c_double(0.0)

        def _cb(_user, it, rnorm):
            if residual_cb is not None:
                residual_cb(it, rnorm)
            if it_callback is not None:
                it_callback(it, _to_np(x), None)

        cb = _lib.RESIDUAL_CB(_cb) if (residual_cb is not None or it_callback is not None) else _lib.RESIDUAL_CB()
        _lib.check(self._lib.vfem_gmg_pcg(self._h, _ptr(x), _ptr(b), int(maxIter), float(tol), int(mgIterations),
                                          int(mgSmoothingIterations), int(bool(fullMultigrid)), cb, None,
                                          ctypes.byref(its), ctypes.byref(rel), _stream()))
        self.last_iterations = its.value
        self.last_relative_residual = rel.value
        return x

    def preconditionedConjugateGradient(self, u, b, maxIter, tol, it_callback=None, mgIterations=1,
                                        mgSmoothingIterations=1, fullMultigrid=False):
        return _to_np(self.preconditionedConjugateGradient_device(u, b, maxIter, tol, it_callback, mgIterations,
                                                                  mgSmoothingIterations, fullMultigrid))


def TensorProductSimulator(degreesPerDimension, domainBBox, elementsPerDimension):
    """VoxelFEM.cc:234-240 (+ the degree-2 instantiations the reference leaves commented out, :227,229)"""
    degs = [int(d) for d in degreesPerDimension]
    classes = {(1, 1, 1): TensorProductSimulator1_1_1, (1, 1): TensorProductSimulator1_1,
               (2, 2): TensorProductSimulator2_2, (2, 2, 2): TensorProductSimulator2_2_2}
    cls = classes.get(tuple(degs))
    if cls is None:
        raise RuntimeError("No template instantiation matching degreesPerDimension!")
    return cls(domainBBox, elementsPerDimension)


def TopologyOptimizationProblem(simulator, objective, constraints, filters):
    return TopologyOptimizationProblem1_1_1(simulator, objective, constraints, filters)


def ComplianceObjective(simulator):
    return ComplianceObjective1_1_1(simulator)


def MultigridComplianceObjective(mg_solver):
    return MultigridComplianceObjective1_1_1(mg_solver)


def OCOptimizer(problem):
    return OCOptimizer1_1_1(problem)


# ----------------------------------------------------------------------------------------------
# benchmark registry (VoxelFEM.cc:245-255)
# ----------------------------------------------------------------------------------------------
import time as _time

_py_timers = {}
_py_running = {}


def benchmark_reset():
    _lib.load().vfem_timers_reset()
    _py_timers.clear()
    _py_running.clear()


def benchmark_start_timer_section(name):
    _py_running[name] = _time.perf_counter()


def benchmark_stop_timer_section(name):
    t0 = _py_running.pop(name, None)
    if t0 is not None:
        s, c = _py_timers.get(name, (0.0, 0))
        _py_timers[name] = (s + _time.perf_counter() - t0, c + 1)


benchmark_start_timer = benchmark_start_timer_section
benchmark_stop_timer = benchmark_stop_timer_section


def benchmark_to_dict():
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.load().vfem_timers_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, sec, calls = line.split("\t")
        out[name] = {"seconds": float(sec), "invocations": int(calls)}
    for name, (s, c) in _py_timers.items():
        out[name] = {"seconds": s, "invocations": c}
    return out


def benchmark_report(include_messages=False):
    for name, e in sorted(benchmark_to_dict().items()):
        print("%-40s %12.6f s  (%d calls)" % (name, e["seconds"], e["invocations"]))


detail = types.ModuleType(__name__ + ".detail")
detail.TensorProductSimulator1_1_1 = TensorProductSimulator1_1_1
detail.TensorProductSimulator2_2_2 = TensorProductSimulator2_2_2
detail.TensorProductSimulator1_1 = TensorProductSimulator1_1
detail.TensorProductSimulator2_2 = TensorProductSimulator2_2
detail.MultigridSolver1_1_1 = MultigridSolver1_1_1
detail.TopologyOptimizationProblem1_1_1 = TopologyOptimizationProblem1_1_1
detail.ComplianceObjective1_1_1 = ComplianceObjective1_1_1
detail.MultigridComplianceObjective1_1_1 = MultigridComplianceObjective1_1_1
detail.OCOptimizer1_1_1 = OCOptimizer1_1_1
detail.Filter = _Filter
detail.Constraint = TotalVolumeConstraint
sys.modules[detail.__name__] = detail
