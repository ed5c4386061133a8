"""TEST INFRASTRUCTURE ONLY: degree- and dimension-generic CPU restatement of the reference's
TensorProductSimulator<Degrees...> / MultigridSolver<Degrees...> templates (VoxelFEM/TensorProductSimulator.hh,
VoxelFEM/MultigridSolver.hh), written with assembled scipy sparse matrices so that it shares no structure with
the HIP kernels it checks.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Pinning: for degree 1 this code is checked against oracle/vfem_oracle.py (the element-loop restatement that
reproduces the reference's logged 2-D and 3-D compliance values) in tests/test_oracle_kats.py; degree 2 runs
through the same code with the degree as a parameter and is additionally pinned by the analytic properties of
the element (rigid-body null space, patch test).  The reference binds no degree-2 simulator, so it holds no
numbers for it.

Conventions (TPS.hh:227, 267-271; Utilities/NDArray.hh:96-105): node grid (p*ne_d + 1) per axis, flat index
row-major with the last axis fastest; local node index likewise; dof = N*node + component.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import vfem_oracle as vo


def lagrange_table(p, xs):
    """values of the p+1 Lagrange polynomials (nodes j/p, LagrangePolynomial.hh:9,42-56) at the points xs: [len(xs), p+1]"""
    return np.array([[vo.lagrange(p, a, float(x)) for a in range(p + 1)] for x in xs])


def reference_stiffness(N, p, h, lam, mu):
    """K0 by (p+1)-point Gauss quadrature per axis (TPS.hh:127-140; TensorProductQuadrature<2p,...>)."""
    xg, wg = vo.gauss_rule(p + 1)
    Nv = np.array([[vo.lagrange(p, a, x) for x in xg] for a in range(p + 1)])       # [basis, point]
    dN = np.array([[vo.dlagrange(p, a, x) for x in xg] for a in range(p + 1)])
    loc = list(np.ndindex(*([p + 1] * N)))
    npe = len(loc)
    K = np.zeros((N * npe, N * npe))
    for q in np.ndindex(*([p + 1] * N)):
        w = np.prod([wg[i] for i in q]) * np.prod(h)
        G = np.zeros((npe, N))
        for n, l in enumerate(loc):
            for d in range(N):
                v = 1.0
                for e in range(N):
                    v *= dN[l[e], q[e]] if e == d else Nv[l[e], q[e]]
                G[n, d] = v / h[d]
        GG = G @ G.T
        blk = lam * np.einsum("ni,mj->nimj", G, G) + mu * np.einsum("nj,mi->nimj", G, G) \
            + mu * np.einsum("nm,ij->nimj", GG, np.eye(N))
        K += w * blk.reshape(N * npe, N * npe)
    return K


class GenericSim:
    """TensorProductSimulator<p,...,p> in N dimensions: grid, K0, SIMP, applyK, sensitivity, Dirichlet data, loads."""

    def __init__(self, N, p, domain, ne, young=1.0, poisson=0.3):
        self.N, self.p = int(N), int(p)
        self.ne = np.asarray(ne, dtype=np.int64)
        assert self.ne.size == N
        self.nn = p * self.ne + 1
        self.lo, self.hi = np.asarray(domain[0], float)[:N], np.asarray(domain[1], float)[:N]
        self.h = (self.hi - self.lo) / self.ne                       # TPS.hh:287
        self.E0, self.Emin, self.gamma = 1.0, 1e-9, 3.0              # TPS.hh:1392-1394
        self.num_elems, self.num_nodes = int(np.prod(self.ne)), int(np.prod(self.nn))
        self.rho = np.ones(self.num_elems)
        self.Ke = None                                               # per-element matrices on coarse levels
        self.set_isotropic(young, poisson)
        ei = np.stack(np.meshgrid(*[np.arange(n) for n in self.ne], indexing="ij"), -1).reshape(-1, N)
        loc = np.stack(np.meshgrid(*[np.arange(p + 1)] * N, indexing="ij"), -1).reshape(-1, N)
        nd = p * ei[:, None, :] + loc[None, :, :]
        self.enodes = np.ravel_multi_index(tuple(nd[..., d] for d in range(N)), tuple(self.nn))     # [ne, npe]
        self.npe = loc.shape[0]
        self.edofs = (N * self.enodes[:, :, None] + np.arange(N)[None, None, :]).reshape(self.num_elems, -1)
        self.mask = np.zeros((self.num_nodes, N), dtype=bool)
        self.dvals = np.zeros((self.num_nodes, N))
        self.loads = np.zeros((self.num_nodes, N))

    def set_isotropic(self, young, poisson):
        self.lam, self.mu = vo.lame(young, poisson, self.N)
        self.K0 = reference_stiffness(self.N, self.p, self.h, self.lam, self.mu)

    def young(self):
        return self.Emin + self.rho ** self.gamma * (self.E0 - self.Emin)      # TPS.hh:725-727

    def node_positions(self):
        idx = np.stack(np.meshgrid(*[np.arange(n) for n in self.nn], indexing="ij"), -1).reshape(-1, self.N)
        return self.lo + idx * (self.hi - self.lo) / (self.nn - 1.0)

    def apply_bc_file(self, path):
        """applyDisplacementsAndLoads (TPS.hh:358-409) for box regions: inclusive test on node positions"""
        X = self.node_positions()
        size = self.hi - self.lo
        for kind, cmask, value, lo, hi, relative in vo.parse_bc_file(path, self.N):
            lo, hi = np.asarray(lo[:self.N]), np.asarray(hi[:self.N])
            if relative:
                lo, hi = self.lo + lo * size, self.lo + hi * size
            sel = np.all((X >= lo) & (X <= hi), axis=1)
            if kind == "force":
                self.loads[sel] = np.asarray(value[:self.N]) / sel.sum()
            else:
                for c in range(self.N):
                    if cmask[c]:
                        self.mask[sel, c] = True
                        self.dvals[sel, c] = value[c]

    def element_matrices(self):
        if self.Ke is not None:
            return self.Ke
        return self.young()[:, None, None] * self.K0[None]

    def assemble(self):
        """stiffness matrix without Dirichlet treatment (what applyK multiplies by), TPS.hh:590-625"""
        Ke = self.element_matrices()
        ks = self.edofs.shape[1]
        rows = np.repeat(self.edofs, ks, axis=1).reshape(-1)
        cols = np.tile(self.edofs, (1, ks)).reshape(-1)
        n = self.N * self.num_nodes
        return sp.csr_matrix((Ke.reshape(-1), (rows, cols)), shape=(n, n))

    def apply_k(self, u):
        u = np.asarray(u, float).reshape(self.num_nodes, self.N)
        ue = u.reshape(-1)[self.edofs]
        fe = np.einsum("eij,ej->ei", self.element_matrices(), ue)
        out = np.zeros(self.N * self.num_nodes)
        np.add.at(out, self.edofs.reshape(-1), fe.reshape(-1))
        return out.reshape(self.num_nodes, self.N)

    def compliance_gradient(self, u):
        """TPS.hh:730-751"""
        ue = np.asarray(u, float).reshape(-1)[self.edofs]
        en = np.einsum("ei,ij,ej->e", ue, self.K0, ue)
        return -0.5 * self.gamma * self.rho ** (self.gamma - 1) * (self.E0 - self.Emin) * en

    def solve(self, f):
        """TPS::solve (TPS.hh:834-865) with SuperLU in place of CHOLMOD; zero Dirichlet values only"""
        K = self.assemble().tocsc()
        free = np.flatnonzero(~self.mask.reshape(-1))
        u = np.zeros(self.N * self.num_nodes)
        u[free] = spla.splu(K[free][:, free]).solve(np.asarray(f, float).reshape(-1)[free])
        return u.reshape(self.num_nodes, self.N)


def prolongation_1d(p, nce):
    """[fine nodes, coarse nodes] matrix of one axis: the coarse element's Lagrange basis at the fine node
    (MG.hh:116-141; TensorProductPolynomialInterpolant.hh:60-101)"""
    nf, nc = 2 * p * nce + 1, p * nce + 1
    W = lagrange_table(p, [t / (2.0 * p) for t in range(2 * p + 1)])
    P = np.zeros((nf, nc))
    for i in range(nf):
        e = min(i // (2 * p), nce - 1)
        P[i, p * e:p * e + p + 1] = W[i - 2 * p * e]
    return P


class GenericMG:
    """MultigridSolver<p,...,p> (MG.hh): hierarchy, coarsened Dirichlet masks, Galerkin operators, multicoloured
    block Gauss-Seidel, V-cycle, full multigrid, PCG."""

    def __init__(self, fine, num_levels):
        self.N, self.p = fine.N, fine.p
        self.sims = [fine]
        dom = (fine.lo, fine.hi)
        for l in range(1, num_levels + 1):
            ne = self.sims[-1].ne
            if np.any(ne % 2 == 1):
                raise RuntimeError("Grid size currently must be divisible by 2^numCoarseningLevels "
                                   "(nonuniform coarsening not yet implemented)")
            c = GenericSim(self.N, self.p, dom, ne // 2)
            c.lam, c.mu, c.K0 = fine.lam, fine.mu, None
            self._coarsen_dirichlet(self.sims[-1], c)
            self.sims.append(c)
        self.L = num_levels
        self.P = []
        for l in range(num_levels):
            mats = [sp.csr_matrix(prolongation_1d(self.p, int(n))) for n in self.sims[l + 1].ne]
            Pn = mats[0]
            for m in mats[1:]:
                Pn = sp.kron(Pn, m, format="csr")
            self.P.append(sp.kron(Pn, sp.identity(self.N), format="csr"))
        self.K = [None] * (num_levels + 1)
        self.symmetric_gs = True
        self.x = [np.zeros((s.num_nodes, self.N)) for s in self.sims]
        self.b = [np.zeros((s.num_nodes, self.N)) for s in self.sims]
        self._colors = [self._color_sets(s) for s in self.sims]

    def _coarsen_dirichlet(self, finer, coarser):
        """MG.hh:57-84 in integer arithmetic"""
        p, N = self.p, self.N
        idx = np.argwhere(finer.mask.any(axis=1)).reshape(-1)
        for nf in idx:
            g = np.unravel_index(nf, tuple(finer.nn))
            ranges, any_b = [], False
            for d in range(N):
                e = min(g[d] // (2 * p), coarser.ne[d] - 1)
                t = g[d] - 2 * p * e
                if t == 0:
                    ranges.append([p * e]); any_b = True
                elif t == 2 * p:
                    ranges.append([p * e + p]); any_b = True
                else:
                    ranges.append(list(range(p * e, p * e + p + 1)))
            if not any_b:
                raise RuntimeError("Dirichlet constraints on internal nodes are not supported")
            for c in np.ndindex(*[len(r) for r in ranges]):
                nc = np.ravel_multi_index(tuple(ranges[d][c[d]] for d in range(N)), tuple(coarser.nn))
                coarser.mask[nc] |= finer.mask[nf]

    def _color_sets(self, sim):
        """visitNodesMulticolored (MG.hh:285-326): colour = local node index; increment p (interior) or 2p (boundary)"""
        p, N = self.p, self.N
        out = []
        for lni in np.ndindex(*([p + 1] * N)):
            axes = []
            for d in range(N):
                inc = (2 if lni[d] in (0, p) else 1) * p
                axes.append(np.arange(lni[d], sim.nn[d], inc))
            grid = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, N)
            out.append(np.ravel_multi_index(tuple(grid[:, d] for d in range(N)), tuple(sim.nn)))
        return out

    def update_element_stiffness(self):
        """updateElementStiffnessMatrices (MG.hh:415-425): Galerkin coarse operators K_{l+1} = P^T K_l P, which is what
        assembling the coarsened per-element matrices of buildPESCoarse (MG.hh:604-669) yields"""
        self.K[0] = self.sims[0].assemble()
        for l in range(self.L):
            self.K[l + 1] = (self.P[l].T @ self.K[l] @ self.P[l]).tocsr()
        Kc = self.K[self.L].tocsc()
        free = np.flatnonzero(~self.sims[self.L].mask.reshape(-1))
        self._coarse_free = free
        self._coarse_lu = spla.splu(Kc[free][:, free])

    def apply_k(self, l, u):
        return (self.K[l] @ np.asarray(u, float).reshape(-1)).reshape(-1, self.N)

    def zero_dirichlet(self, l, u):
        u[self.sims[l].mask] = 0.0
        return u

    def residual(self, l, u, b):
        return self.zero_dirichlet(l, b - self.apply_k(l, u))          # MG.hh:401-413

    def smoothing(self, l, u, b, forward=True):
        """smoothingMulticoloredGS + m_smoothNode (MG.hh:193-265, 336-340); u updated in place"""
        N, K, sim = self.N, self.K[l], self.sims[l]
        uf = u.reshape(-1)
        cols = self._colors[l] if forward else self._colors[l][::-1]
        for nodes in cols:
            dofs = (N * nodes[:, None] + np.arange(N)[None, :])
            Krows = K[dofs.reshape(-1)]
            bms = b[nodes] - (Krows @ uf).reshape(-1, N)
            M = np.zeros((nodes.size, N, N))
            for r in range(N):
                for c in range(N):
                    M[:, r, c] = np.asarray(K[dofs[:, r], dofs[:, c]]).reshape(-1)
            free = ~sim.mask[nodes]
            diff = np.zeros((nodes.size, N))
            order = range(N) if forward else range(N - 1, -1, -1)
            for i in order:
                diff[:, i] = (bms[:, i] - np.einsum("nc,nc->n", M[:, i, :], diff)) * (free[:, i] / M[:, i, i])
            u[nodes] += diff

    def restriction(self, l, fine_vals):
        return (self.P[l].T @ fine_vals.reshape(-1)).reshape(-1, self.N)

    def interpolation(self, l, coarse_vals):
        return (self.P[l] @ coarse_vals.reshape(-1)).reshape(-1, self.N)

    def coarsest_solve(self, b):
        x = np.zeros(b.size)
        x[self._coarse_free] = self._coarse_lu.solve(b.reshape(-1)[self._coarse_free])
        return x.reshape(-1, self.N)

    def enforce_dirichlet(self, l, u, zero):
        sim = self.sims[l]                                            # MG.hh:386-398
        u[sim.mask] = 0.0 if zero else sim.dvals[sim.mask]
        return u

    def vcycle(self, l, nsmooth, residual_system):
        """MG.hh:516-553"""
        if l == self.L:
            self.x[l] = self.coarsest_solve(self.b[l])
            return
        self.enforce_dirichlet(l, self.x[l], residual_system)
        for _ in range(nsmooth):
            self.smoothing(l, self.x[l], self.b[l], True)
        self.b[l + 1] = self.restriction(l, self.residual(l, self.x[l], self.b[l]))
        self.x[l + 1] = np.zeros_like(self.x[l + 1])
        self.vcycle(l + 1, nsmooth, True)
        self.x[l] += self.interpolation(l, self.x[l + 1])
        for _ in range(nsmooth):
            self.smoothing(l, self.x[l], self.b[l], not self.symmetric_gs)

    def full_multigrid(self, l, nsmooth, residual_system):
        """MG.hh:486-508"""
        if l == self.L:
            self.x[l] = self.coarsest_solve(self.b[l])
            return
        self.b[l + 1] = self.restriction(l, self.b[l])
        self.full_multigrid(l + 1, nsmooth, residual_system)
        self.x[l] = self.interpolation(l, self.x[l + 1])
        self.vcycle(l, nsmooth, residual_system)

    def solve(self, u, f, num_steps, nsmooth, stiffness_updated=False, zero_dirichlet=False, fmg=False):
        """MG.hh:447-472"""
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
            return r                                                  # MG.hh:476-479
        return self.solve(np.zeros_like(r), r, num_steps, nsmooth, True, True, fmg).copy()

    def pcg(self, u, b, max_iter, tol, mg_iterations=1, mg_smoothing=1, fmg=False):
        """preconditionedConjugateGradient (MG.hh:679-732); the reference's loop counter is uninitialised
        (MG.hh:710) and is started at 0 here.  Returns x; iteration count in self.last_iters."""
        x = np.array(u, dtype=np.float64, copy=True)
        self.enforce_dirichlet(0, x, False)
        self.update_element_stiffness()
        b_norm_sq = float(np.sum(b * b))
        r = self.residual(0, x, b)
        rMr, d, i = 0.0, None, 0
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
        self.last_iters = i
        self.last_relres = float(np.sqrt(np.sum(r * r) / b_norm_sq)) if b_norm_sq > 0 else 0.0
        return x
