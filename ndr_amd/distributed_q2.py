"""Slab decomposition of the degree-2 (27-node hexahedra) multigrid PCG over the GPUs of one node: BASELINE config 5
(512^3 elements, MultigridSolver<2,2,2>, MG.hh) cannot live on one GPU -- 1025^3 nodes are 26 GB per nodal field and the
level-1 Galerkin element matrices (81 x 81 doubles each) 880 GB -- so every rank keeps an x-slab of every distributed level.

Same scheme as ``distributed.DistributedMGSolver`` (one process per GPU, interface planes computed by both neighbours from
identical inputs, ghost planes refreshed by neighbour messages, replicated coarse levels below), with the differences the
wider basis functions force:

* a coarse basis function of degree 2 reaches 2p - 1 = 3 fine node planes to either side, so a rank holds TWO ghost element
  layers (four ghost node planes) per neighbour on every distributed level; after a halo refresh the restriction to its
  interface plane and the relaxation of that plane need nothing else;
* the 27 colours of the sweep (MG.hh:285-334) are visited in three groups of nine that share the x index of the node in its
  element; only planes of one group change while it is visited and nodes couple within +-2 planes, so one halo refresh per
  group keeps the sweep identical to the single-process one;
* slabs start at multiples of 2^(Ld+1) elements: every local grid of a distributed level then starts at an even global
  element and the colours need no offset;
* the replicated hierarchy always receives the element matrices of its first level from the ranks (all-gather of the slabs'
  own matrices): no process ever forms level-1 matrices of the whole grid.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from .distributed import DistributedMGSolver, HaloExchanger, SlabPartition

P = 2            # polynomial degree
G = 2            # ghost element layers per neighbour


def _axis_windows(n_coarse_planes, n_fine_planes):
    """fine planes whose Dirichlet flags reach coarse plane I of one axis (MG.hh:57-84 for degree 2): a fine node on a coarse
    element boundary marks that coarse plane only, one strictly inside a coarse element marks all three planes of the
    element.  Returns (lo, hi) inclusive per coarse plane."""
    I = np.arange(n_coarse_planes)
    lo = np.where(I % 2 == 0, 2 * I - 3, 2 * I - 1)
    hi = np.where(I % 2 == 0, 2 * I + 3, 2 * I + 1)
    return np.clip(lo, 0, n_fine_planes - 1), np.clip(hi, 0, n_fine_planes - 1)


def _or_windows(a, axis, lo, hi, first_fine, first_coarse, count):
    """out[.., j, ..] = OR of a[.., lo[I]-first_fine .. hi[I]-first_fine, ..] for coarse planes I = first_coarse + j; windows
    are clipped to the planes `a` holds"""
    out_shape = list(a.shape)
    out_shape[axis] = count
    out = np.zeros(out_shape, dtype=np.uint8)
    n = a.shape[axis]
    for j in range(count):
        I = first_coarse + j
        l, h = max(int(lo[I]) - first_fine, 0), min(int(hi[I]) - first_fine, n - 1)
        if h < l:
            continue
        sl = [slice(None)] * a.ndim
        sl[axis] = slice(l, h + 1)
        dst = [slice(None)] * a.ndim
        dst[axis] = j
        out[tuple(dst)] = np.bitwise_or.reduce(a[tuple(sl)], axis=axis)
    return out


def slab_masks_and_loads(bbmin, bbmax, ne, bc_path, geom, T, device):
    """Dirichlet masks (1 byte per node, bit c = component c) of this rank's local grids on levels 0..T and its load vector,
    from the boundary-condition file alone (applyDisplacementsAndLoads, TPS.hh:358-409; coarsened masks MG.hh:57-84).
    Regions are boxes, i.e. index ranges per axis, so any range of planes can be evaluated by itself; level l is derived from
    level l-1 on a range wide enough (3 more fine planes per side and level) that the planes a rank uses are exact.
    Returns (masks[l] flat uint8 for the local grid of geom[l], loads [local nodes, 3] float64 on `device`, masks[T] again)."""
    from .pyVoxelFEM import _parse_regions
    ne = np.asarray(ne, dtype=np.int64)
    nn = P * ne + 1
    size = bbmax - bbmin
    spacing = size / (nn - 1.0)
    coords = [bbmin[d] + np.arange(nn[d]) * spacing[d] for d in range(3)]
    # plane ranges [a_l, b_l) of level l (global planes of that level) needed so that level T's local planes are exact
    need = [None] * (T + 1)
    gT = geom[T]
    need[T] = (gT.xoffn, gT.xoffn + gT.n_planes)
    for l in range(T - 1, -1, -1):
        g = geom[l]
        a = min(g.xoffn, 2 * need[l + 1][0] - 3)
        b = max(g.xoffn + g.n_planes, 2 * (need[l + 1][1] - 1) + 3 + 1)
        need[l] = (max(a, 0), min(b, int(P * (ne[0] >> l) + 1)))
    a0, b0 = need[0]
    m = np.zeros((b0 - a0, int(nn[1]), int(nn[2])), dtype=np.uint8)
    g0 = geom[0]
    loads = torch.zeros((g0.n_planes, int(nn[1]), int(nn[2]), 3), dtype=torch.float64, device=device)
    for kind, comps, value, lo, hi, relative in _parse_regions(bc_path):
        lo, hi = np.array(lo[:3]), np.array(hi[:3])
        if relative:
            lo, hi = bbmin + lo * size, bbmin + hi * size
        sel = [np.flatnonzero((coords[d] >= lo[d]) & (coords[d] <= hi[d])) for d in range(3)]
        count = int(np.prod([s.size for s in sel]))
        if kind == "force":
            if count == 0:
                raise RuntimeError("Force constraint region unmatched")
            sx = sel[0][(sel[0] >= g0.xoffn) & (sel[0] < g0.xoffn + g0.n_planes)] - g0.xoffn
            if sx.size:
                ix = [torch.as_tensor(v, device=device) for v in (sx, sel[1], sel[2])]
                for c in range(3):
                    loads[ix[0][:, None, None], ix[1][None, :, None], ix[2][None, None, :], c] = value[c] / count
        else:
            if count == 0:
                raise RuntimeError("Dirichlet region unmatched")
            if any(abs(value[c]) > 0 for c, name in enumerate("xyz") if name in comps):
                raise RuntimeError("Nonzero Dirichlet constraints currently unsupported")
            sx = sel[0][(sel[0] >= a0) & (sel[0] < b0)] - a0
            bits = sum(1 << c for c, name in enumerate("xyz") if name in comps)
            if sx.size:
                m[np.ix_(sx, sel[1], sel[2])] |= np.uint8(bits)
    masks = []
    cur, cur_first = m, a0
    for l in range(T + 1):
        g = geom[l]
        masks.append(np.ascontiguousarray(cur[g.xoffn - cur_first:g.xoffn - cur_first + g.n_planes]).reshape(-1))
        if l == T:
            break
        # MG.hh:74-76: a constrained fine node strictly inside a coarse element along every axis has no coarse counterpart
        nf = [int(P * (ne[d] >> l) + 1) for d in range(3)]
        inner = [np.flatnonzero(np.arange(cur_first if d == 0 else 0, (cur_first + cur.shape[0]) if d == 0 else nf[d]) % (2 * P) != 0)
                 for d in range(3)]
        if cur[np.ix_(*inner)].any():
            raise RuntimeError("Dirichlet constraints on internal nodes are not supported")
        nc = [int(P * (ne[d] >> (l + 1)) + 1) for d in range(3)]
        a1, b1 = need[l + 1]
        nxt = cur
        for d in range(3):
            lo_w, hi_w = _axis_windows(nc[d], nf[d])
            if d == 0:
                nxt = _or_windows(nxt, 0, lo_w, hi_w, cur_first, a1, b1 - a1)
            else:
                nxt = _or_windows(nxt, d, lo_w, hi_w, 0, 0, nc[d])
        cur, cur_first = nxt, a1
    return masks, loads.reshape(-1, 3), masks[T]


class _LevelGeomQ2:
    """local grid of one rank on level l; planes are node planes of that level (2 per element layer)"""

    def __init__(self, part, l, Ld, ne0):
        s = 2 ** l
        has_l, has_r = part.rank > 0, part.rank < part.world - 1
        self.l = l
        self.X0, self.X1 = part.x0 // s, part.x1 // s
        self.gl, self.gr = (G if has_l else 0), (G if has_r else 0)          # ghost ELEMENT layers
        self.nx = self.X1 - self.X0 + self.gl + self.gr
        self.ny, self.nz = ne0[1] // s, ne0[2] // s
        self.n_planes = P * self.nx + 1
        self.plane = (P * self.ny + 1) * (P * self.nz + 1)
        self.halo_width = P * G
        self.first_owned = P * self.gl
        self.last_owned = self.first_owned + P * (self.X1 - self.X0)
        self.xoffe = self.X0 - self.gl                                       # global element layer of local layer 0
        self.xoffn = P * self.xoffe                                          # global node plane of local plane 0
        pad = G * (2 ** (Ld - l) - 1) if l <= Ld else 0
        self.extra_lo, self.extra_hi = (pad if has_l else 0), (pad if has_r else 0)
        self.xshift = -P * self.gl if l > 0 else 0                           # fine local plane = 2 * local plane + xshift
        self.xparity = self.xoffe & 1
        self.rank, self.world = part.rank, part.world

    def reduction_weight_planes(self):
        return self.first_owned, self.last_owned + (1 if self.rank == self.world - 1 else 0)


class DistributedMGSolverQ2(DistributedMGSolver):
    """Slab-decomposed multigrid PCG for TensorProductSimulator<2,2,2> (one instance per rank); the cycles, the PCG recurrence
    and the reductions are the base class's."""

    C_DRIVER_AVAILABLE = False     # the degree-2 cycle (27 colours in three groups, four-plane halos) is driven from Python
    _MG_PREFIX = "vfem_gmg_"
    KE_DOUBLES = 81 * 81
    COLOR_GROUPS = ((0, 9), (9, 9), (18, 9))
    PARITY_AWARE_HALO = False      # 27 colours in three groups, four ghost planes: every group is followed by an exchange
    MIN_SHARDED_T = 1
    ALWAYS_ASSEMBLE = True

    MAX_AUTO_DIST_LEVELS = 2       # the element arrays of level l keep G (2^(Ld-l) - 1) padding layers per neighbour: 81 x 81
                                   # matrices on level 1, so the automatic choice stops at two coarsenings

    def __init__(self, ne, bbmin, bbmax, bc_path, material_path, num_levels, dist_levels=None, E0=1.0, Emin=1e-4,
                 gamma=3.0, group=None, proxy=None):
        """proxy = (world, rank): the slab of ONE rank of `world` built in a single process (tools/rank_proxy.py q2 ...): messages
        become device copies of the same bytes, reductions stay local, the replicated level's mask is this rank's planes of it plus a
        clamped face x = 0 (values are meaningless, work, launches and memory are the real rank's)."""
        from . import _lib
        from . import pyVoxelFEM as pv
        self._ct, self._lib_mod, self._pv = ctypes, _lib, pv
        self.lib = _lib.load()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.proxy = proxy is not None
        if self.proxy:
            self.world, self.rank = int(proxy[0]), int(proxy[1])
        self.ne = tuple(int(v) for v in ne)
        self.L = int(num_levels)
        if dist_levels is None:
            # deepest distributed level: every rank still owns >= 2 G element layers there
            dist_levels = 0
            while (dist_levels + 1 < self.L and dist_levels < self.MAX_AUTO_DIST_LEVELS
                   and self.ne[0] % (self.world * 2 ** (dist_levels + 2)) == 0
                   and self.ne[0] // (self.world * 2 ** (dist_levels + 1)) >= 2 * G):
                dist_levels += 1
        self.Ld = int(dist_levels)
        if self.Ld + 1 > self.L:
            raise RuntimeError("need at least one replicated level below the distributed ones")
        self.T = self.Ld + 1
        self.part = SlabPartition(self.ne, self.world, self.rank, align=2 ** (self.Ld + 1))
        if self.world > 1 and (self.part.x1 - self.part.x0) >> self.Ld < G:
            raise RuntimeError("slab thinner than the ghost layers on the deepest distributed level")
        self.geom = [_LevelGeomQ2(self.part, l, self.Ld, self.ne) for l in range(self.T + 1)]
        self.dev = torch.device("cuda", torch.cuda.current_device())

        # No object of the size of the whole fine grid exists anywhere: boundary conditions are evaluated for this rank's planes
        # (plus the margin the coarsened masks of its deeper levels depend on), and the replicated hierarchy is created on the
        # grid of level T itself, its level-0 element matrices imported from the ranks (vfem_gmg_import_level_ke).
        bbmin, bbmax = np.asarray(bbmin, float), np.asarray(bbmax, float)
        masks, self._loads_local, maskT = slab_masks_and_loads(bbmin, bbmax, self.ne, bc_path, self.geom, self.T, self.dev)
        self._masks = masks
        neT = [n >> self.T for n in self.ne]
        self.gsim = pv.TensorProductSimulator2_2_2([bbmin, bbmax], neT)
        self.gsim.readMaterial(material_path)
        self.gsim.E_0, self.gsim.E_min, self.gsim.gamma = E0, Emin, gamma
        gT = self.geom[self.T]
        lo_p, hi_p = gT.reduction_weight_planes()
        mine = np.ascontiguousarray(maskT.reshape(gT.n_planes, -1)[lo_p:hi_p])
        if self.proxy and self.world > 1:
            whole = np.zeros((P * neT[0] + 1, mine.shape[1]), dtype=mine.dtype)
            whole[gT.xoffn + lo_p:gT.xoffn + hi_p] = mine
            whole[0] = 7
        elif self.world > 1:
            parts = [None] * self.world
            dist.all_gather_object(parts, mine, group=group)
            whole = np.concatenate(parts, axis=0)
        else:
            whole = mine
        whole = whole.reshape(-1)
        if whole.size != self.gsim.numNodes():
            raise RuntimeError("assembled level-%d mask has %d nodes, expected %d" % (self.T, whole.size, self.gsim.numNodes()))
        self.gsim._mask = np.stack([(whole >> c) & 1 for c in range(3)], axis=1).astype(bool)
        self.gsim._push_dirichlet()
        h = ctypes.c_void_p()
        _lib.check(self.lib.vfem_gmg_create(ctypes.byref(h), self.gsim._h, self.L - self.T))
        self.gmg = h

        # local slab simulator: owned + ghost element layers as its node grid, padding layers in the density array only
        g0 = self.geom[0]
        hx = (bbmax[0] - bbmin[0]) / self.ne[0]
        lo, hi = bbmin.copy(), bbmax.copy()
        lo[0], hi[0] = bbmin[0] + g0.xoffe * hx, bbmin[0] + (g0.xoffe + g0.nx) * hx
        self.lsim = pv.TensorProductSimulator2_2_2([lo, hi], [g0.nx, self.ne[1], self.ne[2]],
                                                   _element_padding=(g0.extra_lo, g0.extra_hi))
        self.lsim.readMaterial(material_path)
        self.lsim.E_0, self.lsim.E_min, self.lsim.gamma = E0, Emin, gamma
        m0 = masks[0]
        self.lsim._mask = np.stack([(m0 >> c) & 1 for c in range(3)], axis=1).astype(bool)
        self.lsim._dvals = np.zeros((m0.size, 3))
        self.lsim._push_dirichlet()

        class _SL(ctypes.Structure):
            _fields_ = [("nx", ctypes.c_int64), ("elem_extra_lo", ctypes.c_int64), ("elem_extra_hi", ctypes.c_int64),
                        ("xshift", ctypes.c_int64), ("xparity", ctypes.c_int32)]
        arr = (_SL * len(self.geom))()
        for l, g in enumerate(self.geom):
            arr[l].nx, arr[l].elem_extra_lo, arr[l].elem_extra_hi = g.nx, g.extra_lo, g.extra_hi
            arr[l].xshift, arr[l].xparity = g.xshift, g.xparity
        mptrs = (ctypes.c_void_p * len(masks))(*[m.ctypes.data_as(ctypes.c_void_p).value for m in masks])
        h2 = ctypes.c_void_p()
        _lib.check(self.lib.vfem_gmg_create_slab(ctypes.byref(h2), self.lsim._h, len(self.geom), arr, mptrs))
        self.lmg = h2
        self.halos = [HaloExchanger(g, group, self.proxy) for g in self.geom]
        z = lambda g: torch.zeros((g.n_planes * g.plane, 3), dtype=torch.float64, device=self.dev)
        self.x = [z(g) for g in self.geom]
        self.b = [z(g) for g in self.geom]
        self.r = [z(g) for g in self.geom[:-1]]
        gT = int(self.lib.vfem_gmg_level_num_nodes(self.gmg, 0))
        self.xT = torch.zeros((gT, 3), dtype=torch.float64, device=self.dev)
        self.bT = torch.zeros((gT, 3), dtype=torch.float64, device=self.dev)
        self.symmetric_gs = True
        self.last_iterations, self.last_relative_residual = 0, 0.0

    def _export_child_level(self):
        return self.T - 1

    def _replicated_level(self):
        return 0                        # the replicated hierarchy lives on the grid of level T: T is its level 0

    def local_loads(self):
        return self._loads_local.clone()

    def set_global_densities(self, rho_global):
        """rho_global: [nx*ny*nz] float64 device tensor, identical on every rank (the rank keeps its layers of it)"""
        g = self.geom[0]
        a0, b0 = g.xoffe - g.extra_lo, g.xoffe + g.nx + g.extra_hi
        self.lsim.setElementDensities_padded(rho_global.view(self.ne[0], -1)[a0:b0].reshape(-1))
        self._sharded = True


def bench_pcg_q2(ne, levels, tol=1e-4):
    """distributed degree-2 CG-MG iterations/s (reference settings: 1 FMG cycle / iteration, 2 + 2 symmetric sweeps, operator
    update inside the solve); every rank holds its own density layers only"""
    import os
    import time
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bc = os.path.join(here, "bcs", "3d", "cantilever_flexion.bc")
    mat = os.path.join(here, "VoxelFEM", "examples", "materials", "B9Creator.material")
    ds = DistributedMGSolverQ2(ne, [0.0, 0.0, 0.0], [2.0, 1.0, 1.0], bc, mat, levels)
    layer = ne[1] * ne[2]
    g = torch.Generator(device="cuda").manual_seed(88 + ds.rank)
    own = torch.rand((ds.part.x1 - ds.part.x0) * layer, dtype=torch.float64, device="cuda", generator=g)
    ds.set_local_densities(own)
    f = ds.local_loads()
    ds.pcg(torch.zeros_like(f), f, 1, tol, 1, 2, True)              # warm-up (allocations, communicator channels)
    ds.set_local_densities(own)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    u = ds.pcg(torch.zeros_like(f), f, 100, tol, 1, 2, True)
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if dist.is_initialized():
        if dist.get_backend() != "gloo":
            dt = dt.cuda()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    return {"grid": "%dx%dx%d" % tuple(ne), "degree": 2, "nodes": int(np.prod([P * n + 1 for n in ne])), "levels": levels,
            "distributed_levels": ds.Ld + 1, "iterations": ds.last_iterations, "seconds": dt,
            "iterations_per_s": ds.last_iterations / dt, "relative_residual": ds.last_relative_residual,
            "compliance": 2.0 * ds.compliance(f, u), "densities": "sharded (owned layers per rank, seeded per rank)",
            "peak_device_GB_this_rank": torch.cuda.max_memory_allocated() / 1e9}
