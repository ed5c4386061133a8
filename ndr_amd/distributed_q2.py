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

    _MG_PREFIX = "vfem_gmg_"
    KE_DOUBLES = 81 * 81
    COLOR_GROUPS = ((0, 9), (9, 9), (18, 9))
    MIN_SHARDED_T = 1
    ALWAYS_ASSEMBLE = True

    def __init__(self, ne, bbmin, bbmax, bc_path, material_path, num_levels, dist_levels=None, E0=1.0, Emin=1e-4,
                 gamma=3.0, group=None):
        from . import _lib
        from . import pyVoxelFEM as pv
        self._ct, self._lib_mod, self._pv = ctypes, _lib, pv
        self.lib = _lib.load()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ne = tuple(int(v) for v in ne)
        self.L = int(num_levels)
        if dist_levels is None:
            # deepest distributed level: every rank still owns >= 2 G element layers there
            dist_levels = 0
            while (dist_levels + 1 < self.L and self.ne[0] % (self.world * 2 ** (dist_levels + 2)) == 0
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

        # replicated (global) simulator: grid, material, boundary conditions -> Dirichlet masks of every level and the loads;
        # its hierarchy holds fields and operators from level T down only
        self.gsim = pv.TensorProductSimulator2_2_2([np.asarray(bbmin, float), np.asarray(bbmax, float)], list(self.ne))
        self.gsim.readMaterial(material_path)
        self.gsim.applyDisplacementsAndLoadsFromFile(bc_path)
        self.gsim.E_0, self.gsim.E_min, self.gsim.gamma = E0, Emin, gamma
        h = ctypes.c_void_p()
        _lib.check(self.lib.vfem_gmg_create_partial(ctypes.byref(h), self.gsim._h, self.L, self.T))
        self.gmg = h

        # local slab simulator: owned + ghost element layers as its node grid, padding layers in the density array only
        g0 = self.geom[0]
        bbmin, bbmax = np.asarray(bbmin, float), np.asarray(bbmax, float)
        hx = (bbmax[0] - bbmin[0]) / self.ne[0]
        lo, hi = bbmin.copy(), bbmax.copy()
        lo[0], hi[0] = bbmin[0] + g0.xoffe * hx, bbmin[0] + (g0.xoffe + g0.nx) * hx
        self.lsim = pv.TensorProductSimulator2_2_2([lo, hi], [g0.nx, self.ne[1], self.ne[2]],
                                                   _element_padding=(g0.extra_lo, g0.extra_hi))
        self.lsim.readMaterial(material_path)
        self.lsim.E_0, self.lsim.E_min, self.lsim.gamma = E0, Emin, gamma

        masks = []
        for l, g in enumerate(self.geom):
            nn = int(self.lib.vfem_gmg_level_num_nodes(self.gmg, l))
            m = np.empty(nn, dtype=np.uint8)
            _lib.check(self.lib.vfem_gmg_level_dirichlet_mask(self.gmg, l, m.ctypes.data_as(ctypes.c_void_p)))
            m = m.reshape(-1, g.plane)[g.xoffn:g.xoffn + g.n_planes]
            masks.append(np.ascontiguousarray(m.reshape(-1)))
        self._masks = masks
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
        self.halos = [HaloExchanger(g, group) for g in self.geom]
        z = lambda g: torch.zeros((g.n_planes * g.plane, 3), dtype=torch.float64, device=self.dev)
        self.x = [z(g) for g in self.geom]
        self.b = [z(g) for g in self.geom]
        self.r = [z(g) for g in self.geom[:-1]]
        gT = int(self.lib.vfem_gmg_level_num_nodes(self.gmg, self.T))
        self.xT = torch.zeros((gT, 3), dtype=torch.float64, device=self.dev)
        self.bT = torch.zeros((gT, 3), dtype=torch.float64, device=self.dev)
        self.symmetric_gs = True
        self.last_iterations, self.last_relative_residual = 0, 0.0

    def _export_child_level(self):
        return self.T - 1

    def local_loads(self):
        g = self.geom[0]
        f = self.gsim.buildLoadVector_device().view(P * self.ne[0] + 1, -1)[g.xoffn:g.xoffn + g.n_planes]
        return f.reshape(-1, 3).clone()

    def set_global_densities(self, rho_global):
        """rho_global: [nx*ny*nz] float64 device tensor, identical on every rank (the rank keeps its layers of it)"""
        g = self.geom[0]
        a0, b0 = g.xoffe - g.extra_lo, g.xoffe + g.nx + g.extra_hi
        self.lsim.setElementDensities_padded(rho_global.view(self.ne[0], -1)[a0:b0].reshape(-1))
        self._sharded = True
