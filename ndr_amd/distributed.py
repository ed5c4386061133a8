"""Domain decomposition of the voxel grid over the GPUs of one node.

One process per GPU (`torch.distributed`; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  The element grid is cut into x-slabs: x is the slowest axis, so a node plane is one contiguous run of
NY*NZ*3 doubles and a halo is a single message per neighbour.  Every rank stores its owned element layers plus
one ghost element layer (= one ghost node plane) towards each neighbour and computes complete results for all
of its owned node planes, interface planes included (both neighbours compute them, identically), so one
exchange of the *input* field per operator application is enough and no partial sums are ever reduced.
Dot products count an interface plane once (the lower rank owns it).

The per-rank numerics are behind `LocalOps`; the product implementation is `HipLocalOps` (libvfem kernels on
device tensors).  Tests substitute an oracle-backed implementation to exercise this host logic on CPU.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist


class SlabPartition:
    """x-slab partition of `ne` = (nx, ny, nz) elements over `world` ranks, aligned to `align` elements so that
    every rank can coarsen its slab locally (`align` = 2^levels)."""

    def __init__(self, ne, world, rank, align=1):
        nx = int(ne[0])
        if nx % align != 0:
            raise RuntimeError("nx must be divisible by the alignment %d" % align)
        units = nx // align
        if units < world:
            raise RuntimeError("grid too small for %d ranks at alignment %d" % (world, align))
        base, rem = divmod(units, world)
        starts = [0]
        for r in range(world):
            starts.append(starts[-1] + (base + (1 if r < rem else 0)) * align)
        self.ne = tuple(int(v) for v in ne)
        self.world, self.rank = world, rank
        self.starts = starts
        self.x0, self.x1 = starts[rank], starts[rank + 1]          # owned element layers [x0, x1)
        self.gl = 1 if rank > 0 else 0                              # ghost element layers
        self.gr = 1 if rank < world - 1 else 0
        self.local_ne = (self.x1 - self.x0 + self.gl + self.gr, self.ne[1], self.ne[2])
        self.plane = (self.ne[1] + 1) * (self.ne[2] + 1)            # nodes per x-plane
        self.n_planes = self.local_ne[0] + 1
        # local plane indices: [first_owned, last_owned] are the planes this rank computes; it "owns" (for
        # reductions) first_owned..last_owned minus the upper interface plane, which the next rank counts
        self.first_owned = self.gl
        self.last_owned = self.gl + (self.x1 - self.x0)

    def local_bbox(self, bbmin, bbmax):
        bbmin, bbmax = np.asarray(bbmin, float), np.asarray(bbmax, float)
        h = (bbmax[0] - bbmin[0]) / self.ne[0]
        lo, hi = bbmin.copy(), bbmax.copy()
        lo[0] = bbmin[0] + (self.x0 - self.gl) * h
        hi[0] = bbmin[0] + (self.x1 + self.gr) * h
        return lo, hi

    def element_slice(self):
        """global element layers held locally (ghost layers included)"""
        return slice(self.x0 - self.gl, self.x1 + self.gr)

    def node_slice(self):
        return slice(self.x0 - self.gl, self.x1 + self.gr + 1)

    def reduction_weight_planes(self):
        """(lo, hi) local plane range [lo, hi) counted by this rank in global dot products"""
        hi = self.last_owned + (1 if self.rank == self.world - 1 else 0)
        return self.first_owned, hi


class HaloExchanger:
    """Refreshes the ghost node planes of a nodal field [n_planes * plane, 3] from the neighbours."""

    def __init__(self, part, group=None):
        self.p = part
        self.group = group

    def exchange(self, field):
        p = self.p
        if p.world == 1:
            return
        v = field.view(p.n_planes, -1)
        # gloo moves host memory: stage device planes through the CPU (test / single-GPU rehearsal path only;
        # with the nccl backend the planes go GPU-to-GPU over xGMI)
        staged = field.is_cuda and dist.get_backend(self.group) == "gloo"
        ops, recvs = [], []

        def add(send_plane, recv_plane, peer):
            sb = v[send_plane].cpu() if staged else v[send_plane]
            rb = torch.empty_like(sb) if staged else v[recv_plane]
            ops.append(dist.P2POp(dist.isend, sb, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, rb, peer, self.group))
            if staged:
                recvs.append((recv_plane, rb))

        if p.gl:   # left neighbour: send my plane first_owned+1, receive my ghost plane 0
            add(p.first_owned + 1, 0, p.rank - 1)
        if p.gr:   # right neighbour: send my plane last_owned-1, receive my ghost plane last_owned+1
            add(p.last_owned - 1, p.last_owned + 1, p.rank + 1)
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for plane, rb in recvs:
            v[plane].copy_(rb)

    def dot(self, a, b):
        """global sum a.b with interface planes counted once"""
        p = self.p
        lo, hi = p.reduction_weight_planes()
        av = a.view(p.n_planes, -1)[lo:hi]
        bv = b.view(p.n_planes, -1)[lo:hi]
        s = (av * bv).sum().reshape(1)
        if p.world > 1:
            if s.is_cuda and dist.get_backend(self.group) == "gloo":
                h = s.cpu()
                dist.all_reduce(h, group=self.group)
                s = h.to(s.device)
            else:
                dist.all_reduce(s, group=self.group)
        return s


class HipLocalOps:
    """Per-rank operator: the libvfem simulator of the local slab (owned + ghost layers)."""

    def __init__(self, part, bbmin, bbmax, young=1.0, poisson=0.3, E0=1.0, Emin=1e-4, gamma=3.0):
        from . import pyVoxelFEM as pv
        lo, hi = part.local_bbox(bbmin, bbmax)
        self.tps = pv.TensorProductSimulator1_1_1([lo, hi], list(part.local_ne))
        from . import _lib
        _lib.check(self.tps._lib.vfem_sim_set_isotropic(self.tps._h, young, poisson))
        self.tps.E_0, self.tps.E_min, self.tps.gamma = E0, Emin, gamma
        self.device = torch.device("cuda", torch.cuda.current_device())

    def set_densities(self, rho_local):
        self.tps.setElementDensities(rho_local)

    def apply(self, u):
        return self.tps.applyK_device(u)


class DistributedStiffness:
    """K(rho) u on the decomposed grid: halo exchange of u, then the local matrix-free apply."""

    def __init__(self, part, ops, group=None):
        self.part, self.ops = part, ops
        self.halo = HaloExchanger(part, group)

    def apply(self, u, exchange=True):
        if exchange:
            self.halo.exchange(u)
        return self.ops.apply(u)


def init_process_group_from_env():
    if dist.is_initialized():
        return
    backend = "nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= int(os.environ.get("WORLD_SIZE", "1")) else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend)


def seeded_slab_density(part, seed=88):
    """the rank's slab of a global seeded U[0,1] density field, generated layer by layer (x-layer l is seeded
    with seed + l) so that no rank ever holds the whole field"""
    ny, nz = part.ne[1], part.ne[2]
    sl = part.element_slice()
    out = torch.empty((sl.stop - sl.start, ny, nz), dtype=torch.float64)
    for i, layer in enumerate(range(sl.start, sl.stop)):
        g = torch.Generator().manual_seed(seed * 1000003 + layer)
        out[i] = torch.rand((ny, nz), dtype=torch.float64, generator=g)
    return out.reshape(-1)


def seeded_slab_field(part, seed=7):
    ny1, nz1 = part.ne[1] + 1, part.ne[2] + 1
    sl = part.node_slice()
    out = torch.empty((sl.stop - sl.start, ny1 * nz1 * 3), dtype=torch.float64)
    for i, pl in enumerate(range(sl.start, sl.stop)):
        g = torch.Generator().manual_seed(seed * 7919 + pl)
        out[i] = torch.randn(ny1 * nz1 * 3, dtype=torch.float64, generator=g)
    return out.reshape(-1, 3)


def bench_apply(ne, steps, warmup):
    """bench.py's N > 1 leg: K steps of {halo exchange + local apply}, max over ranks, whole-grid GVoxel/s."""
    init_process_group_from_env()
    world, rank = dist.get_world_size(), dist.get_rank()
    part = SlabPartition(ne, world, rank, align=2)
    ops = HipLocalOps(part, [0, 0, 0], [1, 1, 1])
    ops.set_densities(seeded_slab_density(part).to(ops.device))
    u = seeded_slab_field(part).to(ops.device)
    K = DistributedStiffness(part, ops)
    for _ in range(warmup):
        K.apply(u)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = K.apply(u)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if dist.get_backend() != "gloo":
        dt = dt.to(ops.device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    wall = float(dt.item())
    chk = K.halo.dot(out, out)
    nvox = ne[0] * ne[1] * ne[2]
    nn = (ne[0] + 1) * (ne[1] + 1) * (ne[2] + 1)
    ab = 2 * nn * 24 + nvox * 8
    return {
        "metric": "matrix-free SpMV GVoxel/s (Q1 fp64, %dx%dx%d); CG-MG iterations/s reported in cg_mg" % tuple(ne),
        "value": nvox / (wall / steps) / 1e9, "unit": "GVoxel/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "K(rho) u on a %dx%dx%d voxel grid, trilinear hexahedra, fp64, x-slab decomposition "
                               "with one halo exchange of u per step" % tuple(ne),
                   "grid": list(ne), "parallelism": "slab%d" % world},
        "roofline": {"bound": "hbm", "achieved": ab / (wall / steps) / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                     "frac": ab / (wall / steps) / 1e9 / (8000.0 * world), "traffic": None,
                     "note": "whole-step time (halo exchange included), aggregate peak of all GPUs"},
        "checksum_KuKu": float(chk.item()),
    }
