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

    def __init__(self, part, group=None, proxy=False):
        self.p = part
        self.group = group
        self.proxy = proxy      # rank proxy (tools/rank_proxy.py): no peers exist; a message becomes a device copy of the same bytes

    def start(self, field, left=True, right=True):
        """Begin the exchange of the two interface planes with each neighbour (`left` / `right`: only that side) and return a
        handle for `finish`.  The transfers
        are posted as one batch of non-blocking sends / receives and run while the caller launches work that does not touch
        the ghost planes (DistributedStiffness.apply: the interior planes).  The same code serves both backends: under nccl
        (RCCL over xGMI) the plane views of the device tensor are sent as they are; under gloo (CPU tests, one-GPU
        rehearsal) a device plane is staged through host memory first and copied back in `finish`."""
        p = self.p
        if p.world == 1:
            return None
        v = field.view(p.n_planes, -1)
        w = getattr(p, "halo_width", 1)        # ghost node planes per side (degree-2 slabs: 4), contiguous in memory
        if self.proxy:
            # what a neighbour would have sent is replaced by this rank's own planes: the same number of bytes moves, on the device
            n = 0
            if p.gl and left:
                v[p.first_owned - w:p.first_owned].copy_(v[p.first_owned + 1:p.first_owned + 1 + w]); n += 1
            if p.gr and right:
                v[p.last_owned + 1:p.last_owned + 1 + w].copy_(v[p.last_owned - w:p.last_owned]); n += 1
            self.messages = getattr(self, "messages", 0) + n
            return None
        staged = field.is_cuda and dist.get_backend(self.group) == "gloo"
        ops, recvs = [], []

        def add(send_first, recv_first, peer):
            send, recv = slice(send_first, send_first + w), slice(recv_first, recv_first + w)
            sb = v[send].cpu() if staged else v[send]
            rb = torch.empty_like(sb) if staged else v[recv]
            ops.append(dist.P2POp(dist.isend, sb, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, rb, peer, self.group))
            if staged:
                recvs.append((recv, rb))

        if p.gl and left:   # left neighbour: send my planes first_owned+1 .., receive my ghost planes 0 ..
            add(p.first_owned + 1, p.first_owned - w, p.rank - 1)
        if p.gr and right:  # right neighbour: send my planes .. last_owned-1, receive my ghost planes last_owned+1 ..
            add(p.last_owned - w, p.last_owned + 1, p.rank + 1)
        if not ops:
            return None
        self.messages = getattr(self, "messages", 0) + len(ops) // 2
        return dist.batch_isend_irecv(ops), recvs, v

    def finish(self, handle):
        """wait for the transfers of `start`; afterwards the ghost planes of the field hold the neighbours' values"""
        if handle is None:
            return
        works, recvs, v = handle
        for w in works:
            w.wait()
        for plane, rb in recvs:
            v[plane].copy_(rb)

    def exchange(self, field):
        self.finish(self.start(field))

    def dot(self, a, b):
        """global sum a.b with interface planes counted once"""
        p = self.p
        lo, hi = p.reduction_weight_planes()
        av = a.view(p.n_planes, -1)[lo:hi]
        bv = b.view(p.n_planes, -1)[lo:hi]
        s = (av * bv).sum().reshape(1)
        if p.world > 1 and not self.proxy:
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

    def apply_planes(self, u, out, lo, hi):
        """output node planes lo..hi (inclusive) only"""
        from . import _lib
        from .pyVoxelFEM import _ptr, _stream
        _lib.check(self.tps._lib.vfem_sim_apply_k_planes(self.tps._h, _ptr(u), _ptr(out), int(lo), int(hi), _stream()))


class DistributedStiffness:
    """K(rho) u on the decomposed grid: halo exchange of u, then the local matrix-free apply."""

    def __init__(self, part, ops, group=None):
        self.part, self.ops = part, ops
        self.halo = HaloExchanger(part, group)

    def apply(self, u, exchange=True):
        p = self.part
        if not exchange or p.world == 1:
            return self.ops.apply(u)
        if not hasattr(self.ops, "apply_planes"):
            self.halo.exchange(u)
            return self.ops.apply(u)
        # overlap: only the first and last owned output planes see the ghost planes of u
        out = torch.empty_like(u)
        ov = out.view(p.n_planes, -1)
        works = self.halo.start(u)
        self.ops.apply_planes(u, out, p.first_owned + (1 if p.gl else 0), p.last_owned - (1 if p.gr else 0))
        if p.gl:
            ov[:p.first_owned].zero_()
        if p.gr:
            ov[p.last_owned + 1:].zero_()
        self.halo.finish(works)
        if p.gl:
            self.ops.apply_planes(u, out, p.first_owned, p.first_owned)
        if p.gr:
            self.ops.apply_planes(u, out, p.last_owned, p.last_owned)
        return out


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


def bench_pcg(ne, levels, tol=1e-4):
    """distributed CG-MG iterations/s with the reference settings (1 FMG cycle / iteration, 2+2 symmetric sweeps)"""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bc = os.path.join(here, "bcs", "3d", "cantilever_flexion.bc")
    mat = os.path.join(here, "VoxelFEM", "examples", "materials", "B9Creator.material")
    ds = DistributedMGSolver(ne, [0.0, 0.0, 0.0], [2.0, 1.0, 1.0], bc, mat, levels)
    g = torch.Generator(device="cuda").manual_seed(88)
    rho = torch.rand(ne[0] * ne[1] * ne[2], dtype=torch.float64, device="cuda", generator=g)   # same on every rank
    # (the same seeded field as the single-GPU bench, generated whole here only to cut this rank's owned layers out of it: the
    # solver itself is handed the owned layers and never sees the rest)
    own = rho.view(ne[0], -1)[ds.part.x0:ds.part.x1].reshape(-1).clone()
    del rho
    sharded = ds.T >= 2
    set_rho = (lambda: ds.set_local_densities(own)) if sharded else None
    if sharded:
        set_rho()
    else:
        g = torch.Generator(device="cuda").manual_seed(88)
        whole = torch.rand(ne[0] * ne[1] * ne[2], dtype=torch.float64, device="cuda", generator=g)
        set_rho = lambda: ds.set_global_densities(whole)
        set_rho()
    f = ds.local_loads()
    ds.pcg(torch.zeros_like(f), f, 1, tol, 1, 2, True)              # warm-up (allocations, NCCL channels)
    set_rho()                                                       # the timed solve rebuilds the coarse operators, as a design iteration does
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
    return {"grid": "%dx%dx%d" % tuple(ne), "levels": levels, "distributed_levels": ds.Ld + 1, "iterations": ds.last_iterations,
            "seconds": dt, "iterations_per_s": ds.last_iterations / dt, "relative_residual": ds.last_relative_residual,
            "compliance": 2.0 * ds.compliance(f, u), "densities": "sharded (owned layers per rank)" if sharded else "replicated",
            "includes_operator_update": True}


def bench_mlp(side=(512, 256, 256), es=1024, nn_=512, nl=4, sigma=4.0, reps=3):
    """MLP-forward voxels/s over the ranks: the density field needs no exchange, every rank evaluates the x-planes of its
    slab (a contiguous voxel range) with replicated weights; time = max over ranks, voxels = the whole grid"""
    import numpy as np
    from .mlp import MLP
    world, rank = dist.get_world_size(), dist.get_rank()
    rng = np.random.default_rng(88)
    B = (rng.standard_normal((es, 3)) * sigma).astype(np.float32)
    Ws = [rng.standard_normal((nn_, 2 * es)).astype(np.float32) / np.sqrt(2 * es)]
    Ws += [rng.standard_normal((nn_, nn_)).astype(np.float32) / np.sqrt(nn_) for _ in range(nl - 2)]
    Ws += [rng.standard_normal((1, nn_)).astype(np.float32) / np.sqrt(nn_)]
    bs = [rng.standard_normal(nn_).astype(np.float32) * 0.1 for _ in range(nl - 1)] + [np.array([0.4], np.float32)]
    m = MLP(3, 1, nn_, nl, es, sigma)
    m.load_arrays(B, Ws, bs)
    x0, x1 = rank * side[0] // world, (rank + 1) * side[0] // world
    plane = side[1] * side[2]
    m.forward_grid_range(side, x0 * plane, (x1 - x0) * plane)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = m.forward_grid_range(side, x0 * plane, (x1 - x0) * plane)
    torch.cuda.synchronize()
    dt = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64)
    chk = out.double().sum().reshape(1)
    if dist.get_backend() != "gloo":
        dt = dt.cuda()
    else:
        chk = chk.cpu()
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dist.all_reduce(chk)
    nv = side[0] * plane
    flop = 2.0 * (3 * es + 2 * es * nn_ + (nl - 2) * nn_ * nn_ + nn_) * nv
    sec = float(dt.item())
    return {"grid": "%dx%dx%d" % tuple(side), "network": "%d->%d x%d->1" % (2 * es, nn_, nl - 1), "seconds": sec,
            "voxels_per_s": nv / sec, "tflops": flop / sec / 1e12, "checksum": float(chk.item())}


def _bench_q2(world):
    """degree-2 CG-MG over the ranks (config 5's path): 256^3 elements = 135 M nodes from 2 ranks on (its level-1 element
    matrices alone are 110 GB, divided over the ranks), 512^3 = 1.08 G nodes on 8"""
    from .distributed_q2 import bench_pcg_q2
    out = []
    for q2ne, levels, need in (((256, 256, 256), 6, 2), ((512, 512, 512), 7, 8)):
        if world < need:
            continue
        try:
            torch.cuda.empty_cache()
            out.append(bench_pcg_q2(q2ne, levels))
        except RuntimeError as e:              # reported, never hidden
            out.append({"grid": "%dx%dx%d" % q2ne, "degree": 2, "error": str(e)})
    return out


def bench_apply(ne, steps, warmup, with_cg=True):
    """bench.py's N > 1 leg: K steps of {halo exchange + local apply}, max over ranks, whole-grid GVoxel/s."""
    init_process_group_from_env()
    world, rank = dist.get_world_size(), dist.get_rank()
    part = SlabPartition(ne, world, rank, align=2)
    ops = HipLocalOps(part, [0, 0, 0], [1, 1, 1])
    ops.set_densities(seeded_slab_density(part).to(ops.device))
    u = seeded_slab_field(part).to(ops.device)
    K = DistributedStiffness(part, ops)
    a_, b_ = torch.zeros_like(u), torch.zeros_like(u)      # setup: both result blocks of the caching allocator touched once
    del a_, b_
    out = None
    for _ in range(warmup):
        out = K.apply(u)
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
    del out, u, K, ops
    torch.cuda.empty_cache()
    cg = []
    if with_cg:
        for cg_ne, cg_levels in ([((256, 256, 256), 5)] + ([((512, 512, 512), 6)] if tuple(ne) == (512, 512, 512) else [])):
            try:
                cg.append(bench_pcg(cg_ne, cg_levels))
            except RuntimeError as e:          # reported, never hidden
                cg.append({"grid": "%dx%dx%d" % cg_ne, "error": str(e)})
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
        "cg_mg": cg,
        "mlp_forward": bench_mlp() if with_cg else None,
        "degree2_cg_mg": _bench_q2(world) if with_cg else None,
    }


# ==============================================================================================
# Distributed multigrid-preconditioned CG (MultigridSolver::preconditionedConjugateGradient, MG.hh:679-732,
# with vcycle / fullMultigrid of MG.hh:486-553) over x-slabs.
#
# Levels 0..Ld are distributed (every rank smooths its slab; one ghost node plane per interior side at every
# level; the element arrays of coarser levels' ghost elements are built locally from a wider padding of the fine
# moduli, so operator construction needs no communication beyond one density gather per solve).  Levels below
# are tiny: their right-hand side is gathered (sum of disjoint contributions) and every rank runs the same
# coarse cycle on a replicated hierarchy.  Per Gauss-Seidel half sweep (4 colours = one x-parity) the ghost
# planes are refreshed; dot products count interface planes once and finish with one all-reduce.
# ==============================================================================================

def auto_dist_levels(nx, world, num_levels, min_layers=8):
    """Number of coarsenings that stay distributed (Ld; levels 0 .. Ld are slab-decomposed, the rest replicated): as many as leave a
    rank at least `min_layers` owned element layers on the deepest of them and keep every slab boundary on an even plane of the next
    level.  Below that a level is launch-floor-bound whether distributed or replicated (rank proxy,
    profiles/r04_rank_proxy_levels512.jsonl: 20.6 / 20.1 / 20.0 ms per iteration with 4 / 5 / 6 distributed levels at 512^3 / 8) while
    every further level adds ~60 messages per iteration (138 / 202 / 278)."""
    ld = 0
    while ld + 1 < num_levels and nx % (world * 2 ** (ld + 2)) == 0 and nx // (world * 2 ** (ld + 1)) >= min_layers:
        ld += 1
    return ld


class _LevelGeom:
    def __init__(self, part, l, Ld, ne0):
        s = 2 ** l
        self.l = l
        self.X0, self.X1 = part.x0 // s, part.x1 // s
        self.gl, self.gr = part.gl, part.gr
        self.nx = self.X1 - self.X0 + self.gl + self.gr
        self.ny, self.nz = ne0[1] // s, ne0[2] // s
        self.n_planes = self.nx + 1
        self.plane = (self.ny + 1) * (self.nz + 1)
        self.first_owned = self.gl
        self.last_owned = self.gl + (self.X1 - self.X0)
        self.xoffn = self.X0 - self.gl
        pad = (2 ** (Ld - l) - 1) if l <= Ld else 0
        self.extra_lo, self.extra_hi = self.gl * pad, self.gr * pad
        self.xshift = -self.gl if l > 0 else 0
        self.xparity = self.xoffn & 1
        self.rank, self.world = part.rank, part.world

    # HaloExchanger duck-typing
    def reduction_weight_planes(self):
        return self.first_owned, self.last_owned + (1 if self.rank == self.world - 1 else 0)


class DistributedMGSolver:
    """Slab-decomposed multigrid PCG on the HIP kernels (one instance per rank)."""

    def __init__(self, ne, bbmin, bbmax, bc_path, material_path, num_levels, dist_levels=None, E0=1.0, Emin=1e-4,
                 gamma=3.0, group=None, proxy=None):
        """proxy = (world, rank): build the slab of ONE rank of `world` in a single process (8-rank readiness measured on one
        GPU, tools/rank_proxy.py): every message becomes a device copy of the same size out of this rank's own planes and the
        all-reduces are skipped, so the VALUES are meaningless -- only the work, the launches and the host-side cost of one
        rank's iteration are those of the real run."""
        import ctypes
        from . import _lib
        from . import pyVoxelFEM as pv
        self._ct, self._lib_mod, self._pv = ctypes, _lib, pv
        self.lib = _lib.load()
        self.group = group
        self.proxy = proxy is not None
        if self.proxy:
            self.world, self.rank = int(proxy[0]), int(proxy[1])
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ne = tuple(int(v) for v in ne)
        self.L = int(num_levels)
        if dist_levels is None:
            dist_levels = auto_dist_levels(self.ne[0], self.world, self.L, self.MIN_LAYERS)
        self.Ld = int(dist_levels)
        if self.Ld + 1 > self.L:
            raise RuntimeError("need at least one replicated level below the distributed ones")
        self.T = self.Ld + 1
        self.part = SlabPartition(self.ne, self.world, self.rank, align=2 ** (self.Ld + 1))
        self.geom = [_LevelGeom(self.part, l, self.Ld, self.ne) for l in range(self.T + 1)]
        self.dev = torch.device("cuda", torch.cuda.current_device())

        # replicated (global) simulator + hierarchy: Dirichlet masks of every level, loads, coarse cycles
        self.gsim = pv.TensorProductSimulator1_1_1([np.asarray(bbmin, float), np.asarray(bbmax, float)], list(self.ne))
        self.gsim.readMaterial(material_path)
        self.gsim.applyDisplacementsAndLoadsFromFile(bc_path)
        self.gsim.E_0, self.gsim.E_min, self.gsim.gamma = E0, Emin, gamma
        h = ctypes.c_void_p()
        _lib.check(self.lib.vfem_mg_create_partial(ctypes.byref(h), self.gsim._h, self.L, self.T))
        self.gmg = h

        # local slab simulator (node grid: owned + ghost layers; element arrays padded)
        g0 = self.geom[0]
        lo, hi = self.part.local_bbox(bbmin, bbmax)
        _lib.check(self.lib.vfem_sim_set_next_element_padding(g0.extra_lo, g0.extra_hi))
        self.lsim = pv.TensorProductSimulator1_1_1([lo, hi], [g0.nx, self.ne[1], self.ne[2]])
        self.lsim.readMaterial(material_path)
        self.lsim.E_0, self.lsim.E_min, self.lsim.gamma = E0, Emin, gamma

        # per-level local Dirichlet masks = slices of the global coarsened masks
        masks = []
        for l, g in enumerate(self.geom):
            nn = int(self.lib.vfem_mg_level_num_nodes(self.gmg, l))
            m = np.empty(nn, dtype=np.uint8)
            _lib.check(self.lib.vfem_mg_level_dirichlet_mask(self.gmg, l, m.ctypes.data_as(ctypes.c_void_p)))
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
        _lib.check(self.lib.vfem_mg_create_slab(ctypes.byref(h2), self.lsim._h, len(self.geom), arr, mptrs))
        self.lmg = h2
        self.halos = [HaloExchanger(g, group, self.proxy) for g in self.geom]
        for hx, g in zip(self.halos, self.geom):
            hx.p.world, hx.p.rank = self.world, self.rank
        z = lambda g: torch.zeros((g.n_planes * g.plane, 3), dtype=torch.float64, device=self.dev)
        self.x = [z(g) for g in self.geom]
        self.b = [z(g) for g in self.geom]
        self.r = [z(g) for g in self.geom[:-1]]
        gT = int(self.lib.vfem_mg_level_num_nodes(self.gmg, self.T))
        self.xT = torch.zeros((gT, 3), dtype=torch.float64, device=self.dev)
        self.bT = torch.zeros((gT, 3), dtype=torch.float64, device=self.dev)
        self.symmetric_gs = True
        self.overlap_sweeps = True          # relax interface planes first and exchange them behind the interior planes (where possible)
        self.last_iterations, self.last_relative_residual = 0, 0.0

    # ---- small helpers -------------------------------------------------------------------
    proxy = False                  # (instances built as a rank proxy set it; subclasses with their own constructor inherit the default)
    MIN_LAYERS = 8                 # owned element layers per rank on the deepest distributed level (automatic choice)
    _MG_PREFIX = "vfem_mg_"        # C entry points of the hierarchy handles (the degree-2 subclass uses vfem_gmg_)
    KE_DOUBLES = 576               # doubles per element matrix of the first replicated level
    COLOR_GROUPS = ((0, 4), (4, 4))   # colours between two halo refreshes: all colours of one x index
    MIN_SHARDED_T = 2              # degree 1: level 1 is virtual, the first level with stored matrices is 2
    ALWAYS_ASSEMBLE = False        # degree 2: the replicated hierarchy never derives its first level from the global moduli

    def _mg(self, name):
        return getattr(self.lib, self._MG_PREFIX + name)

    def _replicated_level(self):
        """index of level T in the replicated hierarchy's own numbering (degree 1: that hierarchy spans all levels)"""
        return self.T

    def _export_child_level(self):
        """level whose stored layers the first replicated level's matrices are built from (degree 1: level 1 is virtual, so
        level 2 is built from the moduli)"""
        return self.T - 1 if self.T >= 3 else 0

    def _s(self):
        return self._ct.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _p(self, t):
        return self._ct.c_void_p(t.data_ptr())

    def _chk(self, status):
        self._lib_mod.check(status)

    def _allreduce(self, t):
        if self.world > 1 and not self.proxy:
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=self.group)
        return t

    def local_loads(self):
        g = self.geom[0]
        f = self.gsim.buildLoadVector_device().view(self.ne[0] + 1, -1)[g.xoffn:g.xoffn + g.n_planes]
        return f.reshape(-1, 3).clone()

    def set_global_densities(self, rho_global):
        """rho_global: [nx*ny*nz] float64 device tensor, identical on every rank"""
        self._sharded = False
        self.gsim.setElementDensities(rho_global)
        g = self.geom[0]
        a0, b0 = g.xoffn - g.extra_lo, g.xoffn + g.nx + g.extra_hi
        self.lsim.setElementDensities_padded(rho_global.view(self.ne[0], -1)[a0:b0].reshape(-1))

    def set_local_densities(self, rho_owned):
        """rho_owned: the densities of this rank's OWNED element layers [x0, x1), flat [(x1 - x0) * ny * nz] float64 on the
        device.  No rank ever holds the whole field: the ghost / padding layers of the local arrays come from the two
        neighbours (one message each way), and the first replicated level gets its Galerkin element matrices from an
        all-gather of the ranks' own slabs of them (`update_operators`)."""
        g, p = self.geom[0], self.part
        layer = self.ne[1] * self.ne[2]
        own = rho_owned.reshape(p.x1 - p.x0, layer)
        lo_need, hi_need = g.gl + g.extra_lo, g.gr + g.extra_hi           # layers wanted from the left / right neighbour
        if lo_need > own.shape[0] or hi_need > own.shape[0]:
            raise RuntimeError("slab thinner than the padding of the local hierarchy")
        local = torch.empty((lo_need + own.shape[0] + hi_need, layer), dtype=torch.float64, device=own.device)
        local[lo_need:lo_need + own.shape[0]] = own
        staged = own.is_cuda and self.world > 1 and not self.proxy and dist.get_backend(self.group) == "gloo"
        ops, recvs = [], []

        def add(send, recv_slice, peer):
            if self.proxy:
                local[recv_slice].copy_(send)
                return
            sb = send.contiguous().cpu() if staged else send.contiguous()
            rb = torch.empty_like(sb) if staged else torch.empty_like(local[recv_slice])
            ops.append(dist.P2POp(dist.isend, sb, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, rb, peer, self.group))
            recvs.append((recv_slice, rb))

        # every interior interface needs the same number of layers on both sides (the padding depends on Ld only)
        if g.gl:
            add(own[:lo_need], slice(0, lo_need), self.rank - 1)          # what the left neighbour needs from me = what I need from it
        if g.gr:
            add(own[own.shape[0] - hi_need:], slice(lo_need + own.shape[0], lo_need + own.shape[0] + hi_need), self.rank + 1)
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for sl, rb in recvs:
            local[sl].copy_(rb)
        self.lsim.setElementDensities_padded(local.reshape(-1))
        self._sharded = True

    def _assemble_first_replicated_level(self):
        """element matrices of level T for the whole grid = the ranks' slabs of them, concatenated along x"""
        gT, child = self.geom[self.T], self.geom[self._export_child_level()]
        nyz = (self.ne[1] >> self.T) * (self.ne[2] >> self.T)
        count = gT.X1 - gT.X0
        KE = self.KE_DOUBLES
        mine = torch.empty(count * nyz * KE, dtype=torch.float64, device=self.dev)
        self._chk(self._mg("export_level_ke")(self.lmg, self.T, child.gl + child.extra_lo, count, self._p(mine), self._s()))
        if self.world == 1:
            whole = mine
        elif self.proxy:
            counts = [(self.part.starts[r + 1] - self.part.starts[r]) >> self.T for r in range(self.world)]
            whole = torch.cat([mine[:c * nyz * KE] if c <= count else mine.repeat(2)[:c * nyz * KE] for c in counts])
        else:
            counts = [(self.part.starts[r + 1] - self.part.starts[r]) >> self.T for r in range(self.world)]
            staged = dist.get_backend(self.group) == "gloo"
            where = "cpu" if staged else self.dev
            # equal-size buffers (slabs may differ by one aligned block; RCCL's all-gather wants one size)
            most = max(counts) * nyz * KE
            send = torch.zeros(most, dtype=torch.float64, device=where)
            send[:mine.numel()].copy_(mine)
            parts = [torch.empty(most, dtype=torch.float64, device=where) for _ in counts]
            dist.all_gather(parts, send, group=self.group)
            whole = torch.cat([b[:c * nyz * KE] for b, c in zip(parts, counts)]).to(self.dev)
        self._chk(self._mg("import_level_ke")(self.gmg, self._replicated_level(), self._p(whole), self._s()))
        torch.cuda.current_stream().synchronize()

    def update_operators(self):
        self._chk(self._mg("update_operators")(self.lmg, self._s()))
        if getattr(self, "_sharded", False) or self.ALWAYS_ASSEMBLE:
            if self.T < self.MIN_SHARDED_T:
                raise RuntimeError("sharded densities need at least two distributed levels (the first replicated level must hold "
                                   "element matrices); use set_global_densities for this configuration")
            self._assemble_first_replicated_level()
        self._chk(self._mg("update_operators")(self.gmg, self._s()))

    # ---- operators on distributed levels -----------------------------------------------------
    def halo(self, l, f):
        self.halos[l].exchange(f)

    PARITY_AWARE_HALO = True       # degree 1: a colour group changes the planes of ONE x parity (MG.hh:292-310)

    def smooth(self, l, x, b, forward):
        """one multicoloured Gauss-Seidel sweep of a distributed level.  A colour group relaxes the planes of one global x parity,
        so a neighbour's ghost plane is stale afterwards only if the plane it mirrors has that parity: the exchange after the
        OTHER group would move unchanged data and is skipped (slab boundaries are even planes on all but possibly the deepest
        distributed level: one exchange per sweep instead of two).  Where the level can be swept plane by plane (the marching
        finest-level kernel), the planes a neighbour is waiting for are relaxed first and travel while the interior planes are
        relaxed.  Same values as the blocking order: a group's planes do not read each other."""
        if not self.PARITY_AWARE_HALO:
            for first, count in self.COLOR_GROUPS:
                self._chk(self._mg("smooth_colors")(self.lmg, l, self._p(x), self._p(b), int(forward), first, count, self._s()))
                self.halo(l, x)
            return
        g, hx = self.geom[l], self.halos[l]
        by_planes = self.overlap_sweeps and bool(self.lib.vfem_mg_can_smooth_planes(self.lmg, l))
        for group, (first, count) in enumerate(self.COLOR_GROUPS):
            cx = group if forward else 1 - group                     # global x parity of the planes this group relaxes
            send_left = bool(g.gl) and ((g.xoffn + g.first_owned + 1) & 1) == cx
            send_right = bool(g.gr) and ((g.xoffn + g.last_owned - 1) & 1) == cx
            if not (send_left or send_right):
                self._chk(self._mg("smooth_colors")(self.lmg, l, self._p(x), self._p(b), int(forward), first, count, self._s()))
                continue
            if not by_planes:
                self._chk(self._mg("smooth_colors")(self.lmg, l, self._p(x), self._p(b), int(forward), first, count, self._s()))
                hx.finish(hx.start(x, send_left, send_right))
                continue
            sweep = lambda lo, hi: self._chk(self.lib.vfem_mg_smooth_group_planes(self.lmg, l, self._p(x), self._p(b), int(forward), group,
                                                                                 lo, hi, self._s()))
            # the planes the neighbours wait for, then everything else between them (the ghost planes themselves are not relaxed:
            # the exchange overwrites them)
            lo_plane, hi_plane = g.first_owned + 1, g.last_owned - 1
            inner_lo, inner_hi = g.first_owned, g.last_owned
            if send_left:
                sweep(lo_plane, lo_plane)
                inner_lo = lo_plane + 1
            if send_right and not (send_left and hi_plane == lo_plane):
                sweep(hi_plane, hi_plane)
                inner_hi = hi_plane - 1
            elif send_right:
                inner_hi = hi_plane - 1
            handle = hx.start(x, send_left, send_right)             # (stream-ordered behind the two small sweeps above)
            sweep(inner_lo, inner_hi)
            # the relaxed parity also lives on planes outside [inner_lo, inner_hi] only as ghosts, which the exchange fills
            hx.finish(handle)

    def residual(self, l, x, b, out):
        self._chk(self._mg("residual")(self.lmg, l, self._p(x), self._p(b), self._p(out), self._s()))
        return out

    def apply_k(self, d_, out):
        self._chk(self._mg("apply_k")(self.lmg, 0, self._p(d_), self._p(out), self._s()))
        self._chk(self._mg("zero_dirichlet")(self.lmg, 0, self._p(out), self._s()))
        return out

    def restrict(self, l, fine, coarse):
        self._chk(self._mg("restrict")(self.lmg, l, self._p(fine), self._p(coarse), self._s()))

    def prolong(self, l, coarse, fine, accumulate):
        self._chk(self._mg("interpolate")(self.lmg, l, self._p(coarse), self._p(fine), int(accumulate), self._s()))

    def dot(self, a, b):
        return float(self.halos[0].dot(a, b).item())

    # ---- replicated coarse cycle ---------------------------------------------------------------
    def coarse_cycle(self, fmg):
        g = self.geom[self.T]
        lo, hi = g.reduction_weight_planes()
        self.bT.zero_()
        bv = self.bT.view(-1, g.plane * 3)
        bv[g.xoffn + lo:g.xoffn + hi] = self.b[self.T].view(g.n_planes, -1)[lo:hi]
        self._allreduce(self.bT)
        self.xT.zero_()
        self._chk(self._mg("cycle_from_level")(self.gmg, self._replicated_level(), self._p(self.xT), self._p(self.bT), self._nsmooth,
                                                   int(fmg), self._s()))
        self.x[self.T].view(g.n_planes, -1).copy_(self.xT.view(-1, g.plane * 3)[g.xoffn:g.xoffn + g.n_planes])

    # ---- cycles (MG.hh:486-553) ----------------------------------------------------------------
    def vcycle(self, l):
        if l == self.T:
            self.coarse_cycle(False)
            return
        x, b, r = self.x[l], self.b[l], self.r[l]
        self._chk(self._mg("zero_dirichlet")(self.lmg, l, self._p(x), self._s()))      # residual system
        for _ in range(self._nsmooth):
            self.smooth(l, x, b, True)
        self.residual(l, x, b, r)
        self.halo(l, r)
        self.restrict(l, r, self.b[l + 1])
        self.x[l + 1].zero_()
        self.vcycle(l + 1)
        self.prolong(l, self.x[l + 1], x, True)
        self.halo(l, x)
        for _ in range(self._nsmooth):
            self.smooth(l, x, b, not self.symmetric_gs)

    def full_multigrid(self, l):
        if l == self.T:
            self.coarse_cycle(True)
            return
        self.halo(l, self.b[l])
        self.restrict(l, self.b[l], self.b[l + 1])
        self.full_multigrid(l + 1)
        self.prolong(l, self.x[l + 1], self.x[l], False)
        self.halo(l, self.x[l])
        self.vcycle(l)

    def precondition(self, r, mg_iterations, nsmooth, fmg):
        self._nsmooth = nsmooth
        self.x[0].zero_()
        self.b[0].copy_(r)
        if fmg:
            self.full_multigrid(0)
            for _ in range(1, mg_iterations):
                self.vcycle(0)
        else:
            for _ in range(mg_iterations):
                self.vcycle(0)
        return self.x[0]

    # ---- PCG (MG.hh:679-732) -------------------------------------------------------------------
    C_DRIVER_AVAILABLE = True      # degree 1: libvfem's vfem_mg_pcg_slab runs the whole solve; this class's methods are its callbacks

    def _pcg_c(self, x, b, max_iter, tol, mg_iterations, nsmooth, fmg, callback):
        """The whole solve as ONE library call (vfem_mg_pcg_slab): the cycle, the sweeps' exchange logic and the CG loop run in C++;
        Python is entered only for the halo exchanges and the all-reduces (torch.distributed), through two callbacks.  The Python
        driver below (`use_c_driver = False`) is the same algorithm call by call and remains the cross-check."""
        ct, lib = self._ct, self.lib
        from . import _lib

        class _DL(ct.Structure):
            _fields_ = [("n_planes", ct.c_int64), ("plane_nodes", ct.c_int64), ("first_owned", ct.c_int64), ("last_owned", ct.c_int64),
                        ("xoffn", ct.c_int64), ("gl", ct.c_int32), ("gr", ct.c_int32), ("x", ct.c_void_p), ("b", ct.c_void_p), ("r", ct.c_void_p)]
        if getattr(self, "_cwork", None) is None:
            self._cwork = (torch.zeros_like(self.x[0]), torch.zeros_like(self.x[0]), torch.zeros(8, dtype=torch.float64, device=self.dev))
        d, Ad, sc = self._cwork
        by_ptr = {}
        for t in self.x + self.b + self.r + [self.xT, self.bT, d, Ad, sc, x]:
            by_ptr[t.data_ptr()] = t
        arr = (_DL * len(self.geom))()
        for l, g in enumerate(self.geom):
            arr[l].n_planes, arr[l].plane_nodes, arr[l].first_owned, arr[l].last_owned = g.n_planes, g.plane, g.first_owned, g.last_owned
            arr[l].xoffn, arr[l].gl, arr[l].gr = g.xoffn, g.gl, g.gr
            arr[l].x, arr[l].b = self.x[l].data_ptr(), self.b[l].data_ptr()
            arr[l].r = self.r[l].data_ptr() if l < len(self.r) else None
        pending, failure = [None], [None]

        def halo_cb(_user, level, ptr, left, right, phase):
            try:
                hx, t = self.halos[level], by_ptr[ptr]
                if phase == 0:
                    hx.finish(hx.start(t, bool(left), bool(right)))
                elif phase == 1:
                    pending[0] = hx.start(t, bool(left), bool(right))
                else:
                    hx.finish(pending[0])
                    pending[0] = None
                return 0
            except BaseException as e:                       # an exception must not unwind through the C frames
                failure[0] = e
                return 1

        def allreduce_cb(_user, ptr, n):
            try:
                base = ptr if ptr in by_ptr else sc.data_ptr()          # (scalars: an address inside the 8-double block)
                t = by_ptr[base].view(-1)
                off = (ptr - base) // 8
                self._allreduce(t[off:off + n])
                return 0
            except BaseException as e:
                failure[0] = e
                return 1

        HALO = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int, ct.c_int)
        ALLR = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int64)
        hcb, acb = HALO(halo_cb), ALLR(allreduce_cb)
        rcb = _lib.RESIDUAL_CB((lambda _u, it, rn: callback(it, rn))) if callback else _lib.RESIDUAL_CB()
        its, rel = ct.c_int(0), ct.c_double(0.0)
        status = lib.vfem_mg_pcg_slab(self.lmg, self.gmg, self.T, arr, self.rank, self.world, self._p(self.xT), self._p(self.bT),
                                      self._p(x), self._p(b), self._p(d), self._p(Ad), self._p(sc), int(max_iter), float(tol),
                                      int(mg_iterations), int(nsmooth), int(bool(fmg)), int(bool(self.overlap_sweeps)),
                                      ct.cast(hcb, ct.c_void_p), ct.cast(acb, ct.c_void_p), None, rcb, None, ct.byref(its), ct.byref(rel), self._s())
        if failure[0] is not None:
            raise failure[0]
        self._chk(status)
        self.last_iterations, self.last_relative_residual = its.value, rel.value
        return x

    def pcg(self, x, b, max_iter, tol, mg_iterations=1, nsmooth=1, fmg=False, callback=None):
        self._nsmooth = nsmooth
        if self.C_DRIVER_AVAILABLE and getattr(self, "use_c_driver", True):
            self._mg("set_symmetric_gauss_seidel")(self.lmg, int(bool(self.symmetric_gs)))
            self._mg("set_symmetric_gauss_seidel")(self.gmg, int(bool(self.symmetric_gs)))
            self._chk(self._mg("zero_dirichlet")(self.lmg, 0, self._p(x), self._s()))
            self.update_operators()
            return self._pcg_c(x, b, max_iter, tol, mg_iterations, nsmooth, fmg, callback)
        self._chk(self._mg("zero_dirichlet")(self.lmg, 0, self._p(x), self._s()))      # zero Dirichlet values only
        self.update_operators()
        bb = self.dot(b, b)
        self.halo(0, x)
        r = torch.empty_like(x)
        self.residual(0, x, b, r)
        rr = self.dot(r, r)
        d = torch.zeros_like(x)
        Ad = torch.empty_like(x)
        rMr, it = 0.0, 0
        while it < max_iter and rr > tol * tol * bb:
            it += 1
            s = self.precondition(r, mg_iterations, nsmooth, fmg) if nsmooth > 0 else r.clone()
            self._chk(self._mg("zero_dirichlet")(self.lmg, 0, self._p(s), self._s()))
            rMr_old, rMr = rMr, self.dot(r, s)
            if it == 1:
                d.copy_(s)
            else:
                d.mul_(rMr / rMr_old).add_(s)
            self.halo(0, d)
            self.apply_k(d, Ad)
            alpha = rMr / self.dot(d, Ad)
            x.add_(d, alpha=alpha)
            r.add_(Ad, alpha=-alpha)
            rr = self.dot(r, r)
            if callback:
                callback(it, rr ** 0.5)
        self.last_iterations = it
        self.last_relative_residual = (rr / bb) ** 0.5 if bb > 0 else 0.0
        return x

    def compliance(self, f, u):
        return 0.5 * self.dot(f, u)

    def compliance_gradient(self, u):
        """sensitivity d(1/2 f.u)/d(rho_e) (TPS::complianceGradient, TPS.hh:730-751) for the element layers this rank owns,
        flat [owned_layers * ny * nz]; element-local, so after one halo refresh of u no further communication"""
        self.halo(0, u)
        g0 = self.geom[0]
        g = self.lsim.complianceGradient_device(u).view(g0.nx, -1)
        return g[g0.gl:g0.gl + (self.part.x1 - self.part.x0)].reshape(-1)

    def owned_element_range(self):
        """(first, count) of this rank's elements in the global flat element order (x-layers are contiguous)"""
        layer = self.ne[1] * self.ne[2]
        return self.part.x0 * layer, (self.part.x1 - self.part.x0) * layer


# ==============================================================================================
# One evaluation of the train_xdg closure (training/train_xdg.py:282-329) over the slab ranks:
#   MLP logits of the rank's planes -> constrained sigmoid (mean over the WHOLE field) -> compliance by the distributed
#   MG-PCG -> sensitivities of the owned elements -> MLP backward on the rank's planes -> one all-reduce of the gradients.
# The density field crosses ranks once per evaluation (all-gather: the slab solver builds its replicated coarse
# hierarchy from the whole field); displacements, sensitivities and activations never do.
# ==============================================================================================
class _ShardedCompliance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, density_local, trainer):
        ds = trainer.solver
        parts = trainer._gather(density_local.detach().to(torch.float64))
        ds.set_global_densities(parts)
        f = ds.local_loads()
        if trainer._u is None or trainer.zero_init:
            trainer._u = torch.zeros_like(f)
        trainer._u = ds.pcg(trainer._u, f, trainer.cg_iter, trainer.tol, 1, 2, True)
        # as in the reference's autograd node the value is 2 J but the gradient is that of J (fem.py:122-126)
        ctx.save_for_backward(ds.compliance_gradient(trainer._u).to(torch.float32))
        return density_local.new_tensor(2.0 * ds.compliance(f, trainer._u))

    @staticmethod
    def backward(ctx, grad_output):
        (g,) = ctx.saved_tensors
        return g * grad_output, None


class DistributedDensityTrainer:
    """density = constrained_sigmoid(mlp(grid)) sharded by x-planes; loss = compliance of the distributed solve"""

    def __init__(self, solver, net, max_volume, tol=1e-4, cg_iter=100, zero_init=False):
        self.solver, self.net, self.max_volume = solver, net, float(max_volume)
        self.tol, self.cg_iter, self.zero_init = tol, cg_iter, zero_init
        self.first, self.count = solver.owned_element_range()
        net.set_grid(solver.ne, voxel_range=(self.first, self.count))
        self._u = None
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def _reduce(self, t, op):
        if self.world == 1:
            return t
        if dist.get_backend() == "gloo":
            h = t.detach().cpu()
            dist.all_reduce(h, op=op)
            return h.to(t.device)
        t = t.detach().clone()
        dist.all_reduce(t, op=op)
        return t

    def _gather(self, local):
        if self.world == 1:
            return local
        counts = [None] * self.world
        dist.all_gather_object(counts, int(local.numel()))
        m = max(counts)                                   # equal-size buffers (slabs differ by at most one aligned block)
        gloo = dist.get_backend() == "gloo"
        mine = torch.zeros(m, dtype=local.dtype, device="cpu" if gloo else local.device)
        mine[:local.numel()] = local.cpu() if gloo else local
        bufs = [torch.empty_like(mine) for _ in counts]
        dist.all_gather(bufs, mine)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)]).to(local.device)

    def loss(self):
        """compliance (2 J) of the constrained density predicted by the network; call .backward() on it"""
        from . import fem
        logits = self.net.forward_grid()
        density = fem.sigmoid_with_constrained_mean(
            logits, torch.tensor(self.max_volume, device=logits.device),
            allsum=lambda t: self._reduce(t, dist.ReduceOp.SUM), allmax=lambda t: self._reduce(t, dist.ReduceOp.MAX))
        self.last_density = density.detach()
        return _ShardedCompliance.apply(density, self)
