"""world_size-2 gloo test of the x-slab decomposition host logic (partition, halo exchange, owned-plane
reductions).  The per-rank numerics are the CPU oracle here, so the test runs without a GPU; on the GPU box
the same classes drive libvfem (tests/test_gpu_distributed.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleLocalOps:
    def __init__(self, part, bbmin, bbmax):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import make_oracle
        lo, hi = part.local_bbox(bbmin, bbmax)
        self.sim = make_oracle(part.local_ne, (lo, hi), None)

    def set_densities(self, rho):
        self.sim.set_densities(rho.numpy())

    def apply(self, u):
        return torch.from_numpy(self.sim.apply_k(u.numpy()))


class OracleLocalOpsWithPlanes(OracleLocalOps):
    """adds the plane-range apply that lets DistributedStiffness overlap the halo exchange with the interior planes, and records the
    calls; the worker logs the exchange's start / finish into the same list (the overlap contract is an ORDER: interior planes
    between start and finish, the two boundary planes after finish -- whether the data has already landed when the interior
    planes are computed is up to the transport, and on a loopback it sometimes has)"""

    def __init__(self, part, bbmin, bbmax):
        super().__init__(part, bbmin, bbmax)
        self.part, self.calls = part, []

    def apply_planes(self, u, out, lo, hi):
        p = self.part
        self.calls.append((int(lo), int(hi)))
        full = self.apply(u).view(p.n_planes, -1)
        out.view(p.n_planes, -1)[lo:hi + 1] = full[lo:hi + 1]


def _worker(rank, world, port, ne, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ndr_amd import distributed as vd
    from helpers import make_oracle
    dom = ([0, 0, 0], [2, 1, 1])
    part = vd.SlabPartition(ne, world, rank, align=2)
    ops = OracleLocalOps(part, *dom)
    ops.set_densities(vd.seeded_slab_density(part))
    u = vd.seeded_slab_field(part)
    # poison the ghost planes: the exchange must repair them
    uv = u.view(part.n_planes, -1)
    if part.gl:
        uv[0] = 1e30
    if part.gr:
        uv[-1] = -1e30
    K = vd.DistributedStiffness(part, ops)
    out = K.apply(u)
    nrm = K.halo.dot(out, out)
    # the overlapped sequence (start -> interior planes -> finish -> boundary planes) against the blocking one, same input
    u2 = vd.seeded_slab_field(part)
    u2v = u2.view(part.n_planes, -1)
    if part.gl:
        u2v[0] = 1e30
    if part.gr:
        u2v[-1] = -1e30
    ops2 = OracleLocalOpsWithPlanes(part, *dom)
    ops2.set_densities(vd.seeded_slab_density(part))
    K2 = vd.DistributedStiffness(part, ops2)
    start, finish = K2.halo.start, K2.halo.finish
    K2.halo.start = lambda f, *a, **k: (ops2.calls.append("start"), start(f, *a, **k))[1]
    K2.halo.finish = lambda h: (finish(h), ops2.calls.append("finish"))[0]
    out2 = K2.apply(u2)
    own = slice(part.first_owned, part.last_owned + 1)
    assert torch.equal(out2.view(part.n_planes, -1)[own], out.view(part.n_planes, -1)[own])
    assert torch.equal(u2, u)                                       # both paths leave the same (repaired) ghost planes
    if world > 1:
        interior = (part.first_owned + (1 if part.gl else 0), part.last_owned - (1 if part.gr else 0))
        assert ops2.calls[:3] == ["start", interior, "finish"], ops2.calls           # interior planes while the planes travel
        assert all(c[0] == c[1] for c in ops2.calls[3:]) and len(ops2.calls) == 3 + part.gl + part.gr, ops2.calls   # boundary planes after
    # global reference on every rank
    full = vd.SlabPartition(ne, 1, 0)
    g = make_oracle(ne, dom, None, vd.seeded_slab_density(full).numpy())
    ref = g.apply_k(vd.seeded_slab_field(full).numpy()).reshape(ne[0] + 1, -1)
    mine = out.numpy().reshape(part.n_planes, -1)[part.first_owned:part.last_owned + 1]
    want = ref[part.x0:part.x1 + 1]
    err = np.abs(mine - want).max() / np.abs(want).max()
    q.put((rank, float(err), float(nrm.item()), float((ref * ref).sum())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ne", [(2, (8, 4, 6)), (3, (12, 4, 4))])
def test_slab_apply_matches_global(world, ne):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = __import__('helpers').free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ne, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = __import__('helpers').collect_from_ranks(q, procs, 240)
    for rank, err, nrm, ref in res:
        assert err < 1e-12, (rank, err)
        assert abs(nrm - ref) < 1e-10 * ref, (rank, nrm, ref)


def test_partition_properties():
    sys.path.insert(0, ROOT)
    from ndr_amd.distributed import SlabPartition
    for world in (1, 2, 4, 8):
        parts = [SlabPartition((512, 256, 256), world, r, align=64) for r in range(world)]
        assert parts[0].x0 == 0 and parts[-1].x1 == 512
        for a, b in zip(parts, parts[1:]):
            assert a.x1 == b.x0 and a.x1 % 64 == 0
        planes = sum(p.reduction_weight_planes()[1] - p.reduction_weight_planes()[0] for p in parts)
        assert planes == 513
    with pytest.raises(RuntimeError):
        SlabPartition((6, 4, 4), 4, 0, align=2)


def _halo4_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ndr_amd import distributed as vd
    from ndr_amd.distributed_q2 import G, P, _LevelGeomQ2
    ne = (8 * world, 2, 3)
    part = vd.SlabPartition(ne, world, rank, align=4)
    g = _LevelGeomQ2(part, 0, 1, ne)
    # every global node plane carries its own index; ghost planes start poisoned
    field = torch.empty((g.n_planes * g.plane, 3), dtype=torch.float64)
    v = field.view(g.n_planes, -1)
    for pl in range(g.n_planes):
        v[pl] = float(g.xoffn + pl)
    if g.gl:
        v[:g.first_owned] = -1e30
    if g.gr:
        v[g.last_owned + 1:] = 1e30
    vd.HaloExchanger(g).exchange(field)          # CPU tensors: the plane views go out as they are (the path RCCL takes)
    want = torch.arange(g.xoffn, g.xoffn + g.n_planes, dtype=torch.float64)
    ok = bool(torch.equal(v[:, 0], want)) and bool(torch.equal(v.min(dim=1).values, want))
    ones = torch.ones_like(field)
    cnt = float(vd.HaloExchanger(g).dot(ones, ones).item())       # every global node counted once
    q.put((rank, ok, g.halo_width == P * G, cnt, 3.0 * (P * ne[0] + 1) * g.plane))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_four_plane_halo_of_the_degree2_slabs(world):
    """the degree-2 slab geometry (two ghost element layers = four ghost node planes per neighbour) through the same
    HaloExchanger: ghost planes receive the neighbours' owned planes, reductions count every plane once"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = __import__('helpers').free_port()
    procs = [ctx.Process(target=_halo4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = __import__('helpers').collect_from_ranks(q, procs, 240)
    for rank, ok, width_ok, cnt, total in res:
        assert ok and width_ok, rank
        assert cnt == total, (cnt, total)


def test_automatic_number_of_distributed_levels():
    """levels stay distributed while a rank keeps at least eight owned element layers on them and slab boundaries stay on even planes
    of the next level (ndr_amd.distributed.auto_dist_levels; the rule the rank proxy measurements of round 4 led to)"""
    from ndr_amd.distributed import auto_dist_levels
    assert auto_dist_levels(512, 8, 6) == 3            # 64 layers per rank: 32, 16, 8 on levels 1 .. 3
    assert auto_dist_levels(256, 8, 5) == 2
    assert auto_dist_levels(256, 2, 5) == 4            # 128 layers per rank: limited by the number of levels
    assert auto_dist_levels(32, 2, 3) == 1 and auto_dist_levels(48, 3, 3) == 1 and auto_dist_levels(64, 4, 4) == 1
    assert auto_dist_levels(24, 3, 3) == 0             # eight layers per rank: nothing to coarsen in place
    assert auto_dist_levels(512, 8, 6, min_layers=2) == 5            # (the rule of rounds 2-3: at least two layers)
    for nx, world, L in ((512, 8, 6), (256, 4, 5), (96, 3, 4)):
        ld = auto_dist_levels(nx, world, L)
        assert ld + 1 <= L and nx % (world * 2 ** (ld + 1)) == 0
