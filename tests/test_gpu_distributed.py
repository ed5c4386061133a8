"""-m gpu: two ranks sharing the one GPU of the test box (gloo rendezvous, planes staged through the host)
run the slab-decomposed apply with the real HIP kernels; it must equal the single-rank result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ne, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from ndr_amd import distributed as vd
    part = vd.SlabPartition(ne, world, rank, align=2)
    ops = vd.HipLocalOps(part, [0, 0, 0], [2, 1, 1])
    ops.set_densities(vd.seeded_slab_density(part).cuda())
    u = vd.seeded_slab_field(part).cuda()
    uv = u.view(part.n_planes, -1)
    if part.gl:
        uv[0] = 1e30
    if part.gr:
        uv[-1] = -1e30
    K = vd.DistributedStiffness(part, ops)
    out = K.apply(u)
    nrm = float(K.halo.dot(out, out).item())
    full = vd.SlabPartition(ne, 1, 0)
    gops = vd.HipLocalOps(full, [0, 0, 0], [2, 1, 1])
    gops.set_densities(vd.seeded_slab_density(full).cuda())
    ref = gops.apply(vd.seeded_slab_field(full).cuda()).view(ne[0] + 1, -1)
    mine = out.view(part.n_planes, -1)[part.first_owned:part.last_owned + 1]
    want = ref[part.x0:part.x1 + 1]
    err = float((mine - want).abs().max() / want.abs().max())
    q.put((rank, err, nrm, float((ref * ref).sum())))
    dist.destroy_process_group()


def test_two_ranks_one_gpu_apply():
    ne, world = (24, 10, 70), 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, ne, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, nrm, ref in res:
        assert err < 1e-12, (rank, err)
        assert abs(nrm - ref) < 1e-10 * ref
