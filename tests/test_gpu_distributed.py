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


def _worker(rank, world, port, ne, q, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # gloo: both ranks share device 0 (planes staged through the host); nccl (= RCCL): one device per rank, planes go GPU to GPU
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
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


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL transport needs one device per rank (the test box has one)")
def test_two_ranks_two_gpus_rccl_apply():
    """the same slab apply over the nccl backend: HaloExchanger.start / finish with device plane views, overlapped with the
    interior planes -- runs wherever at least two devices are visible (the driver's multi-GPU node)"""
    ne, world = (24, 10, 70), 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = __import__('helpers').free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ne, q, "nccl")) for r in range(world)]
    for p in procs:
        p.start()
    res = __import__('helpers').collect_from_ranks(q, procs)
    for rank, err, nrm, ref in res:
        assert err < 1e-12, (rank, err)
        assert abs(nrm - ref) < 1e-10 * ref


def test_two_ranks_one_gpu_apply():
    ne, world = (24, 10, 70), 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = __import__('helpers').free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ne, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = __import__('helpers').collect_from_ranks(q, procs)
    for rank, err, nrm, ref in res:
        assert err < 1e-12, (rank, err)
        assert abs(nrm - ref) < 1e-10 * ref


def _pcg_worker(rank, world, port, ne, levels, q, sharded=False, march=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from helpers import BC_CANTILEVER, MATERIAL, make_hip, seeded_density
    from ndr_amd import distributed as vd
    dom = ([0.0, 0.0, 0.0], [2.0, 1.0, 1.0])
    rho = torch.from_numpy(seeded_density(ne, 88)).cuda()
    ds = vd.DistributedMGSolver(ne, dom[0], dom[1], BC_CANTILEVER, MATERIAL, levels)
    if march is not None:      # (overlap, marching): finest-level sweeps by the marching kernel on the slabs, interface planes first
        from ndr_amd import _lib
        _lib.check(ds.lib.vfem_sim_set_option(ds.lsim._h, 19, 2))
        ds.overlap_sweeps = bool(march)
    if sharded:       # the rank hands over its OWNED layers only; ghosts come from the neighbours, coarse operators from an all-gather
        ds.set_local_densities(rho.view(ne[0], -1)[ds.part.x0:ds.part.x1].reshape(-1).clone())
    else:
        ds.set_global_densities(rho)
    f = ds.local_loads()
    u = ds.pcg(torch.zeros_like(f), f, 100, 1e-8, 1, 2, True)
    comp = 2.0 * ds.compliance(f, u)
    # single-process reference with the same kernels
    t = make_hip(ne, dom, BC_CANTILEVER, rho.cpu().numpy())
    mg = t.multigridSolver(levels)
    fg = t.buildLoadVector_device()
    ug = mg.preconditionedConjugateGradient_device(torch.zeros_like(fg), fg, 100, 1e-8, None, 1, 2, True)
    cg = float((fg * ug).sum())
    g = ds.geom[0]
    mine = u.view(g.n_planes, -1)[g.first_owned:g.last_owned + 1]
    want = ug.view(ne[0] + 1, -1)[ds.part.x0:ds.part.x1 + 1]
    err = float((mine - want).abs().max() / want.abs().max())
    # sensitivity of the owned element layers against the single-process sensitivity
    first, count = ds.owned_element_range()
    gd = ds.compliance_gradient(u)
    gs = t.complianceGradient_device(ug)[first:first + count]
    gerr = float((gd - gs).abs().max() / gs.abs().max())
    err = max(err, gerr)
    if march is not None:
        q.put((rank, ds.Ld, ds.last_iterations, mg.last_iterations, comp, cg, err, [getattr(h, "messages", 0) for h in ds.halos],
               u.view(g.n_planes, -1)[g.first_owned:g.last_owned + 1].cpu().numpy()))
    else:
        q.put((rank, ds.Ld, ds.last_iterations, mg.last_iterations, comp, cg, err))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ne,levels", [(2, (32, 16, 64), 3), (3, (48, 16, 16), 3)])
def test_distributed_sweeps_overlapped_with_the_halo_exchange(world, ne, levels):
    """finest-level sweeps by the marching kernel on the slabs: the planes a neighbour waits for are relaxed first and travel while
    the interior planes are relaxed (vfem_mg_smooth_group_planes) -- same iterates as the blocking order and as the single process;
    and a colour group that did not change the planes the neighbours mirror is followed by no exchange at all"""
    ctx = mp.get_context("spawn")
    out = {}
    for overlap in (1, 0):
        q = ctx.Queue()
        port = __import__('helpers').free_port()
        procs = [ctx.Process(target=_pcg_worker, args=(r, world, port, ne, levels, q, False, overlap)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(__import__('helpers').collect_from_ranks(q, procs), key=lambda r: r[0])
        out[overlap] = res
        for rank, Ld, it_d, it_s, comp, cg, err, msgs, u in res:
            assert Ld >= 1 and it_d == it_s, (it_d, it_s)
            assert abs(comp - cg) < 1e-9 * abs(cg) and err < 1e-7, (comp, cg, err)
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a[8], b[8])                            # overlapped order = blocking order, bit for bit
        assert a[7] == b[7]                                          # and the same messages
    # finest level, per PCG iteration: K d (1), FMG: right-hand side (1), prolongated start (1), 2 + 2 sweeps (4: ONE exchange per sweep,
    # after the group that relaxes the odd planes -- two per sweep before), residual (1), prolongated correction (1) = 9 messages per
    # neighbour (13 with an exchange after every colour group)
    it_d, msgs0 = out[1][0][2], out[1][0][7][0]
    assert msgs0 <= 9 * it_d + 4, (msgs0, it_d)


@pytest.mark.parametrize("world,ne,levels", [(2, (32, 16, 16), 3), (2, (48, 8, 16), 2),
                                             (4, (64, 16, 16), 3), (4, (64, 16, 16), 4)])
def test_distributed_pcg_matches_single_process(world, ne, levels):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = __import__('helpers').free_port()
    procs = [ctx.Process(target=_pcg_worker, args=(r, world, port, ne, levels, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = __import__('helpers').collect_from_ranks(q, procs)
    for rank, Ld, it_d, it_s, comp, cg, err in res:
        assert Ld >= 1
        assert it_d == it_s, (it_d, it_s)
        assert abs(comp - cg) < 1e-9 * abs(cg), (comp, cg)
        assert err < 1e-7, err


def _driver_worker(rank, world, port, ne, levels, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from helpers import BC_CANTILEVER, MATERIAL, seeded_density
    from ndr_amd import distributed as vd
    dom = ([0.0, 0.0, 0.0], [2.0, 1.0, 1.0])
    rho = torch.from_numpy(seeded_density(ne, 88)).cuda()
    ds = vd.DistributedMGSolver(ne, dom[0], dom[1], BC_CANTILEVER, MATERIAL, levels)
    ds.set_global_densities(rho)
    f = ds.local_loads()
    out = {}
    for params in ((1, 2, True, True), (2, 1, False, True), (1, 1, True, False)):      # (mgIterations, smoothing steps, FMG, symmetric GS)
        ds.symmetric_gs = params[3]
        res = []
        for c_driver in (True, False):
            ds.use_c_driver = c_driver
            hist = []
            u = ds.pcg(torch.zeros_like(f), f, 60, 1e-8, params[0], params[1], params[2], callback=lambda it, rn: hist.append(rn))
            res.append((ds.last_iterations, hist, u.clone()))
        (it_c, h_c, u_c), (it_p, h_p, u_p) = res
        out[params] = (it_c, it_p, max(abs(a - b) / b for a, b in zip(h_c, h_p)) if h_p else 0.0, float((u_c - u_p).abs().max() / u_p.abs().max()))
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ne,levels", [(2, (32, 16, 16), 3), (3, (48, 16, 16), 3)])
def test_c_driver_of_the_slab_solve_equals_the_python_driver(world, ne, levels):
    """vfem_mg_pcg_slab (the rank's whole solve in one library call, halo / all-reduce by callbacks) against the same algorithm driven
    call by call from Python: same iteration counts, residual histories and displacements to rounding (the two differ only in the
    summation order of the dot products), for the parameterisations of tests/test_gpu_parity.py"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = __import__('helpers').free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, ne, levels, q)) for r in range(world)]
    for p in procs:
        p.start()
    for rank, out in __import__('helpers').collect_from_ranks(q, procs):
        for params, (it_c, it_p, herr, uerr) in out.items():
            assert it_c == it_p and it_c > 0, (params, it_c, it_p)
            assert herr < 1e-6 and uerr < 1e-9, (params, herr, uerr)


@pytest.mark.parametrize("world,ne,levels", [(2, (32, 16, 16), 3), (4, (64, 16, 16), 4), (3, (48, 16, 16), 3)])
def test_distributed_pcg_with_sharded_densities(world, ne, levels):
    """no rank holds the whole density field (set_local_densities): same iterations, compliance and displacements as the
    single-process solve"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = __import__('helpers').free_port()
    procs = [ctx.Process(target=_pcg_worker, args=(r, world, port, ne, levels, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = __import__('helpers').collect_from_ranks(q, procs)
    for rank, Ld, it_d, it_s, comp, cg, err in res:
        assert Ld >= 1
        assert it_d == it_s, (it_d, it_s)
        assert abs(comp - cg) < 1e-9 * abs(cg), (comp, cg)
        assert err < 1e-7, err


def _mlp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from ndr_amd.mlp import TrainableMLP
    side = (24, 10, 12)
    plane = side[1] * side[2]
    torch.manual_seed(5)                                   # same initial weights and B on every rank
    net = TrainableMLP(3, 1, 64, 4, 64, 2.0)
    x0, x1 = rank * side[0] // world, (rank + 1) * side[0] // world
    tgt = torch.linspace(0, 1, side[0] * plane, device="cuda")
    # sharded: every rank differentiates through its planes only, gradients are summed over the group
    net.set_grid(side, voxel_range=(x0 * plane, (x1 - x0) * plane))
    net.zero_grad()
    rho = net.forward_grid()
    loss = ((rho - tgt[x0 * plane:x1 * plane]) ** 2).sum()
    loss.backward()
    sharded = [p.grad.clone() for p in net.parameters()]
    # the same step on the whole grid in this process
    net.set_grid(side)
    net.zero_grad()
    rho_full = net.forward_grid()
    ((rho_full - tgt) ** 2).sum().backward()
    full = [p.grad.clone() for p in net.parameters()]
    err = max(float((a - b).norm() / b.norm()) for a, b in zip(sharded, full))
    same = bool(torch.equal(rho.detach(), rho_full.detach()[x0 * plane:x1 * plane]))
    q.put((rank, err, same))
    dist.destroy_process_group()


def test_mlp_training_step_sharded_over_ranks_equals_whole_grid():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = __import__('helpers').free_port()
    procs = [ctx.Process(target=_mlp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = __import__('helpers').collect_from_ranks(q, procs)
    for rank, err, same in res:
        assert same, rank
        assert err < 5e-3, (rank, err)       # fp16 operands: chunk boundaries differ between the two evaluations


def _trainer_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from helpers import BC_CANTILEVER, MATERIAL
    from ndr_amd import distributed as vd, fem, pyVoxelFEM
    from ndr_amd.mlp import TrainableMLP
    grid, dom, v0, levels = (32, 16, 16), ([0.0, 0.0, 0.0], [2.0, 1.0, 1.0]), 0.5, 2
    torch.manual_seed(7)
    net = TrainableMLP(3, 1, 64, 4, 64, 1.5)
    with torch.no_grad():                                     # a non-trivial field: scale the output layer up
        net._linears()[-1].weight.mul_(3.0)
    # sharded evaluation of the closure
    ds = vd.DistributedMGSolver(grid, dom[0], dom[1], BC_CANTILEVER, MATERIAL, levels)
    tr = vd.DistributedDensityTrainer(ds, net, v0, tol=1e-9, zero_init=True)
    net.zero_grad()
    loss_d = tr.loss()
    loss_d.backward()
    gd = [p.grad.clone() for p in net.parameters()]
    vol = float(tr._reduce(tr.last_density.sum().reshape(1), dist.ReduceOp.SUM)[0]) / (grid[0] * grid[1] * grid[2])
    # the same closure in this process alone (fem.satisfy_volume_constraint + VoxelFEMFunction)
    tps = fem.initializeTensorProductSimulator([1, 1, 1], [np.array(dom[0]), np.array(dom[1])], list(grid), v0, 1, 1e-4, 3, MATERIAL, BC_CANTILEVER)
    obj = pyVoxelFEM.MultigridComplianceObjective(tps.multigridSolver(levels))
    obj.tol, obj.mgIterations, obj.fullMultigrid, obj.zeroInit, obj.mgSmoothingIterations = 1e-9, 1, True, True, 2
    top = pyVoxelFEM.TopologyOptimizationProblem(tps, obj, [pyVoxelFEM.TotalVolumeConstraint(v0)], [])
    net.set_grid(grid)
    net.zero_grad()
    density = fem.satisfy_volume_constraint(net.forward_grid().view(grid), torch.tensor(v0, device="cuda"), mode="constrained_sigmoid")
    loss_s = fem.VoxelFEMFunction.apply(density.flatten(), top)
    loss_s.backward()
    gs = [p.grad.clone() for p in net.parameters()]
    # one norm over all parameters: the output bias has a vanishing gradient under the mean constraint (a constant added to
    # every logit is absorbed by the shift), so its own relative error is noise over noise
    fa, fb = torch.cat([g.reshape(-1) for g in gd]), torch.cat([g.reshape(-1) for g in gs])
    gerr = float((fa - fb).norm() / fb.norm())
    q.put((rank, float(loss_d.detach()), float(loss_s.detach()), vol, gerr))
    dist.destroy_process_group()


def test_distributed_closure_equals_single_process_closure():
    """train_xdg closure over 2 slab ranks: loss (2 J), volume and parameter gradients equal the single-process evaluation"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = __import__('helpers').free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = __import__('helpers').collect_from_ranks(q, procs)
    for rank, ld, ls, vol, gerr in res:
        assert abs(vol - 0.5) < 1e-6, vol
        assert abs(ld - ls) < 2e-5 * abs(ls), (ld, ls)          # float32 loss value of the autograd node
        assert gerr < 1e-2, gerr
