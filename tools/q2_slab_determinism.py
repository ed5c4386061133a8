"""Determinism check of the degree-2 slab solve: the same solve twice per process set (with the allocator's free memory
scribbled over in between) must give bitwise equal residual histories; likewise the single-process solve."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist, torch.multiprocessing as mp


def scribble():
    junk = [torch.full((1 << 24,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    del junk


def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from helpers import BC_BRIDGE, MATERIAL, seeded_density
    from ndr_amd import pyVoxelFEM as pv
    from ndr_amd.distributed_q2 import DistributedMGSolverQ2
    ne, levels = (64, 8, 16), 3
    dom = ([0.0, 0.0, 0.0], [2.0, 1.0, 1.0])
    rho = torch.from_numpy(seeded_density(ne, 88)).cuda()
    scribble()
    ds = DistributedMGSolverQ2(ne, dom[0], dom[1], BC_BRIDGE, MATERIAL, levels)
    own = rho.view(ne[0], -1)[ds.part.x0:ds.part.x1].reshape(-1).clone()
    hists = []
    for rep in range(3):
        scribble()
        ds.set_local_densities(own)
        f = ds.local_loads()
        h = []
        ds.pcg(torch.zeros_like(f), f, 100, 1e-8, 1, 2, True, callback=lambda it, r: h.append(r))
        hists.append(h)
    if "--all-ranks" in sys.argv or rank == 0:
        print("distributed: iterations", [len(h) for h in hists], "bitwise equal:", hists[0] == hists[1] == hists[2], flush=True)
        t = pv.TensorProductSimulator([2, 2, 2], dom, list(ne))
        t.readMaterial(MATERIAL); t.applyDisplacementsAndLoadsFromFile(BC_BRIDGE); t.E_min = 1e-4
        t.setElementDensities(rho)
        mg = t.multigridSolver(levels)
        fg = t.buildLoadVector_device()
        hs = []
        for rep in range(3):
            scribble()
            h = []
            mg.preconditionedConjugateGradient_device(torch.zeros_like(fg), fg, 100, 1e-8, None, 1, 2, True, residual_cb=lambda it, r: h.append(r))
            hs.append(h)
        print("rank", rank, "single process: iterations", [len(h) for h in hs], "bitwise equal:", hs[0] == hs[1] == hs[2], flush=True)
        n = min(len(hists[0]), len(hs[0]))
        print("max rel diff distributed vs single:", max(abs(a - b) / b for a, b in zip(hists[0][:n], hs[0][:n])), flush=True)
        for k in (1, 2):
            d = [i for i, (a, b) in enumerate(zip(hists[0], hists[k])) if a != b]
            if d:
                print("distributed rep 0 vs %d first difference at iteration %d: %.17g %.17g" % (k, d[0] + 1, hists[0][d[0]], hists[k][d[0]]))
            d = [i for i, (a, b) in enumerate(zip(hs[0], hs[k])) if a != b]
            if d:
                print("single rep 0 vs %d first difference at iteration %d: %.17g %.17g" % (k, d[0] + 1, hs[0][d[0]], hs[k][d[0]]))
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 4, port)) for r in range(4)]
    [p.start() for p in ps]; [p.join() for p in ps]
