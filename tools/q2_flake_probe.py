"""Reproduces tests/test_gpu_distributed_q2.py's bridge case and prints the single-process compliance of every rank."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from helpers import BC_BRIDGE, MATERIAL, seeded_density
    from ndr_amd import pyVoxelFEM as pv
    from ndr_amd.distributed_q2 import DistributedMGSolverQ2
    ne, levels = (64, 8, 16), 3
    dom = ([0.0, 0.0, 0.0], [2.0, 1.0, 1.0])
    rho = torch.from_numpy(seeded_density(ne, 88)).cuda()
    if mode != "single-only":
        ds = DistributedMGSolverQ2(ne, dom[0], dom[1], BC_BRIDGE, MATERIAL, levels)
        ds.set_local_densities(rho.view(ne[0], -1)[ds.part.x0:ds.part.x1].reshape(-1).clone())
        f = ds.local_loads()
        u = ds.pcg(torch.zeros_like(f), f, 100, 1e-8, 1, 2, True)
        comp = 2.0 * ds.compliance(f, u)
    else:
        comp = 0.0
    if mode == "barrier":
        torch.cuda.synchronize(); dist.barrier()
    out = []
    for rep in range(2):
        t = pv.TensorProductSimulator([2, 2, 2], dom, list(ne))
        t.readMaterial(MATERIAL); t.applyDisplacementsAndLoadsFromFile(BC_BRIDGE); t.E_min = 1e-4
        t.setElementDensities(rho)
        mg = t.multigridSolver(levels)
        fg = t.buildLoadVector_device()
        h = []
        ug = mg.preconditionedConjugateGradient_device(torch.zeros_like(fg), fg, 100, 1e-8, None, 1, 2, True,
                                                       residual_cb=(lambda it, r: h.append(r)) if mode != "no-callback" else None)
        out.append((mg.last_iterations, float((fg * ug).sum())))
    print("mode %s rank %d distributed %.15g single %s" % (mode, rank, comp, out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    for mode in sys.argv[1:]:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        ctx = mp.get_context("spawn")
        ps = [ctx.Process(target=worker, args=(r, 4, port, mode)) for r in range(4)]
        [p.start() for p in ps]; [p.join() for p in ps]
