"""Is the dense coarsest inverse (rocSOLVER potrf / potri inside the operator update) reproducible when several processes run
the update at the same moment?  WORLD processes, REPS synchronised updates each; prints the number of distinct results of
x = Ainv b per process, and of the Galerkin matrices' action on the coarsest level."""
import ctypes, os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.multiprocessing as mp


def worker(rank, world, reps, bar, q):
    torch.cuda.set_device(0)
    from helpers import BC_BRIDGE, MATERIAL, seeded_density
    from ndr_amd import _lib, pyVoxelFEM as pv
    lib = _lib.load()
    ne, levels = (64, 8, 16), 3
    dom = ([0.0, 0.0, 0.0], [2.0, 1.0, 1.0])
    rho = torch.from_numpy(seeded_density(ne, 88)).cuda()
    t = pv.TensorProductSimulator([2, 2, 2], dom, list(ne))
    t.readMaterial(MATERIAL); t.applyDisplacementsAndLoadsFromFile(BC_BRIDGE); t.E_min = 1e-4
    t.setElementDensities(rho)
    mg = t.multigridSolver(levels)
    g = torch.Generator(device="cuda").manual_seed(5)
    nL = mg._nn(levels)
    b = torch.randn((nL, 3), dtype=torch.float64, device="cuda", generator=g)
    inv, ku = set(), set()
    dg = lambda x: hashlib.md5(x.cpu().numpy().tobytes()).hexdigest()[:10]
    for rep in range(reps):
        bar.wait()
        mg.updateElementStiffnessMatrices()
        x = torch.zeros_like(b)
        _lib.check(lib.vfem_gmg_cycle_from_level(mg._h, levels, pv._ptr(x), pv._ptr(b), 2, 1, pv._stream()))
        inv.add(dg(x))
        ku.add(dg(mg.applyK_device(levels, b)))
    q.put((rank, len(inv), len(ku)))


if __name__ == "__main__":
    world, reps = int(sys.argv[1]), int(sys.argv[2])
    ctx = mp.get_context("spawn")
    q, bar = ctx.Queue(), ctx.Barrier(world)
    ps = [ctx.Process(target=worker, args=(r, world, reps, bar, q)) for r in range(world)]
    [p.start() for p in ps]
    out = [q.get(timeout=600) for _ in ps]
    [p.join() for p in ps]
    for rank, ninv, nku in sorted(out):
        print("rank %d: distinct Ainv b: %d, distinct K_coarsest b: %d (of %d synchronised updates)" % (rank, ninv, nku, reps), flush=True)
