"""Which operation of the degree-2 path is not reproducible under contention?  WORLD processes share the GPU and each repeats
every operation REPS times on identical inputs; an operation whose outputs are not bitwise equal across repetitions is reported."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.multiprocessing as mp


def digest(t):
    return hashlib.md5(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]


def worker(rank, world, reps, q):
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
    fields = {l: (torch.randn((mg._nn(l), 3), dtype=torch.float64, device="cuda", generator=g),
                  torch.randn((mg._nn(l), 3), dtype=torch.float64, device="cuda", generator=g)) for l in range(levels + 1)}
    res = {}

    def rec(name, tensor):
        res.setdefault(name, set()).add(digest(tensor))

    f = t.buildLoadVector_device()
    for rep in range(reps):
        mg.updateElementStiffnessMatrices()
        for l in range(levels + 1):
            u, b = fields[l]
            rec("apply level %d" % l, mg.applyK_device(l, u))
            rec("residual level %d" % l, mg.computeResidual_device(l, u, b))
            if l < levels:
                rec("sweep forward level %d" % l, mg.smoothing_device(l, u, b, True))
                rec("sweep backward level %d" % l, mg.smoothing_device(l, u, b, False))
                rec("restrict from level %d" % l, mg.restriction_device(l, u))
                rec("prolong to level %d" % l, mg.interpolation_device(l, fields[l + 1][0]))
        x = torch.zeros_like(f)
        rec("one FMG cycle after a fresh operator update", mg.solve_device(x, f, 1, 2, True, True, None, True))
        for k in range(4):
            rec("FMG cycle, operators untouched", mg.solve_device(x, f, 1, 2, True, True, None, True))
            rec("V cycle, operators untouched", mg.solve_device(x, f, 1, 2, True, True, None, False))
        rec("pcg 10 iterations", mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, 10, 1e-30, None, 1, 2, True))
    bad = {k: len(v) for k, v in res.items() if len(v) > 1}
    q.put((rank, bad))


if __name__ == "__main__":
    world, reps = int(sys.argv[1]), int(sys.argv[2])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, reps, q)) for r in range(world)]
    [p.start() for p in ps]
    out = [q.get(timeout=900) for _ in ps]
    [p.join() for p in ps]
    for rank, bad in sorted(out):
        print("rank", rank, "operations with more than one distinct result:", bad or "none", flush=True)
