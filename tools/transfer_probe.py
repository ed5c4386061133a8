"""Times of the level-0 <-> level-1 grid transfers (k_restrict, k_prolong) through the C ABI:  python tools/transfer_probe.py [n ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
for n in [int(a) for a in sys.argv[1:]] or [256, 512]:
    tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    mg = tps.multigridSolver(3)
    g = torch.Generator(device="cuda").manual_seed(1)
    fine = torch.randn((mg._nn(0), 3), dtype=torch.float64, device="cuda", generator=g)
    coarse = torch.randn((mg._nn(1), 3), dtype=torch.float64, device="cuda", generator=g)
    out = torch.zeros_like(fine)

    def timed(fn, reps=6):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    gb_f, gb_c = fine.numel() * 8 / 1e9, coarse.numel() * 8 / 1e9
    tr = timed(lambda: mg.restriction_device(0, fine))
    tp = timed(lambda: mg.interpolation_device(0, coarse))
    ta = timed(lambda: mg.interpolation_device(0, coarse, out))
    print("n=%d restrict %.3f ms (%.2f TB/s)  prolong %.3f ms (%.2f TB/s)  prolong+= %.3f ms (%.2f TB/s)" % (
        n, tr, (gb_f + gb_c) / tr, tp, (gb_f + gb_c) / tp, ta, (2 * gb_f + gb_c) / ta), flush=True)
    del mg, tps, fine, coarse, out
    torch.cuda.empty_cache()
