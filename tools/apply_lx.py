"""apply kernel with and without the line-exclusive tiling (VFEM_OPT_DMA_LX), event-timed: python tools/apply_lx.py [n]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench
from helpers import make_hip
from ndr_amd import _lib
lib = _lib.load()
for n in ([int(a) for a in sys.argv[1:]] or [512, 256]):
    t = make_hip((n, n, n), ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    ref = None
    for rep in range(2):
        for lx in (0, 2, 1):
            _lib.check(lib.vfem_sim_set_option(t._h, 11, lx))
            sec = bench.time_apply(t, u, 30, 10)
            out = t.applyK_device(u)
            if ref is None: ref = out.clone()
            ab = bench.algorithmic_bytes((n, n, n))
            print("n %d lx %d: %.4f ms  frac %.4f  equal %s" % (n, lx, sec * 1e3, ab / sec / 8e12, bool(torch.equal(out, ref))), flush=True)
    # residual variant through the multigrid handle
    del t, u, ref
    torch.cuda.empty_cache()
