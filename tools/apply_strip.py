"""Apply kernel with and without the strip launch for the z remainder (VFEM_OPT_DMA_STRIP 0|1): timing and agreement."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_hip
from ndr_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
for ne in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(512, 512, 512), (256, 256, 256)]:
    tps = make_hip(ne, ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((tps.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty_like(u)
    res = {}
    for strip in (0, 1, 2, 0, 1, 2):
        set_knob(tps, 9, strip)
        for _ in range(30):
            lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), 0, _stream())
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), 0, _stream())
        b.record(); torch.cuda.synchronize()
        res[strip] = out.clone()
        print("%s strip=%d: %.4f ms" % (ne, strip, a.elapsed_time(b) / 20), flush=True)
    print("   max |difference| strip vs no strip: %.3e (of %.3e)" % (float((res[0] - res[1]).abs().max()), float(res[0].abs().max())), flush=True)
    set_knob(tps, 9, 1)
    del tps, u, out, res
    torch.cuda.empty_cache()
