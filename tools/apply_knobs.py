import sys, time
import numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import make_hip
from ndr_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import set_knob
lib = _lib.load()
n = 512
t = make_hip((n, n, n), ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device='cuda'))
u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device='cuda')
ref = t.applyK_device(u, 1)
for rnd in range(2):
    for skel in (0, 1):
        for pd in (0,):
            set_knob(t, 4, skel); set_knob(None, 3, 0)
            if skel == 0 and rnd == 0:
                o = t.applyK_device(u); print("store mode", pd, "rel err", float((o - ref).abs().max() / ref.abs().max()))
            t.applyK_device(u); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): t.applyK_device(u)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print(f"round {rnd} impl(0=dma,1=reg)={skel} store={pd}: {dt*1e3:.3f} ms  {t.numElements()/dt/1e9:.1f} GVoxel/s", flush=True)
