"""Where a step of the marching level-0 sweep spends its cycles: s_memtime stamps of one workgroup (7 compute waves, 8 steps).
   python3 tools/gs_march_stamps.py [n=256]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_CANTILEVER, make_hip
from ndr_amd import _lib
from ndr_amd.pyVoxelFEM import _ptr, _stream
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
form = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NW = 4 if form == 2 else 7
tps = make_hip((n, n, n), ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
g = torch.Generator(device="cuda").manual_seed(88)
tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
_lib.check(lib.vfem_sim_set_option(tps._h, 19, 2))
_lib.check(lib.vfem_sim_set_option(tps._h, 23, form))
mg = tps.multigridSolver(0)
nn = mg._nn(0)
u = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
b = torch.randn((nn, 3), dtype=torch.float64, device="cuda", generator=g)
for rep in range(2):
    _lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(u), _ptr(b), 1, 2, _stream()))
torch.cuda.synchronize()
st = torch.zeros(7 * 8 * 16, dtype=torch.int64, device="cuda")
lib.vfem_debug_gsm_stamps.argtypes = [ctypes.c_void_p]
lib.vfem_debug_gsm_stamps(ctypes.c_void_p(st.data_ptr()))
_lib.check(lib.vfem_mg_smooth_sweeps(mg._h, 0, _ptr(u), _ptr(b), 1, 1, _stream()))
torch.cuda.synchronize()
lib.vfem_debug_gsm_stamps(None)
t = st.cpu().view(7, 8, 16)[:NW]
names = ["arrive B0", "leave B0", "end colour 0", "end colour 1", "row stored", "leave B1", "end colour 2", "end colour 3 + store"]
t0 = int(t[:, 2, 0].min())
print("time in units of 10 ns (s_memrealtime, 100 MHz) relative to the first arrival at B0 of step 2; rows = waves")
for m in (2, 3, 4):
    print("step", m)
    for w in range(NW):
        print("  wave %d: " % w + "  ".join("%s %6d" % (names[k][:12], int(t[w, m, k]) - t0) for k in range(8)))
d = t[:, 2:7, :].double()
print("mean durations over waves 0-5, steps 2-6 (10 ns): wait B0 %.0f | colour 0 %.0f | colour 1 %.0f | store %.0f | wait B1 %.0f | colour 2 %.0f | colour 3 + store %.0f | step %.0f" % (
    float((d[:NW - 1, :, 1] - d[:NW - 1, :, 0]).mean()), float((d[:NW - 1, :, 2] - d[:NW - 1, :, 1]).mean()), float((d[:NW - 1, :, 3] - d[:NW - 1, :, 2]).mean()),
    float((d[:NW - 1, :, 4] - d[:NW - 1, :, 3]).mean()), float((d[:NW - 1, :, 5] - d[:NW - 1, :, 4]).mean()), float((d[:NW - 1, :, 6] - d[:NW - 1, :, 5]).mean()),
    float((d[:NW - 1, :, 7] - d[:NW - 1, :, 6]).mean()), float((t[:NW - 1, 3:7, 0] - t[:NW - 1, 2:6, 0]).double().mean())))

e = t[:NW - 1, 2:7, :].double()
print("inside colour 2 (10 ns): reads + moduli landed %.0f | far-plane multiply-adds %.0f | own-plane multiply-adds %.0f | sums + shuffle %.0f | solve + write %.0f" % (
    float((e[:, :, 8] - e[:, :, 5]).mean()), float((e[:, :, 9] - e[:, :, 8]).mean()), float((e[:, :, 10] - e[:, :, 9]).mean()),
    float((e[:, :, 11] - e[:, :, 10]).mean()), float((e[:, :, 6] - e[:, :, 11]).mean())))
for w in range(0):
    print("  wave %d step 3: leave B1 %d  reads %d  far %d  mid %d  shuffle %d  end %d" % ((w,) + tuple(int(t[w, 3, q]) - int(t[w, 3, 5]) for q in (5, 8, 9, 10, 11, 6))))
