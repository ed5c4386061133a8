"""CG-MG rate against the number of multigrid levels (the coarsest level is solved with a dense inverse)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from bench import pcg_rate
for n, lvs in ((256, (4, 5, 6)), (512, (5, 6, 7))):
    for lv in lvs:
        try:
            r = pcg_rate((n, n, n), lv, ([0, 0, 0], [2, 1, 1]))
            print(n, lv, "iterations", r["iterations"], "%.2f it/s" % r["iterations_per_s"], "compliance %.8f" % r["compliance"], flush=True)
        except RuntimeError as e:
            print(n, lv, "error", e, flush=True)
