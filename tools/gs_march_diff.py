import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
e = np.abs(a - b).max(axis=3)
print("max abs diff", e.max(), "of", np.abs(b).max())
bad = np.argwhere(e > 1e-9)
print(len(bad), "nodes differ of", e.size)
for px in (0, 1):
    for py in (0, 1):
        for pz in (0, 1):
            sub = e[px::2, py::2, pz::2]
            print("parity", px, py, pz, "max", sub.max(), "count", int((sub > 1e-9).sum()), "of", sub.size)
if len(bad):
    print("x planes:", sorted(set(bad[:, 0]))[:40])
    print("y rows:", sorted(set(bad[:, 1]))[:60])
    print("z cols:", sorted(set(bad[:, 2]))[:80])
    print(bad[:10])
