"""CPU check of the merged evaluation of a level-1 node row (kernels_l1_merged.hip):
   sum over the 8 incident coarse elements x 8 children  ==  sum over the 8 mirror classes g of
   (moduli combined over the elements that share a neighbour) x (signed entries of cK0[0]).
   Run: python tools/l1_merged_check.py"""
import itertools
import numpy as np

rng = np.random.default_rng(3)
K = rng.standard_normal((24, 24)); K = K + K.T                      # stands for cK0[0]
bit = lambda v, a: (v >> (2 - a)) & 1                                # axis a (0 = x) is bit 2 - a
sg = lambda v, a: -1.0 if bit(v, a) else 1.0

def cK(f, n, a, m, b):                                               # the mirror images (kernels_mg.hip:510)
    return sg(f, a) * sg(f, b) * K[3 * (n ^ f) + a, 3 * (m ^ f) + b]

Ef = rng.random((4, 4, 4))                                           # fine moduli around the node: index p = fine offset + 2
u = rng.standard_normal((3, 3, 3, 3))                                # neighbours o + 1, component

# direct: elements d in {0,1}^3 (element index = node - 1 + d), children f
S = np.zeros(3); M = np.zeros((3, 3))
for d in itertools.product((0, 1), repeat=3):
    li = sum((1 - d[t]) << (2 - t) for t in range(3))
    for f in range(8):
        p = tuple(2 * d[t] + bit(f, t) for t in range(3))           # fine element 2(i-1+d)+f  ->  offset -2+2d+f -> p
        E = Ef[p]
        for m in range(8):
            o = tuple(d[t] - 1 + bit(m, t) + 1 for t in range(3))   # neighbour index in u (0..2)
            for a in range(3):
                for b in range(3):
                    c = E * cK(f, li, a, m, b)
                    if m == li: M[a, b] += c
                    else: S[a] += c * u[o][b]

# merged
off = lambda dd, gg: gg if dd else -1 - gg                           # fine offset of the element of side dd in mirror class gg
S2 = np.zeros(3); M2 = np.zeros((3, 3))
for g in range(8):
    A = np.zeros((2, 2, 2))
    for d in itertools.product((0, 1), repeat=3):
        A[d] = Ef[tuple(off(d[t], bit(g, t)) + 2 for t in range(3))]
    for w in range(8):
        C = np.array([[sg(g, a) * sg(g, b) * K[3 * g + a, 3 * (g ^ w) + b] for b in range(3)] for a in range(3)])
        T = np.zeros((3, 3))
        for o in itertools.product((-1, 0, 1), repeat=3):
            if sum((1 if o[t] else 0) << (2 - t) for t in range(3)) != w: continue
            for a in range(3):
                for b in range(3):
                    W = 0.0
                    for d in itertools.product((0, 1), repeat=3):
                        if any(o[t] != 0 and o[t] != 2 * d[t] - 1 for t in range(3)): continue
                        s = 1.0 if a == b else (2 * d[a] - 1) * (2 * d[b] - 1)
                        W += s * A[d]
                    if w == 0: T[a, b] += W
                    else: T[a, b] += W * u[o[0] + 1, o[1] + 1, o[2] + 1, b]
        if w == 0: M2 += C * T
        else: S2 += (C * T).sum(axis=1)
print("S", np.abs(S - S2).max(), "M", np.abs(M - M2).max())
assert np.abs(S - S2).max() < 1e-11 and np.abs(M - M2).max() < 1e-11
print("ok")
