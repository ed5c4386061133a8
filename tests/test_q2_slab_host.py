"""Host logic of the degree-2 slab decomposition (ndr_amd/distributed_q2.py), no GPU: geometry of the local grids and the
slab-local evaluation of boundary conditions / coarsened Dirichlet masks against a direct restatement of the reference rules
on the whole grid (applyDisplacementsAndLoads, TPS.hh:358-409; coarsened masks, MG.hh:57-84 with degree 2)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import BC_BRIDGE, BC_CANTILEVER  # noqa: E402

P = 2


def _global_masks_and_loads(bbmin, bbmax, ne, bc_path, T):
    from oracle.vfem_oracle import parse_bc_file
    ne = np.asarray(ne)
    nn = P * ne + 1
    size = bbmax - bbmin
    pos = [bbmin[d] + np.arange(nn[d]) * (size[d] / (nn[d] - 1.0)) for d in range(3)]
    mask = np.zeros(tuple(nn), dtype=np.uint8)
    loads = np.zeros(tuple(nn) + (3,))
    for kind, comp, val, lo, hi, rel in parse_bc_file(bc_path, 3):
        if rel:
            lo, hi = bbmin + lo * size, bbmin + hi * size
        inside = [(pos[d] >= lo[d]) & (pos[d] <= hi[d]) for d in range(3)]
        box = inside[0][:, None, None] & inside[1][None, :, None] & inside[2][None, None, :]
        if kind == "force":
            loads[box] = val / box.sum()
        else:
            mask[box] |= np.uint8(sum(1 << c for c in range(3) if comp[c]))
    masks = [mask]
    for l in range(T):
        f = masks[-1]
        nec = ne >> (l + 1)
        c = np.zeros(tuple(P * nec + 1), dtype=np.uint8)
        for g in np.argwhere(f != 0):                            # MG.hh:57-84, node by node
            rng = []
            for d in range(3):
                e = min(g[d] // (2 * P), nec[d] - 1)
                t = g[d] - 2 * P * e
                rng.append((P * e, P * e) if t == 0 else ((P * e + P, P * e + P) if t == 2 * P else (P * e, P * e + P)))
            c[rng[0][0]:rng[0][1] + 1, rng[1][0]:rng[1][1] + 1, rng[2][0]:rng[2][1] + 1] |= f[tuple(g)]
        masks.append(c)
    return masks, loads


@pytest.mark.parametrize("bc", [BC_CANTILEVER, BC_BRIDGE])
@pytest.mark.parametrize("world,ne,Ld", [(2, (32, 4, 8), 2), (4, (32, 8, 4), 1), (3, (24, 4, 4), 1), (1, (16, 4, 4), 1)])
def test_slab_masks_and_loads_equal_the_global_rules(bc, world, ne, Ld):
    from ndr_amd.distributed import SlabPartition
    from ndr_amd.distributed_q2 import G, _LevelGeomQ2, slab_masks_and_loads
    bbmin, bbmax = np.array([0.0, 0.0, 0.0]), np.array([2.0, 1.0, 1.0])
    T = Ld + 1
    gmasks, gloads = _global_masks_and_loads(bbmin, bbmax, ne, bc, T)
    assembled = []
    for rank in range(world):
        part = SlabPartition(ne, world, rank, align=2 ** (Ld + 1))
        geom = [_LevelGeomQ2(part, l, Ld, ne) for l in range(T + 1)]
        masks, loads, maskT = slab_masks_and_loads(bbmin, bbmax, ne, bc, geom, T, torch.device("cpu"))
        for l, g in enumerate(geom):
            # geometry: local grid = owned + G ghost element layers per neighbour, 2 planes per layer, even global start
            assert g.nx == (part.x1 - part.x0) // 2 ** l + G * ((rank > 0) + (rank < world - 1))
            assert g.xoffe % 2 == 0 and g.xoffn == P * g.xoffe and g.halo_width == P * G
            mine = masks[l].reshape(g.n_planes, -1)
            want = gmasks[l].reshape(gmasks[l].shape[0], -1)[g.xoffn:g.xoffn + g.n_planes]
            # exact on every plane the rank computes on (owned + interface); ghost planes of levels >= 1 may lack flags
            assert np.array_equal(mine[g.first_owned:g.last_owned + 1], want[g.first_owned:g.last_owned + 1]), (rank, l)
            if l == 0:
                assert np.array_equal(mine, want)
        g0 = geom[0]
        want = gloads.reshape(gloads.shape[0], -1)[g0.xoffn:g0.xoffn + g0.n_planes]
        assert np.array_equal(loads.numpy().reshape(g0.n_planes, -1), want)
        gT = geom[T]
        lo, hi = gT.reduction_weight_planes()
        assembled.append(maskT.reshape(gT.n_planes, -1)[lo:hi])
        if rank > 0:        # stored element layers halve exactly from level to level (vfem_gmg_create_slab's requirement)
            for l in range(1, Ld + 1):
                f, c = geom[l - 1], geom[l]
                assert f.nx + f.extra_lo + f.extra_hi == 2 * (c.nx + c.extra_lo + c.extra_hi)
                assert f.xoffe - f.extra_lo == 2 * (c.xoffe - c.extra_lo)
    whole = np.concatenate(assembled, axis=0)
    assert np.array_equal(whole, gmasks[T].reshape(gmasks[T].shape[0], -1))


def test_internal_dirichlet_nodes_are_refused(tmp_path):
    """MG.hh:74-76: a constrained node strictly inside a coarse element has no coarse counterpart"""
    import json
    from ndr_amd.distributed import SlabPartition
    from ndr_amd.distributed_q2 import _LevelGeomQ2, slab_masks_and_loads
    bc = tmp_path / "inner.bc"
    bc.write_text(json.dumps({"regions": [
        {"type": "dirichlet", "value": [0, 0, 0], "box%": {"minCorner": [0.3, 0.3, 0.3], "maxCorner": [0.36, 0.4, 0.4]}},
        {"type": "force", "value": [0, -1, 0], "box%": {"minCorner": [0.99, -0.1, -0.1], "maxCorner": [1.1, 1.1, 1.1]}}]}))
    ne = (16, 4, 4)
    part = SlabPartition(ne, 1, 0, align=4)
    geom = [_LevelGeomQ2(part, l, 1, ne) for l in range(3)]
    with pytest.raises(RuntimeError, match="internal nodes"):
        slab_masks_and_loads(np.zeros(3), np.ones(3), ne, str(bc), geom, 2, torch.device("cpu"))
