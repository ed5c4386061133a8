"""Shared builders for the parity tests: the same problem set up on the CPU oracle and on the HIP path."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MATERIAL = os.path.join(ROOT, "VoxelFEM", "examples", "materials", "B9Creator.material")      # the reference's input data, at the reference's paths
BC_CANTILEVER = os.path.join(ROOT, "bcs", "3d", "cantilever_flexion.bc")
BC_BRIDGE = os.path.join(ROOT, "bcs", "3d", "bridge.bc")


def seeded_density(ne, seed=88, kind="uniform"):
    """D1 (i.i.d. U[0,1]) and D2 ("mid-optimisation" proxy) densities of SURVEY 8(d)."""
    rng = np.random.default_rng(seed)
    ne = tuple(int(n) for n in ne)
    if kind == "uniform":
        return rng.uniform(0.0, 1.0, size=int(np.prod(ne)))
    idx = np.stack(np.meshgrid(*[np.linspace(0, 1, n) for n in ne], indexing="ij"), -1)
    smooth = np.zeros(ne)
    for _ in range(6):
        k = rng.uniform(0.5, 3.0, size=len(ne)) * np.pi
        ph = rng.uniform(0, 2 * np.pi)
        smooth += np.cos(idx @ k + ph)
    lo, hi = smooth.min(), smooth.max()
    for _ in range(60):
        tau = 0.5 * (lo + hi)
        rho = 1.0 / (1.0 + np.exp(-8.0 * (smooth - tau)))
        if rho.mean() > 0.4:
            lo = tau
        else:
            hi = tau
    return rho.reshape(-1)


def make_oracle(ne, domain, bc, rho=None, v0=0.5, Emin=1e-4):
    from oracle import vfem_oracle as vo
    sim = vo.OracleSim(domain, ne)
    sim.read_material(MATERIAL)
    sim.set_uniform_densities(v0)
    if bc:
        sim.apply_bc_file(bc)
    sim.E0, sim.Emin, sim.gamma = 1.0, Emin, 3.0
    if rho is not None:
        sim.set_densities(rho)
    return sim


def make_hip(ne, domain, bc, rho=None, v0=0.5, Emin=1e-4):
    from ndr_amd import pyVoxelFEM as pv
    tps = pv.TensorProductSimulator([1, 1, 1], [np.array(domain[0], float), np.array(domain[1], float)], list(ne))
    tps.readMaterial(MATERIAL)
    tps.setUniformDensities(v0)
    if bc:
        tps.applyDisplacementsAndLoadsFromFile(bc)
    tps.E_0, tps.E_min, tps.gamma = 1.0, Emin, 3.0
    if rho is not None:
        tps.setElementDensities(rho)
    return tps


def record_deltas(name, payload):
    """achieved errors of the GPU parity runs, merged into gpurun_out/parity_deltas.json (the summaries judged are copied
    to profiles/); never fails a test"""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "parity_deltas.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[name] = payload
        with open(path, "w") as fh:
            json.dump(cur, fh, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def seeded_mlp_weights(es, nn_, nl, sigma, seed):
    """The build's own initialiser for full-size MLP checks: numpy PCG64 streams, N(0, 1/fan_in) weights,
    N(0, 0.1) biases, B ~ N(0,1) * sigma.  The full-size fixture (tests/golden/mlpfull_*.npz) stores only the reference's
    outputs for these weights plus a checksum of them, so 6.3 MB of weights stay out of git."""
    rng = np.random.default_rng(seed)
    B = (rng.standard_normal((es, 3)) * sigma).astype(np.float32)
    dims = [2 * es] + [nn_] * (nl - 1) + [1]
    Ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(nl)]
    bs = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(nl)]
    return B, Ws, bs


def mlp_weight_checksum(B, Ws, bs):
    return float(np.sum([np.abs(w.astype(np.float64)).sum() for w in Ws + bs + [B]]))


def free_port():
    """a TCP port nobody listens on right now (rendezvous of the multi-process tests): asking the kernel beats arithmetic on
    the pid, which collides with a lingering socket of an earlier run now and then"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def collect_from_ranks(q, procs, timeout=400):
    """results of the rank processes of a multi-process test; a rank that dies (its traceback is on stderr) fails the test at
    once instead of leaving the others waited for until the timeout"""
    import queue
    res, waited = [], 0
    while len(res) < len(procs):
        try:
            res.append(q.get(timeout=5))
        except queue.Empty:
            waited += 5
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or waited > timeout:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("rank process failed (exit codes %s) or timed out" % dead)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res
