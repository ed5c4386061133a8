"""8-rank readiness measured on ONE GPU.

Builds the slab of one interior rank of N (512^3 / 8 = 64 x 512 x 512 elements plus its ghost layers, the replicated coarse
hierarchy beside it) exactly as `DistributedMGSolver` would on that rank, replaces every message by a device copy of the same
bytes out of the rank's own planes and skips the all-reduces (`proxy=(N, rank)`), and times PCG iterations through the same
driver.  Values are meaningless (the ghost planes are not the neighbours'), work, launches and host-side cost are the real
rank's.  Reported:  T_rank (one rank's iteration), T_1 (the single-process iteration on the whole grid), the ratio
T_rank / (T_1 / N) -- 1.0 would be perfect strong scaling before any wire time --, the host-side share of T_rank (time the
driver spends outside its blocking reads of the dot products) and the messages per iteration.

usage: python tools/rank_proxy.py [N] [grid ...]      e.g.  python tools/rank_proxy.py 8 256 512
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch


def _single(ne, levels, iters):
    from helpers import BC_CANTILEVER, make_hip
    tps = make_hip(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    rho = torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g)
    tps.setElementDensities(rho)
    mg = tps.multigridSolver(levels)
    f = tps.buildLoadVector_device()
    mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, 2, 0.0, None, 1, 2, True)     # operators built, buffers touched
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, iters, 0.0, None, 1, 2, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert mg.last_iterations == iters
    del mg, tps
    torch.cuda.empty_cache()
    return dt / iters


def _proxy(ne, levels, world, rank, iters, c_driver=None, dist_levels=None):
    from helpers import BC_CANTILEVER, MATERIAL
    from ndr_amd import distributed as vd
    ds = vd.DistributedMGSolver(ne, [0.0, 0.0, 0.0], [2.0, 1.0, 1.0], BC_CANTILEVER, MATERIAL, levels, dist_levels=dist_levels, proxy=(world, rank))
    if c_driver is not None:
        ds.use_c_driver = bool(c_driver)
    g = torch.Generator(device="cuda").manual_seed(88)
    own = torch.rand((ds.part.x1 - ds.part.x0) * ne[1] * ne[2], dtype=torch.float64, device="cuda", generator=g)
    if ds.T >= 2:
        ds.set_local_densities(own)
    else:
        whole = torch.rand(ne[0] * ne[1] * ne[2], dtype=torch.float64, device="cuda", generator=g)
        ds.set_global_densities(whole)
    f = ds.local_loads()
    f += 1e-3 * torch.randn(f.shape, dtype=torch.float64, device="cuda", generator=g)      # (a slab away from the load has f = 0: PCG would stop at once)
    if getattr(ds, "use_c_driver", True):
        # one library call per solve: the host side is the callbacks (device copies here) and one blocking read per iteration
        ds.pcg(torch.zeros_like(f), f, 2, 0.0, 1, 2, True)
        torch.cuda.synchronize()
        for hx in ds.halos:
            hx.messages = 0
        t0 = time.perf_counter()
        ds.pcg(torch.zeros_like(f), f, iters, 0.0, 1, 2, True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        done = ds.last_iterations
        out = {"seconds_per_iteration": dt / max(done, 1), "iterations_timed": done,
               "messages_per_iteration": sum(getattr(hx, "messages", 0) for hx in ds.halos) / max(done, 1),
               "distributed_levels": ds.Ld + 1, "slab_elements": [ds.geom[0].nx, ne[1], ne[2]],
               "driver": "C (vfem_mg_pcg_slab: one call per solve, callbacks for halo / all-reduce)"}
        del ds
        torch.cuda.empty_cache()
        return out
    waits = [0.0]
    plain_dot = ds.dot

    def timed_dot(a, b):
        v = ds.halos[0].dot(a, b)
        t0 = time.perf_counter()
        out = float(v.item())
        waits[0] += time.perf_counter() - t0
        return out

    ds.dot = timed_dot
    ds.pcg(torch.zeros_like(f), f, 2, 0.0, 1, 2, True)
    torch.cuda.synchronize()
    for hx in ds.halos:
        hx.messages = 0
    waits[0] = 0.0
    t0 = time.perf_counter()
    ds.pcg(torch.zeros_like(f), f, iters, 0.0, 1, 2, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    done = ds.last_iterations
    out = {"seconds_per_iteration": dt / max(done, 1), "iterations_timed": done,
           "host_side_seconds_per_iteration": (dt - waits[0]) / max(done, 1),
           "messages_per_iteration": sum(getattr(hx, "messages", 0) for hx in ds.halos) / max(done, 1),
           "distributed_levels": ds.Ld + 1, "slab_elements": [ds.geom[0].nx, ne[1], ne[2]],
           "driver": "C (one call per solve, callbacks for halo / all-reduce)" if getattr(ds, "use_c_driver", False) else "python"}
    ds.dot = plain_dot
    del ds
    torch.cuda.empty_cache()
    return out


def _proxy_q2(ne, levels, world, rank, iters):
    """one rank of `world` of the DEGREE-2 slab solver (BASELINE config 5: 512^3 over 8 ranks does not fit one device as a whole,
    a rank's slab does): seconds per PCG iteration, the operator update, peak device memory"""
    from helpers import BC_CANTILEVER, MATERIAL
    from ndr_amd import distributed_q2 as vq
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    ds = vq.DistributedMGSolverQ2(ne, [0.0, 0.0, 0.0], [2.0, 1.0, 1.0], BC_CANTILEVER, MATERIAL, levels, proxy=(world, rank))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    g = torch.Generator(device="cuda").manual_seed(88)
    own = torch.rand((ds.part.x1 - ds.part.x0) * ne[1] * ne[2], dtype=torch.float64, device="cuda", generator=g)
    ds.set_local_densities(own)
    f = ds.local_loads()
    f += 1e-3 * torch.randn(f.shape, dtype=torch.float64, device="cuda", generator=g)
    def solve(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds.pcg(torch.zeros_like(f), f, n, 0.0, 1, 2, True)         # (every call refreshes the operators first, as the reference's solve does)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    t_first = solve(1)                                              # allocations, operator update, one iteration
    t_one = solve(1)
    for hx in ds.halos:
        hx.messages = 0
    dt = solve(1 + iters)
    msgs = sum(getattr(hx, "messages", 0) for hx in ds.halos)
    per_it = (dt - t_one) / iters
    done = ds.last_iterations
    g0 = ds.geom[0]
    out = {"degree": 2, "grid": "%dx%dx%d" % tuple(ne), "ranks": world, "rank": rank, "levels": levels, "distributed_levels": ds.Ld + 1,
           "slab_elements": [g0.nx, ne[1], ne[2]], "slab_nodes": int(g0.n_planes * g0.plane),
           "seconds_build": t_build, "seconds_first_call": t_first, "seconds_solve_of_1_iteration": t_one,
           "seconds_solve_of_%d_iterations" % (1 + iters): dt, "iterations_done": done,
           "seconds_per_iteration": per_it, "seconds_operator_update": t_one - per_it,
           "messages_per_solve": msgs,
           "peak_device_GB": torch.cuda.max_memory_allocated() / 1e9,
           "relative_residual": ds.last_relative_residual}
    del ds
    torch.cuda.empty_cache()
    return out


def run(world=8, grids=(256, 512), iters=5):
    res = []
    for n in grids:
        ne, levels = (n, n, n), {128: 4, 256: 5, 512: 6}[n]
        t1 = _single(ne, levels, iters)
        row = {"grid": "%dx%dx%d" % ne, "levels": levels, "ranks": world, "T_1_seconds_per_iteration": t1}
        pr = _proxy(ne, levels, world, world // 2, iters, c_driver=True)
        row["rank_proxy"] = pr
        row["rank_proxy_python_driver"] = _proxy(ne, levels, world, world // 2, iters, c_driver=False)
        row["T_rank_over_T_1_per_rank"] = pr["seconds_per_iteration"] / (t1 / world)
        row["strong_scaling_upper_bound"] = t1 / pr["seconds_per_iteration"]
        res.append(row)
    return res


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "levels":
    # python tools/rank_proxy.py levels N grid : one rank's iteration for every choice of the number of distributed levels
    w, n = int(sys.argv[2]), int(sys.argv[3])
    ne, levels = (n, n, n), {128: 4, 256: 5, 512: 6}[n]
    for ld in range(0, levels):
        try:
            r = _proxy(ne, levels, w, w // 2, 5, c_driver=True, dist_levels=ld)
            print(json.dumps({"grid": n, "ranks": w, "dist_levels_arg": ld, **r}), flush=True)
        except RuntimeError as e:
            print(json.dumps({"grid": n, "dist_levels_arg": ld, "error": str(e)}), flush=True)
    sys.exit(0)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "q2":
    # python tools/rank_proxy.py q2 N grid [levels] [iterations]: one rank of N of the degree-2 slab solver
    w, n = int(sys.argv[2]), int(sys.argv[3])
    levels = int(sys.argv[4]) if len(sys.argv) > 4 else {64: 4, 128: 5, 256: 6, 512: 7}[n]
    print(json.dumps(_proxy_q2((n, n, n), levels, w, w // 2, int(sys.argv[5]) if len(sys.argv) > 5 else 4), indent=1))
    sys.exit(0)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "one":
    # python tools/rank_proxy.py one N grid [dist_levels]: a single proxy run (for rocprofv3)
    w, n = int(sys.argv[2]), int(sys.argv[3])
    ne, levels = (n, n, n), {128: 4, 256: 5, 512: 6}[n]
    print(json.dumps(_proxy(ne, levels, w, w // 2, 5, c_driver=True, dist_levels=int(sys.argv[4]) if len(sys.argv) > 4 else None)))
    sys.exit(0)
if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    grids = tuple(int(a) for a in sys.argv[2:]) or (256, 512)
    print(json.dumps(run(w, grids), indent=1))
