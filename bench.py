#!/usr/bin/env python3
"""Benchmark of the voxel-FEM hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

A "step" is one matrix-free stiffness apply  Ku = K(rho) u  (the SpMV of BASELINE.json's metric) over the
whole Q1 fp64 voxel grid (default 512^3, `--grid` to change) with u, rho and Ku resident in HBM.  With N > 1
ranks the grid is split into x-slabs (one per GPU) and every step includes the halo exchange of u
(`ndr_amd.distributed`); the work is fixed as N grows (strong scaling).  Beside the headline value the same
JSON line reports
  * CG-MG iterations/s of the multigrid-preconditioned CG compliance solve (reference settings: tol 1e-4,
    one full-multigrid cycle per iteration, 2+2 symmetric coloured Gauss-Seidel sweeps, zero initial guess),
  * `roofline`: algorithmic HBM bytes of one apply / its measured duration, against the 8 TB/s HBM peak,
  * `cpu_baseline`: the CPU oracle (a port of the reference's element loop with its thread-private
    accumulators) timed on this host on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(ne):
    """read u once, write Ku once (numNodes x 3 fp64 each), read rho once."""
    nn = (ne[0] + 1) * (ne[1] + 1) * (ne[2] + 1)
    nel = ne[0] * ne[1] * ne[2]
    return 2 * nn * 3 * 8 + nel * 8


def time_apply(tps, u, steps, warmup, variant=0):
    """kernel time per launch from HIP events on the launch stream (torch's current stream)."""
    for _ in range(warmup):
        tps.applyK_device(u, variant)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    out = torch.empty_like(u)
    from ndr_amd import _lib
    from ndr_amd.pyVoxelFEM import _ptr, _stream
    lib = _lib.load()
    _lib.check(lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), int(variant), _stream()))     # first touch of `out`, untimed
    for a, b in evs:
        a.record()
        _lib.check(lib.vfem_sim_apply_k(tps._h, _ptr(u), _ptr(out), int(variant), _stream()))
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e-3


def hbm_stream_calibration(nbytes):
    """What THIS device's HBM gives plain streaming kernels (torch elementwise ops on arrays of the apply's size): boxes of the pool
    differ by up to 17 % on write-heavy streams (profiles/r03_apply_mixprobe.txt: 4.7 vs 5.5 TB/s on the apply's 1.5 : 1 read : write
    mix), which moves the apply's fraction of the 8 TB/s peak with it.  Reported beside `roofline`, never used in it."""
    n = nbytes // 8
    a = torch.zeros(n, dtype=torch.float64, device="cuda")
    b = torch.ones(n, dtype=torch.float64, device="cuda")
    c = torch.empty(n, dtype=torch.float64, device="cuda")

    def rate(fn, moved, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return moved * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12

    out = {"array_GB": nbytes / 1e9,
           "fill_0R1W_TBs": rate(lambda: c.fill_(1.0), nbytes),
           "copy_1R1W_TBs": rate(lambda: c.copy_(a), 2 * nbytes),
           "add_2R1W_TBs": rate(lambda: torch.add(a, b, out=c), 3 * nbytes),
           "note": "the apply moves 1.5 bytes read per byte written: its mix lies between copy and add"}
    del a, b, c
    torch.cuda.empty_cache()
    return out


def spmv_rate(ne, steps=50):
    """the metric's second grid (256^3): event-timed kernel rate of the same apply"""
    from helpers import make_hip
    tps = make_hip(ne, ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((tps.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:          # the clocks take ~0.1 s of continuous work to settle (profiles/r01_bench_apply512_per_call.json)
        tps.applyK_device(u)
        torch.cuda.synchronize()
    sec = time_apply(tps, u, steps, 10)
    ab = algorithmic_bytes(ne)
    return {"grid": "%dx%dx%d" % ne, "kernel_ms": sec * 1e3, "gvoxel_per_s": ne[0] * ne[1] * ne[2] / sec / 1e9,
            "algorithmic_GBs": ab / sec / 1e9, "frac_of_8TBs": ab / sec / 1e9 / HBM_PEAK_GBS}


def host_description():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "logical_cores": os.cpu_count() or 1,
            "usable_cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)}


def cpu_cg_mg(ne=(128, 64, 64), levels=3, only_cap=False):
    """the reference-algorithm CG-MG (the oracle: element-loop applyK with thread-private accumulators, 8-colour
    Gauss-Seidel, the same FMG/PCG control flow and settings as the GPU leg) on config 2's grid, timed on the host with
    (i) the reference drivers' thread cap, min(cores, 8) (train_voxelfem.py:38-39), and (ii) all usable cores (at most 64)."""
    from oracle import vfem_oracle as vo
    from helpers import BC_CANTILEVER, make_oracle, seeded_density
    cores = host_description()["usable_cores"]
    o = make_oracle(ne, ([0, 0, 0], [2, 1, 1]), BC_CANTILEVER, seeded_density(ne, 88))
    f = o.build_load_vector()
    out = {"grid": "%dx%dx%d" % tuple(ne), "levels": levels, "settings": "tol 1e-4, 1 FMG cycle per iteration, 2+2 symmetric sweeps, zero initial guess"}
    for key, threads in (("policy_i_reference_cap", min(cores, 8)), ("policy_ii_all_cores", min(cores, 64))):
        if key == "policy_ii_all_cores" and only_cap:
            continue
        if key == "policy_ii_all_cores" and threads == min(cores, 8):
            out[key] = "same as policy (i): the host offers %d cores" % cores
            continue
        mg = vo.OracleMG(o, levels, nthreads=threads)
        t0 = time.perf_counter()
        u = mg.pcg(np.zeros_like(f), f, 100, 1e-4, 1, 2, True)
        dt = time.perf_counter() - t0
        out[key] = {"threads": threads, "iterations": mg.last_iters, "seconds": dt, "iterations_per_s": mg.last_iters / dt,
                    "compliance": float(np.sum(f * u)), "includes": "operator update (Galerkin element matrices) as in MG.hh:690-691"}
    return out


def cpu_baseline(sample_ne, seconds_budget=8.0):
    """The oracle's element-loop applyK (ParallelAssembly-style private accumulators) on the host cores."""
    from oracle import vfem_oracle as vo
    from helpers import make_oracle, seeded_density
    cores = host_description()["usable_cores"]
    threads = min(cores, 8)          # the reference drivers cap applyK at min(physical cores, 8) threads
    o = make_oracle(sample_ne, ([0, 0, 0], [1, 1, 1]), None, seeded_density(sample_ne, 88))
    u = np.random.default_rng(0).standard_normal((o.num_nodes, 3))
    o.apply_k(u, threads)
    t0, reps = time.perf_counter(), 0
    while True:
        o.apply_k(u, threads)
        reps += 1
        if time.perf_counter() - t0 > seconds_budget or reps >= 50:
            break
    dt = (time.perf_counter() - t0) / reps
    res = {"value": o.num_elems / dt / 1e9, "unit": "GVoxel/s", "cores": threads, "kind": "port",
           "sample": "%dx%dx%d Q1 fp64 applyK, %d repetitions, %d OpenMP threads of %d host cores"
                     % (sample_ne[0], sample_ne[1], sample_ne[2], reps, threads, cores)}
    # the same loop with every parallel region on (up to 64 of) the host's cores, SURVEY 8(d) policy (ii)
    many = min(cores, 64)
    if many > threads:
        o.apply_k(u, many)
        t0, reps2 = time.perf_counter(), 0
        while True:
            o.apply_k(u, many)
            reps2 += 1
            if time.perf_counter() - t0 > 5.0 or reps2 >= 50:
                break
        res["all_cores"] = {"value": o.num_elems * reps2 / (time.perf_counter() - t0) / 1e9, "cores": many, "repetitions": reps2}
    res["host"] = host_description()
    # the metric's other grid, 256^3 (one repetition is ~1 s of CPU work), and the solve on config 2's grid
    del o, u
    ne256 = (256, 256, 256)
    o = make_oracle(ne256, ([0, 0, 0], [1, 1, 1]), None, seeded_density(ne256, 88))
    u = np.random.default_rng(0).standard_normal((o.num_nodes, 3))
    o.apply_k(u, threads)
    t0, reps3 = time.perf_counter(), 0
    while reps3 < 2:
        o.apply_k(u, threads)
        reps3 += 1
    res["spmv_256"] = {"value": o.num_elems * reps3 / (time.perf_counter() - t0) / 1e9, "unit": "GVoxel/s", "cores": threads,
                       "sample": "256x256x256 Q1 fp64 applyK, %d repetitions" % reps3}
    del o, u
    res["cg_mg"] = cpu_cg_mg()
    # the reference itself is built with -march=native -ffast-math (VoxelFEM/CMakeLists.txt:43); the checker
    # above is gcc -O2.  The same loops once more from a copy compiled here with the reference's flags (+ -O3), so that the stated
    # baseline does not flatter: `value` of this object stays the -O2 port the tests check, `reference_flags` is the faster one
    vo.use_native_build(True)
    try:
        o = make_oracle(sample_ne, ([0, 0, 0], [1, 1, 1]), None, seeded_density(sample_ne, 88))
        u = np.random.default_rng(0).standard_normal((o.num_nodes, 3))
        o.apply_k(u, threads)
        t0, repsn = time.perf_counter(), 0
        while True:
            o.apply_k(u, threads)
            repsn += 1
            if time.perf_counter() - t0 > 4.0 or repsn >= 50:
                break
        dtn = (time.perf_counter() - t0) / repsn
        del o, u
        res["reference_flags"] = {"flags": "gcc -O3 -march=native -ffast-math -funroll-loops -fopenmp", "value": int(np.prod(sample_ne)) / dtn / 1e9,
                                  "unit": "GVoxel/s", "cores": threads, "repetitions": repsn, "cg_mg": cpu_cg_mg(only_cap=True)}
    finally:
        vo.use_native_build(False)
    return res


def pcg_rate(ne, levels, dom):
    from helpers import BC_CANTILEVER, make_hip
    tps = make_hip(ne, dom, BC_CANTILEVER, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    rho = torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g)
    tps.setElementDensities(rho)
    mg = tps.multigridSolver(levels)
    f = tps.buildLoadVector_device()
    x0 = torch.zeros_like(f)
    mg.preconditionedConjugateGradient_device(x0, f, 1, 1e-4, None, 1, 2, True)       # warm-up (allocations, first-touch)
    # the timed solve includes the per-solve operator update (Galerkin matrices, stencils, dense coarsest inverse).
    # The library skips that update when the moduli have not changed since the last one, so the densities are set again here,
    # as a design iteration would
    # median of three solves (one perf_counter sample is not a measurement)
    samples = []
    for _ in range(3):
        tps.setElementDensities(rho)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        u = mg.preconditionedConjugateGradient_device(x0, f, 100, 1e-4, None, 1, 2, True)
        torch.cuda.synchronize()
        samples.append(time.perf_counter() - t0)
    dt = float(np.median(samples))
    return {"grid": "%dx%dx%d" % tuple(ne), "levels": levels, "iterations": mg.last_iterations,
            "seconds": dt, "seconds_samples": samples, "iterations_per_s": mg.last_iterations / dt,
            "relative_residual": mg.last_relative_residual, "compliance": float((f * u).sum()),
            "includes_operator_update": True}


def mlp_rate(side=(512, 256, 256), es=1024, nn_=512, nl=4, sigma=4.0, reps=3):
    """MLP-forward voxels/s at the run.md sizes (2048 -> 512 -> 512 -> 512 -> 1), random-init weights, whole grid."""
    from ndr_amd.mlp import MLP
    rng = np.random.default_rng(88)
    B = (rng.standard_normal((es, 3)) * sigma).astype(np.float32)
    Ws = [rng.standard_normal((nn_, 2 * es)).astype(np.float32) / np.sqrt(2 * es)]
    Ws += [rng.standard_normal((nn_, nn_)).astype(np.float32) / np.sqrt(nn_) for _ in range(nl - 2)]
    Ws += [rng.standard_normal((1, nn_)).astype(np.float32) / np.sqrt(nn_)]
    bs = [rng.standard_normal(nn_).astype(np.float32) * 0.1 for _ in range(nl - 1)] + [np.array([0.4], np.float32)]
    m = MLP(3, 1, nn_, nl, es, sigma)
    m.load_arrays(B, Ws, bs)
    nv = int(np.prod(side))
    flop = 2.0 * (3 * es + 2 * es * nn_ + (nl - 2) * nn_ * nn_ + nn_) * nv

    def timed(precision):
        m.precision = precision
        m.forward_grid(side)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            m.forward_grid(side)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # the forward pass at the reference's precision (networks.MLP is fp32): fused kernel, split fp16 operands, 3 MFMA products per
    # product (kernels_mlp_x3.hip).  "tflops" counts the network's 3.15 MFLOP per voxel once; the matrix pipe executes three times
    # that, which is what the fraction of the dense f16 peak is taken on; the fp32 MFMA peak (157 TF) is quoted for scale
    dt = timed("fp32")
    res = {"grid": "%dx%dx%d" % tuple(side), "network": "%d->%d x%d->1" % (2 * es, nn_, nl - 1),
           "operands": "split f16 (hi + lo 2^-11), f32 accumulate: reference (fp32) precision",
           "seconds": dt, "voxels_per_s": nv / dt, "tflops": flop / dt / 1e12,
           "mfma_frac_of_2.5PF": 3.0 * flop / dt / 2.5e15, "times_the_157TF_fp32_mfma_peak": flop / dt / 157.3e12}
    # the fast option: plain fp16 operands (4e-4 on the logits, 1e-5 on the config-4 compliance)
    d16 = timed("fp16")
    res["fp16_option"] = {"seconds": d16, "voxels_per_s": nv / d16, "tflops": flop / d16 / 1e12, "mfma_frac_of_2.5PF": flop / d16 / 2.5e15,
                          "operands": "f16, f32 accumulate"}
    # parameter gradients of the whole grid (recomputed forward with saved activations + data path + weight-gradient kernels)
    g = torch.randn(nv, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    m.backward_grid(side, g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.backward_grid(side, g)
    torch.cuda.synchronize()
    db = time.perf_counter() - t0
    # a training step as TrainableMLP runs it: the forward keeps the first layer's activations (68.7 GB at this size), the backward starts from them
    m.precision = "fp32"
    m.set_keep_first_layer(True)
    m.forward_grid(side)
    m.backward_grid(side, g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.forward_grid(side)
    torch.cuda.synchronize()
    tf = time.perf_counter() - t0
    t0 = time.perf_counter()
    m.backward_grid(side, g)
    torch.cuda.synchronize()
    tb = time.perf_counter() - t0
    m.set_keep_first_layer(False)
    res["training_step"] = {"forward_seconds": tf, "backward_seconds": tb, "seconds": tf + tb,
                            "note": "VFEM_MLP_OPT_KEEP_FIRST: first-layer activations kept by the forward, not recomputed by the backward; same gradients bit for bit"}
    macs = (3 * es + 2 * es * nn_ + (nl - 2) * nn_ * nn_ + nn_) + (nl - 2) * nn_ * nn_ + (2 * es * nn_ + (nl - 2) * nn_ * nn_ + nn_)
    res["backward"] = {"seconds": db, "voxels_per_s": nv / db, "tflops": 2.0 * macs * nv / db / 1e12,
                       "note": "reference precision throughout: forward recompute with saved split activations (kernels_mlp_x3.hip) + data pass + own "
                               "weight-gradient kernels with in-kernel Fourier features (kernels_mlp_bwd.hip); no library GEMM; tflops counts the "
                               "network's multiply-adds once, the matrix pipe executes three times that"}
    return res


def degree2_rate(ne=(512, 512, 512), reps=3):
    """matrix-free SpMV of the 27-node (degree-2) elements, SURVEY 8(d) M1 (Q2): 393 B/voxel algorithmic"""
    from ndr_amd import pyVoxelFEM as pv
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [1, 1, 1]), list(ne))
    t.E_min = 1e-4
    g = torch.Generator(device="cuda").manual_seed(88)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((t.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)
    for _ in range(3):                 # the first write into a fresh 26 GB allocation costs ~0.7 s (page population): touch both
        out = t.applyK_device(u)       # buffers the caching allocator alternates between before timing
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = t.applyK_device(u)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nvox = ne[0] * ne[1] * ne[2]
    ab = 2 * t.numNodes() * 24 + nvox * 8
    del out
    return {"grid": "%dx%dx%d" % tuple(ne), "nodes": t.numNodes(), "seconds": dt, "gvoxel_per_s": nvox / dt / 1e9,
            "algorithmic_GBs": ab / dt / 1e9, "frac_of_8TBs": ab / dt / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_voxel": ab / nvox,
            "note": "marching kernel: reflection-mode blocks (855 of 6561 multiply-adds), x-march with in-block y hand-off, rows loaded and stored per lane by buffer accesses (no LDS transposes), 2 colour launches"}


def degree2_pcg_rate(n=128, levels=5):
    """CG-MG iterations/s of the degree-2 (27-node) discretisation, cantilever, reference solver settings"""
    from helpers import BC_CANTILEVER, MATERIAL
    from ndr_amd import pyVoxelFEM as pv
    t = pv.TensorProductSimulator([2, 2, 2], ([0, 0, 0], [2, 1, 1]), [n, n, n])
    t.readMaterial(MATERIAL)
    t.applyDisplacementsAndLoadsFromFile(BC_CANTILEVER)
    t.E_min = 1e-4
    g = torch.Generator(device="cuda").manual_seed(88)
    t.setElementDensities(torch.rand(t.numElements(), dtype=torch.float64, device="cuda", generator=g))
    mg = t.multigridSolver(levels)
    f = t.buildLoadVector_device()
    mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, 1, 1e-4, None, 1, 2, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    u = mg.preconditionedConjugateGradient_device(torch.zeros_like(f), f, 100, 1e-4, None, 1, 2, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"grid": "%dx%dx%d" % (n, n, n), "nodes": t.numNodes(), "levels": levels, "iterations": mg.last_iterations, "seconds": dt,
            "iterations_per_s": mg.last_iterations / dt, "relative_residual": mg.last_relative_residual,
            "compliance": float((f * u).sum())}


def _rank_main(rank, world, port, argv):
    """entry of a rank process started by `launch_ranks` (fresh interpreter, nothing has touched the GPU yet)"""
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    sys.argv = [sys.argv[0]] + list(argv)
    main()


def launch_ranks(world, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (spawn context: fresh children, no exec, and this
    parent never initialises the GPU), wait for them, return the worst exit code.  Rank 0 prints the JSON line."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, argv)) for r in range(world)]
    for pr in procs:
        pr.start()
    code = 0
    for pr in procs:
        pr.join()
        code = max(code, abs(pr.exitcode or 0))
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--no-cg", action="store_true", help="skip the CG-MG side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--rank-proxy", type=int, default=0, metavar="N",
                    help="also time one rank's slab of an N-rank CG-MG run on this GPU (tools/rank_proxy.py): an upper bound on strong scaling")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks ourselves.  With fewer devices than ranks (one-GPU box) the ranks share the devices
        # and torch.distributed runs on gloo -- a rehearsal of the same code path, not a measurement.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the voxel-FEM path has no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    ne = tuple(args.grid)

    if world > 1:
        from ndr_amd import distributed as vd
        res = vd.bench_apply(ne, args.steps, args.warmup, with_cg=not args.no_cg)
        if rank == 0:
            print(json.dumps(res))
        return

    from helpers import make_hip
    tps = make_hip(ne, ([0, 0, 0], [1, 1, 1]), None, None, v0=0.5)
    g = torch.Generator(device="cuda").manual_seed(88)
    tps.setElementDensities(torch.rand(tps.numElements(), dtype=torch.float64, device="cuda", generator=g))
    u = torch.randn((tps.numNodes(), 3), dtype=torch.float64, device="cuda", generator=g)

    # timed region: exactly K steps between two synchronisations (single rank: no barrier partner)
    # setup, not a step: the results of successive steps alternate between two blocks of torch's caching allocator; a block that
    # has never been written costs ~20 ms per GB on first touch (page population), which must not land in the timed steps
    a_, b_ = torch.zeros_like(u), torch.zeros_like(u)
    del a_, b_
    out = None
    for _ in range(args.warmup):
        out = tps.applyK_device(u)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tps.applyK_device(u)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms_per_step = wall / args.steps * 1e3
    nvox = ne[0] * ne[1] * ne[2]

    kernel_s = time_apply(tps, u, args.steps, 1)
    ab = algorithmic_bytes(ne)
    # HBM bytes per launch by the PMC counters: collected in separate rocprofv3 --pmc passes (FETCH_SIZE x 2, WRITE_SIZE; the
    # guide's gfx950 correction) and committed under profiles/ -- NOT measured inside this run; the source is named in the line
    traffic, traffic_source = None, None
    prof = os.path.join(ROOT, "profiles", "apply_traffic.json")
    if os.path.exists(prof):
        with open(prof) as fh:
            t = json.load(fh)
        if t.get("grid") == list(ne):
            traffic, traffic_source = t.get("hbm_bytes_per_launch"), "from profile " + str(t.get("source"))
    roofline = {"bound": "hbm", "achieved": ab / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ab / kernel_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": "vfem::k_apply_dma", "algorithmic_bytes_per_launch": ab, "kernel_ms": kernel_s * 1e3}
    del out
    roofline["same_device_streaming"] = hbm_stream_calibration(3 * (ne[0] + 1) * (ne[1] + 1) * (ne[2] + 1) * 8)

    result = {
        "metric": "matrix-free SpMV GVoxel/s (Q1 fp64, %dx%dx%d); CG-MG iterations/s reported in cg_mg" % ne,
        "value": nvox / (wall / args.steps) / 1e9, "unit": "GVoxel/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "K(rho) u on a %dx%dx%d voxel grid, trilinear hexahedra, fp64, seeded U[0,1] densities "
                               "(SIMP p=3, Emin=1e-4), u ~ N(0,1)" % ne,
                   "grid": list(ne), "parallelism": "1 GPU"},
        "roofline": roofline,
    }
    if not args.no_cg:
        del u, tps
        torch.cuda.empty_cache()
        result["spmv_other_grids"] = [spmv_rate((256, 256, 256))]
        cg = []
        try:
            cg.append(pcg_rate((256, 256, 256), 5, ([0, 0, 0], [2, 1, 1])))
            if ne == (512, 512, 512):
                cg.append(pcg_rate((512, 512, 512), 6, ([0, 0, 0], [2, 1, 1])))
        except RuntimeError as e:       # reported, never hidden
            cg.append({"error": str(e)})
        result["cg_mg"] = cg
        try:
            result["mlp_forward"] = mlp_rate()
        except RuntimeError as e:
            result["mlp_forward"] = {"error": str(e)}
        result["degree2_cg_mg"] = []
        for q2n, q2l in ((128, 5), (256, 6)):      # 256^3: 135 M nodes; level 1 on the fly (its stored matrices would be 110 GB)
            torch.cuda.empty_cache()
            try:
                result["degree2_cg_mg"].append(degree2_pcg_rate(q2n, q2l))
            except RuntimeError as e:
                result["degree2_cg_mg"].append({"grid": "%dx%dx%d" % (q2n, q2n, q2n), "error": str(e)})
        result["degree2_spmv"] = []
        for q2ne in ((256, 256, 256), (512, 512, 512)):
            torch.cuda.empty_cache()
            try:
                result["degree2_spmv"].append(degree2_rate(q2ne))
            except RuntimeError as e:
                result["degree2_spmv"].append({"grid": "%dx%dx%d" % q2ne, "error": str(e)})
    if args.rank_proxy > 1:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import rank_proxy
        torch.cuda.empty_cache()
        result["rank_proxy"] = rank_proxy.run(args.rank_proxy)
    if not args.no_cpu:
        torch.cuda.synchronize()
        result["cpu_baseline"] = cpu_baseline((160, 160, 160))
    print(json.dumps(result))


if __name__ == "__main__":
    main()
