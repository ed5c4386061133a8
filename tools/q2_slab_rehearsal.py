"""Rehearsal of the slab-decomposed degree-2 CG-MG on ONE GPU: N rank processes (gloo, planes staged through the host) share
the device.  Not a measurement of multi-GPU speed -- it checks that the distributed path runs at size (memory per rank,
iteration count against the single-process solve of the same grid) and prints what each side took.
    python tools/q2_slab_rehearsal.py RANKS N LEVELS [--single]
"""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _rank(rank, world, port, n, levels):
    import torch
    import torch.distributed as dist
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from ndr_amd.distributed_q2 import bench_pcg_q2
    res = bench_pcg_q2((n, n, n), levels)
    res["rank"] = rank
    print(json.dumps(res), flush=True)
    dist.destroy_process_group()


def main():
    world, n, levels = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    if "--single" in sys.argv:
        import bench
        print(json.dumps({"single_process": bench.degree2_pcg_rate(n, levels)}), flush=True)
        return
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank, args=(r, world, port, n, levels)) for r in range(world)]
    for p in procs:
        p.start()
    code = 0
    for p in procs:
        p.join()
        code = max(code, abs(p.exitcode or 0))
    raise SystemExit(code)


if __name__ == "__main__":
    main()
