#!/usr/bin/env python3
"""Row-owner mode smoke run: WORLD rank processes on one GPU over gloo (host transport), one graph, optimize(iters);
prints per-rank results.  Usage: python scripts/owner_smoke.py [world] [V] [E] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, V, E, iters):
    import faulthandler
    faulthandler.enable()
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparse_gslam_amd import capi, synth
    g = synth.manhattan(V, E, seed=2, info_mode="full")

    def allreduce(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t)

    with capi.Optimizer(0, verbose=1 if rank == 0 else 0) as o:
        o.comm_init_host(world, rank, allreduce)
        print(f"[rank {rank}] set_graph", flush=True)
        o.set_graph(*g.arrays())
        print(f"[rank {rank}] {o.solver_description()} bytes {o.level0_bytes()}", flush=True)
        done, st = o.optimize(iters)
        P = o.get_poses()
        print(f"[rank {rank}] done {done} pcg {st['pcg_iters']} chi2 {st['chi2'][-1]:.9g} posesum {np.abs(P).sum():.12g}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, world, port, V, E, iters)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=150)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    print("exit codes", codes, flush=True)
    sys.exit(0 if all(c == 0 for c in codes) else 1)
