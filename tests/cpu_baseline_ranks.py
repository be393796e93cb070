#!/usr/bin/env python3
"""CPU baseline of an N-rank run, ONE PROCESS = ONE CORE PER RANK ("8 cores, one
per subdomain, for 8-GPU configs").  TEST INFRASTRUCTURE: the kernel C-ABI is served by tests/cpu_shim, i.e. by the
serial C restatement of the reference's kernels in oracle/ (OCCA-Serial semantics: every kernel a sequential loop nest),
under the same host-layer solver the GPU ranks run, with a `gloo` group standing where MPI stood.  Only bench.py's
`cpu_baseline` leg runs this (as a child process), never the product.

    python tests/cpu_baseline_ranks.py <ranks> <elements per rank and direction> <N> <reduction> <steps> <block_local 0/1>

prints one JSON line: {"seconds": ..., "nodes": ..., "steps": ..., "ranks": ..., "elements": [...]}
"""
import json
import os
import socket
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def worker(rank, world, port, e, N, red, steps, block_local, out_file):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FDD_HOST_THREADS"] = "1"  # one core per rank, setup included
    import torch

    torch.set_num_threads(1)
    import torch.distributed as dist

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H, lib

    lib._host = lib._Lib(os.path.join(HERE, "cpu_shim", "_build", "libfdd_host_cpu.so"), os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H.init(0, use_torch_stream=False)
        H.set_print(False)
        if world > 1:
            H.comm_torch_callbacks(on_gpu=False)
        else:
            H.comm_single()
        P = H.rank_grid(world)
        E = tuple(e * p for p in P)
        t0 = time.perf_counter()
        p = H.Problem.box(E, P, N, red, True, block_local=bool(block_local))
        p.set_flag("sub_use_preconditioner", 0)  # the headline configuration: inner GMRES(4) without the V-cycle
        _, f = p.make_rhs(function_id=4, seed=1234 + rank)
        setup = time.perf_counter() - t0
        p.pcg_begin(f)
        H.barrier()
        t0 = time.perf_counter()
        p.pcg_steps(steps)
        H.barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            with open(out_file, "w") as fh:
                json.dump({"seconds": float(t.item()), "setup_seconds": setup, "nodes": p.info["num_total_nodes"], "steps": steps, "ranks": world, "elements": list(E)}, fh)
        p.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile

    import torch.multiprocessing as mp

    world, e, N, red, steps, block_local = (int(a) for a in sys.argv[1:7])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = os.path.join(tempfile.mkdtemp(), "baseline.json")
    mp.spawn(worker, args=(world, port, e, N, red, steps, block_local, out), nprocs=world, join=True)
    print(open(out).read())
