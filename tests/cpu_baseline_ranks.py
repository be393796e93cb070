#!/usr/bin/env python3
"""CPU baseline of an N-rank run, ONE HOST CORE PER RANK ("8 cores, one per
subdomain, for 8-GPU configs").  TEST INFRASTRUCTURE: the kernel C-ABI is served by tests/cpu_shim, i.e. by the serial C
restatement of the reference's kernels in oracle/ (OCCA-Serial semantics: every kernel a sequential loop nest), under the
same host-layer solver the GPU ranks run.  Every rank is a host thread of this one process (the in-process communicator
of host/comm.hpp stands where MPI stood; the kernels run outside the interpreter lock, so N ranks occupy N cores).  No
GPU library is loaded: this process must not count against the GPUs of the box it runs on.  Only bench.py's
`cpu_baseline` leg runs this (as a child process), never the product.

    python tests/cpu_baseline_ranks.py <ranks> <elements per rank and direction> <N> <reduction> <steps> <block_local 0/1>

prints one JSON line: {"seconds": ..., "setup_seconds": ..., "nodes": ..., "steps": ..., "ranks": ..., "elements": [...]}
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    world, e, N, red, steps, block_local = (int(a) for a in sys.argv[1:7])
    os.environ["FDD_HOST_THREADS"] = "1"  # one core per rank, setup included
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H, lib

    lib._host = lib._Lib(os.path.join(HERE, "cpu_shim", "_build", "libfdd_host_cpu.so"), os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
    P = H.rank_grid(world)
    E = tuple(e * p for p in P)

    def body(rank, size):
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
        nodes = p.info["num_total_nodes"]
        p.close()
        return dt, setup, nodes

    if world == 1:
        H.init(0, use_torch_stream=False)
        H.set_print(False)
        H.comm_single()
        out = [body(0, 1)]
    else:
        out = H.run_local_ranks(world, body)
    print(json.dumps({"seconds": max(o[0] for o in out), "setup_seconds": max(o[1] for o in out), "nodes": out[0][2], "steps": steps, "ranks": world, "elements": list(E)}))


if __name__ == "__main__":
    main()
