#!/usr/bin/env python3
"""Host-side cost of building the full-domain-decomposition composite on BASELINE config C4's topology
(64^3 elements over a 2x2x2 rank grid), without a GPU: eight gloo ranks drive the product's host layer over
the CPU stand-in of the kernel C-ABI (tests/cpu_shim, test infrastructure: it links the oracle,
which is why this script lives under tests/).  The degree is lowered (default
N = 3, levels 3/1) so that eight ranks fit this container's memory: the superdomain -- every element that is
not in a rank's rings, at degree 1, graded by aggregation -- has exactly C4's size, and it is the part of
the setup whose cost grows with the rank count.

    python tests/c4_topology_setup_check.py [--E 64] [--N 3] [--reduction 2]
"""
import argparse
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, rdzv, E, N, red):
    import torch.distributed as dist

    import support as S
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    shim = os.path.join(ROOT, "tests", "cpu_shim", "_build", "libfdd_host_cpu.so")
    lib._host = lib._Lib(shim, os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
    import rendezvous

    rendezvous.init_gloo(rank, world, rdzv)
    try:
        H.init(0, use_torch_stream=False)
        H.set_print(False)
        H.comm_torch_callbacks(on_gpu=False)
        t0 = time.time()
        p = H.Problem.box((E, E, E), S.rank_grid(world), N, red, True)
        t1 = time.time()
        si = p.sub_info()
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20
        if rank == 0:
            print(f"composite built in {t1 - t0:.1f} s on rank 0 (peak RSS {rss:.2f} GiB)")
            print({k: si[k] for k in si if k != "is_composite"})
            print("superdomain levels:", list(p.sub_composite_levels()))
        p.close()
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--E", type=int, default=64)
    ap.add_argument("--N", type=int, default=3)
    ap.add_argument("--reduction", type=int, default=2)
    ap.add_argument("--ranks", type=int, default=8)
    a = ap.parse_args()
    import torch.multiprocessing as mp

    import rendezvous

    port = rendezvous.new()
    mp.spawn(worker, args=(a.ranks, port, a.E, a.N, a.reduction), nprocs=a.ranks, join=True)


if __name__ == "__main__":
    main()
