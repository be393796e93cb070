#!/usr/bin/env python3
"""Host time of building the full-domain-decomposition composite on C4's topology (64^3 elements, N = 7, 2x2x2 ranks)
with the eight ranks as threads of this process on the one GPU of the box (FDD_SETUP_TIMING=1: rank 0 prints the
phases of composite::build).  python tools/composite_setup_timing.py [ranks] [elements per rank and direction] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 7
P = H.rank_grid(world)
E = tuple(e * p for p in P)


def body(rank, size):
    H.set_print(False)
    t0 = time.time()
    p = H.Problem.box(E, P, N, N - 1, True)
    t1 = time.time()
    H.barrier()
    si = p.sub_info()
    p.close()
    return t1 - t0, si["num_values"]


out = H.run_local_ranks(world, body)
print("problem built: max over ranks %.1f s, min %.1f s; values per rank %s" % (max(o[0] for o in out), min(o[0] for o in out), sorted(set(o[1] for o in out))))
