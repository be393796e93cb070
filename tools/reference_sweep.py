#!/usr/bin/env python3
"""The parameter sweep of the reference's harness (run.py:25-47,150-156: inner steps 1 / 2 / 4 / 8, Chebyshev order 1 / 2,
double / float, one V-cycle, polynomial reduction 6, outer GMRES(20) as poisson.cpp:224 hard-codes it) on one GPU, on the box
and on the Kershaw mesh (eps = 0.3) at 32^3 elements of degree 7 -- as run-time switches instead of sed + make.
Prints a markdown table: outer iterations and milliseconds to the reference's tolerance 1e-7.

    python tools/reference_sweep.py [--elements 32] [--degree 7] > profiles/r04_reference_sweep.md
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--elements", type=int, default=32)
    ap.add_argument("--degree", type=int, default=7)
    ap.add_argument("--reduction", type=int, default=6)
    a = ap.parse_args()
    import torch

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H

    torch.cuda.set_device(0)
    H.init(0, use_torch_stream=True)
    H.comm_single()
    H.set_print(False)
    E = (a.elements,) * 3
    print("# The reference's sweep (run.py:150-156) as run-time switches, one MI355X, %dx%dx%d elements, N = %d, reduction %d, outer GMRES(20)\n" % (E + (a.degree, a.reduction)))
    print("| mesh | Chebyshev order | precision | inner steps | outer iterations | converged | ms to 1e-7 | ms per Arnoldi step |")
    print("|---|---|---|---|---|---|---|---|")
    for mesh in ("box", "kershaw 0.3"):
        for cheby in (1, 2):
            p = H.Problem.box(E, (1, 1, 1), a.degree, a.reduction, True) if mesh == "box" else H.Problem.kershaw(E, (1, 1, 1), a.degree, a.reduction, 0.3)
            p.set_flag("amg_cheby_order", cheby)
            t0 = time.perf_counter()
            p.amg_build()
            build_s = time.perf_counter() - t0
            _, f = p.make_rhs(function_id=4, seed=1234)
            for bits in (64, 32):
                if bits == 32 and cheby < 2:
                    continue  # the float V-cycle runs the fused smoother, which needs order >= 2
                p.set_flag("preconditioner_precision", bits)
                for inner in (1, 2, 4, 8):
                    p.set_options(sub_num_vectors=inner, sub_max_iterations=inner)
                    p.solve_timed(f, "gmres")  # first solve of a setting: graph capture, float copies
                    its, hist, sec = p.solve_timed(f, "gmres")
                    rel = hist[-1] / hist[0]
                    print("| %s | %d | f%d | %d | %d | %s | %.1f | %.2f |" % (mesh, cheby, bits, inner, its, "yes" if rel <= 1e-7 else "NO (%.1e)" % rel, sec * 1e3, sec * 1e3 / max(its, 1)), flush=True)
            print("| %s | %d | | hierarchy built in %.1f s | | | | |" % (mesh, cheby, build_s), flush=True)
            p.close()


if __name__ == "__main__":
    main()
