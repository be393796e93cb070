#!/usr/bin/env python3
"""Why the composite WITHOUT the V-cycle needs more outer iterations than block-local.

Measures, on the CPU stand-in of the kernel C-ABI under a gloo group (test infrastructure, like
tests/composite_iteration_counts.py), for one rank of an R-rank box:

  1. the diagonal of the dof-space operator of the inner iteration (Qt A_L Q | A_sup), by probing with unit vectors,
     split by dof class: own dofs, ring dofs at degree N, ring dofs at the lower degrees, interface, superdomain;
  2. what four steps of unpreconditioned GMRES do to the right-hand side the first preconditioner application sees:
     residual reduction overall and BY CLASS, for block-local, the composite, and the composite after symmetric
     diagonal scaling D^-1/2 A D^-1/2.

python tests/composite_operator_analysis.py [ranks] [E per rank] [N]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gmres(op, f, m):
    """m steps of GMRES from zero; returns x and the true residual"""
    n = len(f)
    V = np.zeros((m + 1, n))
    H = np.zeros((m + 1, m))
    beta = np.linalg.norm(f)
    V[0] = f / beta
    for j in range(m):
        w = op(V[j])
        for i in range(j + 1):
            H[i, j] = w @ V[i]
        for i in range(j + 1):
            w -= H[i, j] * V[i]
        H[j + 1, j] = np.linalg.norm(w)
        V[j + 1] = w / H[j + 1, j]
    e1 = np.zeros(m + 1)
    e1[0] = beta
    y = np.linalg.lstsq(H, e1, rcond=None)[0]
    x = V[:m].T @ y
    return x, f - op(x)


def worker(rank, world, rdzv, e, N, red, out_file):
    import torch.distributed as dist

    import support as S
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H, lib

    lib._host = lib._Lib(os.path.join(ROOT, "tests/cpu_shim/_build/libfdd_host_cpu.so"), os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
    import rendezvous

    rendezvous.init_gloo(rank, world, rdzv)
    H.init(0, use_torch_stream=False)
    H.set_print(False)
    H.comm_torch_callbacks(on_gpu=False)
    Pg = S.rank_grid(world)
    E = tuple(e * p for p in Pg)
    res = {"ranks": world, "elements": E, "N": N, "reduction": red}
    for name, bl in (("block_local", True), ("composite", False)):
        p = H.Problem.box(E, Pg, N, red, True, block_local=bl)
        p.set_flag("sub_use_preconditioner", 0)
        _, f = p.make_rhs(function_id=4, seed=1234 + rank)
        si = p.sub_info()
        n = si["unique_dofs"]
        fd = p.sub_dof_rhs(f)  # collective
        dist.barrier()
        if rank != 0:
            # the other ranks only serve the collectives above
            p.close()
            continue
        pd = p.sub_point_dofs()
        cls = np.full(n, "", dtype=object)
        if bl:
            cls[:] = "own"
        else:
            ids, lv = p.sub_region()
            own_pts = si["own_points"]
            n_own = int(pd[:own_pts].max()) + 1
            cls[:si["sub_dofs"] - si["interface_dofs"]] = "ring, lower degree"
            # dofs touched by degree-N ring points
            degs = S.level_degrees(N, red)
            off = 0
            for k, l in enumerate(lv):
                npts = (degs[l] + 1) ** 3
                if l == 0:
                    d = pd[off:off + npts]
                    d = d[(d >= 0) & (d < n)]
                    cls[d] = "ring, degree N"
                off += npts
                if off >= len(pd):
                    break
            cls[:n_own] = "own"
            cls[si["sub_dofs"] - si["interface_dofs"]:si["sub_dofs"]] = "interface"
            cls[si["sub_dofs"]:] = "superdomain"
        # 1. the diagonal, by probing
        diag = np.zeros(n)
        x = np.zeros(n)
        for d in range(n):
            x[d] = 1.0
            diag[d] = p.sub_dof_operator(x)[d]
            x[d] = 0.0
        dj = p.sub_jacobi_diagonal()  # the product's setup-time diagonal (exact, element by element) against the probe
        entry_jacobi_error = float(np.abs(dj - diag).max() / np.abs(diag).max())
        classes = [c for c in ("own", "ring, degree N", "ring, lower degree", "interface", "superdomain") if (cls == c).any()]
        entry = {"dofs": n, "jacobi_diagonal_vs_probe": entry_jacobi_error, "classes": {}}
        for c in classes:
            m = cls == c
            entry["classes"][c] = {"count": int(m.sum()), "diag_min": float(diag[m].min()), "diag_median": float(np.median(diag[m])), "diag_max": float(diag[m].max()), "rhs_rms": float(np.sqrt(np.mean(fd[m] ** 2)))}

        # 2. four steps of GMRES on the first right-hand side
        def report(r, label):
            out = {"relative_residual": float(np.linalg.norm(r) / np.linalg.norm(fd))}
            for c in classes:
                m = cls == c
                out[c] = float(np.linalg.norm(r[m]) / max(np.linalg.norm(fd[m]), 1e-300))
            entry[label] = out

        op = p.sub_dof_operator
        x4, r4 = gmres(op, fd, 4)
        report(r4, "gmres4")
        s = 1.0 / np.sqrt(diag)
        xs, _ = gmres(lambda v: s * op(s * v), s * fd, 4)
        report(fd - op(s * xs), "gmres4, symmetric diagonal scaling")
        xr, _ = gmres(lambda v: op(v / diag), fd, 4)
        report(fd - op(xr / diag), "gmres4, point-Jacobi in the preconditioner slot")
        # how good is the correction where it is used: the own dofs, against a converged solve of the same system
        xe, re = gmres(lambda v: s * op(s * v), s * fd, 400)
        xe = s * xe
        entry["converged_solve_residual"] = float(np.linalg.norm(fd - op(xe)) / np.linalg.norm(fd))
        own = cls == "own"
        for label, xx in (("gmres4", x4), ("gmres4, symmetric diagonal scaling", s * xs), ("gmres4, point-Jacobi in the preconditioner slot", xr / diag)):
            entry[label]["own_dof_error_vs_converged"] = float(np.linalg.norm((xx - xe)[own]) / np.linalg.norm(xe[own]))
        res[name] = entry
        p.close()
    if rank == 0:
        with open(out_file, "w") as fh:
            json.dump(res, fh, indent=1)
        print(json.dumps(res, indent=1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    e = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    out = sys.argv[4] if len(sys.argv) > 4 else "/tmp/composite_operator_analysis.json"
    import rendezvous

    port = rendezvous.new()
    mp.spawn(worker, args=(world, port, e, N, 6 if N == 7 else 2, out), nprocs=world, join=True)
