#!/usr/bin/env python3
"""Per-kernel achieved-bandwidth microbenchmark at BASELINE config C2 sizes
(32^3 elements, N=7: P = 16 777 216 points, 225^3 nodes).  Development tool:
prints one line per kernel with algorithmic GB/s (BASELINE.md section 4
byte counts) against the 8 TB/s HBM3E peak.

    python tools/microbench.py [--quick] [--only NAME]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib  # noqa: E402
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd.kernels import k, reduce_workspace  # noqa: E402

PEAK = 8.0e12


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def report(name, nbytes, t, results):
    gbs = nbytes / t / 1e9
    print(f"{name:42s} {t*1e6:10.1f} us  {gbs:9.1f} GB/s  {100*gbs*1e9/PEAK:5.1f}% of 8 TB/s", flush=True)
    results[name] = {"us": t * 1e6, "GBps": gbs, "frac_peak": gbs * 1e9 / PEAK, "bytes": nbytes}


def box_Q(E, N, dev):
    """Boolean scatter Q (points x nodes) and gather Qt of a single-rank box,
    local node numbering in first-encounter order (domain.tpp:249-281)."""
    n = N + 1
    G = E * N + 1
    e = torch.arange(E, device=dev)
    i = torch.arange(n, device=dev)
    # element-major, x fastest: glo[ez,ey,ex,k,j,i]
    gx = (e[:, None] * N + i[None, :])  # [E, n]
    gi = gx[None, None, :, None, None, :]
    gj = gx[None, :, None, None, :, None]
    gk = gx[:, None, None, :, None, None]
    glo = (gi + G * (gj + G * gk)).reshape(-1)
    uniq, inv = torch.unique(glo, return_inverse=True)
    # first occurrence of each unique id
    P = glo.numel()
    first = torch.full((uniq.numel(),), P, dtype=torch.int64, device=dev)
    first.scatter_reduce_(0, inv, torch.arange(P, device=dev), reduce="amin")
    order = torch.argsort(first)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=dev)
    col = rank[inv].to(torch.int32)
    nodes = uniq.numel()
    Q = (torch.arange(P + 1, dtype=torch.int32, device=dev), col, torch.ones(P, dtype=torch.float64, device=dev))
    # transpose
    perm = torch.argsort(col.to(torch.int64), stable=True)
    counts = torch.bincount(col.to(torch.int64), minlength=nodes)
    tptr = torch.zeros(nodes + 1, dtype=torch.int32, device=dev)
    tptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    Qt = (tptr, perm.to(torch.int32), torch.ones(P, dtype=torch.float64, device=dev))
    return Q, Qt, P, nodes


def stencil27(m, dev):
    idx = torch.arange(m**3, device=dev)
    z, y, x = idx // (m * m), (idx // m) % m, idx % m
    cols, valid = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (z + dz >= 0) & (z + dz < m) & (y + dy >= 0) & (y + dy < m) & (x + dx >= 0) & (x + dx < m)
                cols.append((idx + dz * m * m + dy * m + dx).to(torch.int32))
                valid.append(ok)
    cols = torch.stack(cols, 1)
    valid = torch.stack(valid, 1)
    counts = valid.sum(1)
    ptr = torch.zeros(m**3 + 1, dtype=torch.int32, device=dev)
    ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    col = cols[valid].contiguous()
    del cols, valid
    val = torch.rand(col.numel(), dtype=torch.float64, device=dev) - 0.5
    return ptr, col, val


def make_plan(ptr_dev, rows, cols, nnz, unit_values=False):
    ptr_h = ptr_dev.cpu().numpy()
    plan = ctypes.c_void_p()
    lib.hip().call("fdd_csr_plan_create", ctypes.byref(plan), ctypes.c_void_p(ptr_h.ctypes.data), rows, cols, nnz)
    if unit_values:
        lib.hip().call("fdd_csr_plan_set_unit_values", plan, 1)
    return plan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--E", type=int, default=32)
    ap.add_argument("--N", type=int, default=7)
    ap.add_argument("--stencil", type=int, default=225)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = lib.hip()
    E, N = (16, 7) if args.quick else (args.E, args.N)
    n = N + 1
    P = E**3 * n**3
    results = {}

    def want(name):
        return args.only is None or args.only in name

    vecs = [torch.rand(P, dtype=torch.float64, device=dev) for _ in range(6)]
    a, b, c, d, e_, f = vecs
    ws = reduce_workspace(dev)
    out = torch.zeros(2, dtype=torch.float64, device=dev)
    out8 = torch.zeros(16, dtype=torch.float64, device=dev)

    if want("blas1"):
        report("blas1.set_to_value", 8 * P, timeit(lambda: k("fdd_set_to_value", a, 1.0, P, 0)), results)
        report("blas1.vector_scaling", 16 * P, timeit(lambda: k("fdd_vector_scaling", a, 0.5, b, P)), results)
        report("blas1.vector_vector_addition", 24 * P, timeit(lambda: k("fdd_vector_vector_addition", a, 0.5, b, 0.25, c, P)), results)
        report("dom.initialize_arrays", 24 * P, timeit(lambda: k("fdd_dom_initialize_arrays", a, b, c, P)), results)
        report("dom.solution_and_residual_update", 48 * P, timeit(lambda: k("fdd_dom_solution_and_residual_update", a, b, c, d, e_, 1e-3, P)), results)
        report("dom.residual_and_search_update", 40 * P, timeit(lambda: k("fdd_dom_residual_and_search_update", a, b, c, d, 1e-3, P)), results)
    if want("dot"):
        report("dot.sub_inner_product", 16 * P, timeit(lambda: k("fdd_sub_inner_product", out, ws, a, b, P)), results)
        report("dot.dom_residual_norm", 24 * P, timeit(lambda: k("fdd_dom_residual_norm", out, ws, a, b, c, P)), results)
        report("dot.dom_projection_inner_products", 32 * P, timeit(lambda: k("fdd_dom_projection_inner_products", out, ws, a, b, c, d, P)), results)

    if want("streams"):
        # what plain streaming reaches with as many concurrent streams as the stiffness kernel has (7 reads + 1 write)
        extra = [torch.rand(P, dtype=torch.float64, device=dev) for _ in range(3)]
        vs = [b, c, d, e_, f] + extra
        for m in (1, 2, 4, 6, 8):
            report(f"streams.multi_dot {m + 2}R", 8 * P * (m + 2), timeit(lambda: k("fdd_multi_weighted_inner_product", out8, ws, a, vs[:m], m, vs[-1], P)), results)
        for m in (1, 2, 4, 6):
            coef = (ctypes.c_double * m)(*([1e-3] * m))
            report(f"streams.multi_axpy {m + 1}R+1W", 8 * P * (m + 2), timeit(lambda: k("fdd_multi_axpy", a, coef, vs[:m], m, P)), results)
        del extra, vs

    if want("stiffness"):
        _, w, _ = gll(N)
        Dh = torch.tensor(gll(N)[2], dtype=torch.float64, device=dev)
        G = [torch.rand(P, dtype=torch.float64, device=dev) for _ in range(6)]
        u = a
        Au = b
        report(f"stiffness.fused N={N}", 64 * P, timeit(lambda: k("fdd_dom_stiffness_matrix", Au, u, Dh, G, E**3, N)), results)
        (_, qc0, _), _, _, nodes0 = box_Q(E, N, dev)
        v_nodes = torch.rand(nodes0, dtype=torch.float64, device=dev)
        report(f"stiffness.fused+gather N={N}", 60 * P + 8 * nodes0, timeit(lambda: k("fdd_sub_stiffness_matrix_gather", Au, v_nodes, qc0, Dh, G, None, E**3, N)), results)
        del qc0
        if N >= 8:
            report(f"stiffness.mfma_f64 N={N}", 64 * P, timeit(lambda: k("fdd_stiffness_matrix_mfma", Au, u, Dh, G, None, E**3, N)), results)
        # affine elements: six numbers per element + the GLL weights instead of six streamed arrays (16 B/pt; with the gather 12 B/pt + the dofs)
        Gc = torch.rand(E**3 * 6, dtype=torch.float64, device=dev)
        wg = torch.tensor(w, dtype=torch.float64, device=dev)
        report(f"stiffness.affine N={N}", 16 * P, timeit(lambda: k("fdd_stiffness_matrix_affine", Au, u, None, None, Dh, Gc, wg, None, E**3, N)), results)
        (_, qc1, _), _, _, nodes1 = box_Q(E, N, dev)
        v1 = torch.rand(nodes1, dtype=torch.float64, device=dev)
        report(f"stiffness.affine+gather N={N}", 12 * P + 8 * nodes1, timeit(lambda: k("fdd_stiffness_matrix_affine", Au, v1, None, qc1, Dh, Gc, wg, None, E**3, N)), results)
        if N >= 8:
            report(f"stiffness.mfma_affine N={N}", 16 * P, timeit(lambda: k("fdd_stiffness_matrix_mfma_affine", Au, u, None, None, Dh, Gc, wg, None, E**3, N)), results)
            report(f"stiffness.mfma_affine+gather N={N}", 12 * P + 8 * nodes1, timeit(lambda: k("fdd_stiffness_matrix_mfma_affine", Au, v1, None, qc1, Dh, Gc, wg, None, E**3, N)), results)
        del qc1, v1
        GDu = [c, d, e_]

        def two():
            k("fdd_dom_stiffness_matrix_1", GDu, u, Dh, G, P, N, 3)
            k("fdd_dom_stiffness_matrix_2", Au, GDu, Dh, P, N, 3)

        report(f"stiffness.two_launch N={N} (112 B/pt)", 112 * P, timeit(two, iters=5), results)
        # the same buffers read as 2-D elements of (N+1)^2 points: fused 2-D kernel (40 B/pt) against its two-launch form (72 B/pt)
        E2 = P // (n * n)
        report(f"stiffness.fused_2d N={N}", 40 * P, timeit(lambda: k("fdd_stiffness_matrix_2d", Au, u, Dh, G, None, E2, N)), results)

        def two_2d():
            k("fdd_dom_stiffness_matrix_1", GDu, u, Dh, G, P, N, 2)
            k("fdd_dom_stiffness_matrix_2", Au, GDu, Dh, P, N, 2)

        report(f"stiffness.two_launch_2d N={N} (72 B/pt)", 72 * P, timeit(two_2d, iters=5), results)
        del G

    if want("restriction"):
        # the degree tree of tree_operator (subdomain.tpp:4593-4607): degree N elements restricted to degree 1 and N-2
        from_nodes = np.asarray(gll(N)[0])
        for Nc in (1, max(N - 2, 1)):
            to_nodes = np.asarray(gll(Nc)[0])
            # J_cf[i + l*n_c]: value at fine node l of the coarse Lagrange polynomial i (tests/support.py:J_cf)
            J = np.zeros((N + 1, Nc + 1))
            for i in range(Nc + 1):
                for l in range(N + 1):
                    J[l, i] = np.prod([(from_nodes[l] - to_nodes[m]) / (to_nodes[i] - to_nodes[m]) for m in range(Nc + 1) if m != i])
            Jd = torch.tensor(J.reshape(-1), dtype=torch.float64, device=dev)
            uc = torch.empty(E**3 * (Nc + 1) ** 3, dtype=torch.float64, device=dev)
            report(f"restriction.fused N={N}->{Nc}", 8 * (P + uc.numel()), timeit(lambda: k("fdd_sub_restriction", uc, Jd, a, E**3, N + 1, Nc + 1)), results)

    if want("csr"):
        (qp, qc, qv), (tp, tc, tv), Pq, nodes = box_Q(E, N, dev)
        x_nodes = torch.rand(nodes, dtype=torch.float64, device=dev)
        y_pts = torch.empty(Pq, dtype=torch.float64, device=dev)
        bytes_Q = 12 * Pq + 12 * Pq + 8 * nodes
        bytes_Qt = 12 * Pq + 12 * nodes + 8 * Pq
        planQ = make_plan(qp, Pq, nodes, Pq)
        planQt = make_plan(tp, nodes, Pq, Pq)
        planQu = make_plan(qp, Pq, nodes, Pq, unit_values=True)
        planQtu = make_plan(tp, nodes, Pq, Pq, unit_values=True)
        report("csr.Q multiply, unit-value plan", bytes_Q, timeit(lambda: k("fdd_csr_plan_multiply", planQu, y_pts, qp, qc, qv, x_nodes, None)), results)
        report("csr.Q multiply_weight, unit-value plan", bytes_Q + 8 * Pq, timeit(lambda: k("fdd_csr_plan_multiply", planQu, y_pts, qp, qc, qv, x_nodes, a)), results)
        report("csr.Qt multiply, unit-value plan", bytes_Qt, timeit(lambda: k("fdd_csr_plan_multiply", planQtu, x_nodes, tp, tc, tv, y_pts, None)), results)
        report("csr.Q multiply (scatter, 1 nnz/row)", bytes_Q, timeit(lambda: k("fdd_csr_plan_multiply", planQ, y_pts, qp, qc, qv, x_nodes, None)), results)
        report("csr.Q multiply_weight", bytes_Q + 8 * Pq, timeit(lambda: k("fdd_csr_plan_multiply", planQ, y_pts, qp, qc, qv, x_nodes, a)), results)
        report("csr.Qt multiply (gather, 1-8 nnz/row)", bytes_Qt, timeit(lambda: k("fdd_csr_plan_multiply", planQt, x_nodes, tp, tc, tv, y_pts, None)), results)
        # fused gather-scatter dssum and the gather passes of the restructured inner solve
        wn = torch.rand(nodes, dtype=torch.float64, device=dev)
        b_ds = 4.0 * nodes + 20.0 * Pq
        report("dssum.fused (plain)", b_ds, timeit(lambda: k("fdd_dssum_fused", y_pts, None, tp, tc, b, None, None, 0, nodes)), results)
        report("dssum.fused (weight+mask)", b_ds + 8.0 * Pq + 8.0 * nodes, timeit(lambda: k("fdd_dssum_fused", y_pts, None, tp, tc, b, wn, a, 0, nodes)), results)
        report("dssum.fused in place (weight+mask)", b_ds + 8.0 * Pq + 8.0 * nodes, timeit(lambda: k("fdd_dssum_fused", b, None, tp, tc, b, wn, a, 0, nodes)), results)
        report("dssum.gather (weight)", 4.0 * nodes + 12.0 * Pq + 16.0 * nodes, timeit(lambda: k("fdd_dssum_gather", x_nodes, tp, tc, b, wn, 0, nodes)), results)
        report("dssum.gather_weighted_norm2", 4.0 * nodes + 12.0 * Pq + 8.0 * nodes, timeit(lambda: k("fdd_gather_weighted_norm2", out, ws, tp, tc, b, wn, nodes)), results)
        report("dssum[plan].fused (plain)", b_ds, timeit(lambda: k("fdd_csr_plan_dssum", planQtu, y_pts, None, tp, tc, b, None, None, 0, nodes, 0)), results)
        report("dssum[plan].fused (weight+mask)", b_ds + 8.0 * Pq + 8.0 * nodes, timeit(lambda: k("fdd_csr_plan_dssum", planQtu, y_pts, None, tp, tc, b, wn, a, 0, nodes, 0)), results)
        report("dssum[plan].fused in place (weight+mask)", b_ds + 8.0 * Pq + 8.0 * nodes, timeit(lambda: k("fdd_csr_plan_dssum", planQtu, b, None, tp, tc, b, wn, a, 0, nodes, 0)), results)
        report("dssum[plan].gather (weight)", 4.0 * nodes + 12.0 * Pq + 16.0 * nodes, timeit(lambda: k("fdd_csr_plan_dssum", planQtu, None, x_nodes, tp, tc, b, wn, None, 0, nodes, 1)), results)
        report("dssum[plan].gather_weighted_norm2", 4.0 * nodes + 12.0 * Pq + 8.0 * nodes, timeit(lambda: k("fdd_csr_plan_gather_weighted_norm2", planQtu, out, ws, tp, tc, b, wn)), results)
        del qp, qc, qv, tp, tc, tv
        m = 97 if args.quick else args.stencil
        sp, sc, sv = stencil27(m, dev)
        nnz = sc.numel()
        rows = m**3
        xs = torch.rand(rows, dtype=torch.float64, device=dev)
        ys = torch.empty(rows, dtype=torch.float64, device=dev)
        bytes_S = 12 * nnz + 12 * rows + 8 * rows
        planS = make_plan(sp, rows, rows, nnz)
        kind = ctypes.c_int()
        L.call("fdd_csr_plan_kind", planS, ctypes.byref(kind))
        report(f"csr.stencil27 {m}^3 planned(kind={kind.value})", bytes_S, timeit(lambda: k("fdd_csr_plan_multiply", planS, ys, sp, sc, sv, xs, None), iters=10), results)
        report(f"csr.stencil27 {m}^3 thread-per-row", bytes_S, timeit(lambda: k("fdd_csr_multiply", ys, sp, sc, sv, xs, rows), iters=5), results)

    if args.json:
        with open(args.json, "w") as fh:
            json.dump(results, fh, indent=1)


def gll(N):
    with open(os.path.join(ROOT, "tests", "golden", "gll_tables.json")) as fh:
        t = json.load(fh)["levels"][str(N)]
    return np.array(t["z"]), np.array(t["w"]), np.array(t["D_hat"])


if __name__ == "__main__":
    main()
