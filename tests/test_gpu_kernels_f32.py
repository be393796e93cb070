"""GPU parity of the single-precision inner solve, kernel by kernel (the reference's PTYPE = Float = float,
config.hpp:19-20, poisson.cpp:206): every `_f32` entry of libfdd_hip.so that the float preconditioner launches,
called through the C-ABI, against its IEEE-single twin in oracle/fdd_oracle_f32.c on the same seeded inputs.

Bar: where the operation order is defined (element stiffness, axpby, scaling, gathers, index copies, the solution
update) the float results must be BIT-IDENTICAL to the oracle's float arithmetic (both sides -ffp-contract=off).
The two reductions carry double accumulators over float data and sum in a different tree:
|gpu - oracle| <= 1e-13 * sum|terms| for the scalars; the float vector the fused axpy+norm stores is again bit-exact.
"""
import ctypes

import numpy as np
import pytest
import torch

import support as S
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd.kernels import k, reduce_workspace

pytestmark = pytest.mark.gpu

vp = ctypes.c_void_p
f32 = np.float32
SIZES = [1, 2, 3, 127, 128, 129, 4097, 1000003]


def P(a):
    return vp(a.ctypes.data)


def ptrs(arrays):
    return (vp * len(arrays))(*[a.ctypes.data for a in arrays])


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def rnd32(n, seed):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n).astype(f32)


@pytest.mark.parametrize("N", [1, 2, 3, 4, 6, 7, 9, 15])
def test_stiffness_f32_gather_scaled_bit_exact(gpu, N):
    """fdd_sub_stiffness_matrix_gather_scaled_f32 (subdomain.okl:4-101 with DType = float, the scatter Q and the
    vector scaling fused into the load): contiguous and offset-list element order, with and without the scale,
    points without a dof."""
    L = S.oracle()
    n3 = (N + 1) ** 3
    D = S.gll(N)[2].astype(f32)
    for E in (1, 6, 41):
        rng = np.random.default_rng(700 + 13 * N + E)
        G = [rng.uniform(0.5, 1.5, E * n3).astype(f32) if g < 3 else rng.uniform(-0.2, 0.2, E * n3).astype(f32) for g in range(6)]
        ndof = max(1, (E * n3) // 3)
        pd = rng.integers(-1, ndof, E * n3).astype(np.int32)
        v = rng.uniform(-1, 1, ndof).astype(f32)
        dG = [dev(g, gpu) for g in G]
        for scale in (None, 0.37251):
            ref = np.zeros(E * n3, f32)
            sc32 = None if scale is None else np.array([scale], np.float64).astype(f32)
            L.orc_f32_sub_stiffness(P(ref), P(v), P(pd), None if sc32 is None else P(sc32), P(D), ptrs(G), E, N)
            dsc = None if scale is None else dev(np.array([scale]), gpu)
            out = torch.full((E * n3,), 3.0, dtype=torch.float32, device=gpu)
            k("fdd_sub_stiffness_matrix_gather_scaled_f32", out, dev(v, gpu), dsc, dev(pd, gpu), dev(D, gpu), dG, None, E, N)
            assert np.array_equal(host(out), ref), (N, E, scale)
            eo = (np.arange(E)[::-1] * n3).astype(np.int32)
            out2 = torch.full((E * n3,), 5.0, dtype=torch.float32, device=gpu)
            k("fdd_sub_stiffness_matrix_gather_scaled_f32", out2, dev(v, gpu), dsc, dev(pd, gpu), dev(D, gpu), dG, dev(eo, gpu), E, N)
            assert np.array_equal(host(out2), ref), (N, E, scale, "offset list")


@pytest.mark.parametrize("n", SIZES)
def test_axpby_and_scaling_f32(gpu, n):
    L = S.oracle()
    u, v = rnd32(n, 1), rnd32(n, 2)
    a, b = f32(1.25), f32(-0.7)
    ref = np.zeros(n, f32)
    L.orc_f32_vector_vector_addition(P(ref), ctypes.c_float(a), P(u), ctypes.c_float(b), P(v), n)
    du, dv = dev(u, gpu), dev(v, gpu)
    out = torch.zeros(n, dtype=torch.float32, device=gpu)
    k("fdd_vector_vector_addition_f32", out, float(a), du, float(b), dv, n)
    assert np.array_equal(host(out), ref)
    k("fdd_vector_vector_addition_f32", du, float(a), du, float(b), dv, n)  # aliased with the first operand (r = f - q in place)
    assert np.array_equal(host(du), ref)

    s = 1.0 / 3.0
    L.orc_f32_vector_scaling(P(ref), ctypes.c_double(s), P(u), n)
    out = torch.zeros(n, dtype=torch.float32, device=gpu)
    k("fdd_vector_scaling_dev_f32", out, dev(np.array([s]), gpu), dev(u, gpu), n)
    assert np.array_equal(host(out), ref)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("m", [1, 3, 5])
def test_multi_inner_product_and_axpy_norm_f32(gpu, n, m):
    """The Gram-Schmidt pair of the float inner GMRES: out[k] = sum a (s_k b_k) and dst = y - sum c_k (s_k x_k),
    |dst|^2 -- float data, double accumulators."""
    L = S.oracle()
    ws = reduce_workspace(gpu)
    a = rnd32(n, 10)
    b = [rnd32(n, 20 + i) for i in range(m)]
    b[0] = a  # the self-dot of the residual reads one stream
    scales = np.random.default_rng(5).uniform(0.5, 2.0, 8)
    da = dev(a, gpu)
    db = [da] + [dev(x, gpu) for x in b[1:]]
    dsc = dev(scales, gpu)
    for with_scale in (False, True):
        ref = np.zeros(m)
        L.orc_f32_multi_inner_product_scaled(P(ref), P(a), ptrs(b), P(scales) if with_scale else None, m, n)
        out = torch.zeros(8, dtype=torch.float64, device=gpu)
        k("fdd_multi_inner_product_scaled_f32", out, ws, da, db, dsc if with_scale else None, m, n)
        got = host(out)[:m]
        for i in range(m):
            bound = 1e-12 * np.sum(np.abs(a.astype(np.float64) * b[i].astype(np.float64))) * (scales[i] if with_scale else 1.0)
            assert abs(got[i] - ref[i]) <= bound, (i, got[i], ref[i])

    c = np.random.default_rng(6).uniform(-1, 1, 8)
    y = rnd32(n, 40)
    x = [rnd32(n, 50 + i) for i in range(m)]
    dx = [dev(v, gpu) for v in x]
    for with_scale in (False, True):
        ref_dst = np.zeros(n, f32)
        L.orc_f32_multi_axpy_norm2_scaled.restype = ctypes.c_double
        ref_norm = L.orc_f32_multi_axpy_norm2_scaled(P(ref_dst), P(y), P(c), ctypes.c_double(-1.0), ptrs(x), P(scales) if with_scale else None, m, n)
        dst = torch.zeros(n, dtype=torch.float32, device=gpu)
        out = torch.zeros(1, dtype=torch.float64, device=gpu)
        k("fdd_multi_axpy_norm2_scaled_dev_f32", out, ws, dst, dev(y, gpu), dev(c, gpu), -1.0, dx, dsc if with_scale else None, m, n)
        assert np.array_equal(host(dst), ref_dst)  # the stored float vector: same double sum, one rounding
        assert abs(host(out)[0] - ref_norm) <= 1e-12 * ref_norm + 1e-300  # a serial sum on one side, a tree on the other


@pytest.mark.parametrize("n", [1, 129, 4097, 1000003])
def test_multi_lincomb_f32(gpu, n):
    """The solution update u~ (+)= sum_{k <= last} y_k (s_k v_k) of the float inner GMRES, column count on the device."""
    L = S.oracle()
    m = 4
    c = np.random.default_rng(7).uniform(-1, 1, 8)
    scales = np.random.default_rng(8).uniform(0.5, 2.0, 8)
    v = [rnd32(n, 60 + i) for i in range(m)]
    dv = [dev(x, gpu) for x in v]
    q0 = rnd32(n, 70)
    for q_is_zero in (1, 0):
        for last in (None, 0, 2, 7):
            for with_scale in (False, True):
                ref = q0.copy()
                L.orc_f32_multi_lincomb(P(ref), q_is_zero, P(c), ptrs(v), P(scales) if with_scale else None, -1 if last is None else last, m, n)
                q = dev(q0, gpu)
                dlast = None if last is None else dev(np.array([float(last)]), gpu)
                k("fdd_multi_lincomb_limited_dev_f32", q, q_is_zero, dev(c, gpu), dv, dev(scales, gpu) if with_scale else None, dlast, m, n)
                assert np.array_equal(host(q), ref), (q_is_zero, last, with_scale)


def test_boolean_gathers_f32(gpu):
    """Qt of the float solve: fdd_gather_rows_f32 (one lane per row) and fdd_csr_plan_gather_f32 (row blocks through
    LDS) against csr_matrix.okl:5-18 in float; full and partial row ranges, empty rows, rows of up to 8 entries."""
    L = S.oracle()
    rng = np.random.default_rng(11)
    rows, cols = 50021, 131071
    counts = rng.integers(0, 9, rows)
    counts[rng.integers(0, rows, 200)] = 0
    ptr = np.zeros(rows + 1, np.int32)
    ptr[1:] = np.cumsum(counts)
    col = rng.integers(0, cols, ptr[-1]).astype(np.int32)
    u = rnd32(cols, 12)
    dptr, dcol, du = dev(ptr, gpu), dev(col, gpu), dev(u, gpu)
    plan = vp()
    lib.hip().call("fdd_csr_plan_create", ctypes.byref(plan), vp(ptr.ctypes.data), rows, cols, int(ptr[-1]))
    try:
        for lo, hi in ((0, rows), (17, 40001), (rows - 3, rows)):
            ref = np.full(rows, 9.0, f32)
            L.orc_f32_csr_gather(P(ref), P(ptr), P(col), P(u), lo, hi)
            t = torch.full((rows,), 9.0, dtype=torch.float32, device=gpu)
            k("fdd_gather_rows_f32", t, dptr, dcol, du, lo, hi)
            assert np.array_equal(host(t), ref), (lo, hi)
            t2 = torch.full((rows,), 9.0, dtype=torch.float32, device=gpu)
            k("fdd_csr_plan_gather_f32", plan, t2, dptr, dcol, du, lo, hi)
            assert np.array_equal(host(t2), ref), ("plan", lo, hi)
    finally:
        lib.hip().call("fdd_csr_plan_destroy", plan)


@pytest.mark.parametrize("n", [1, 129, 1000003])
def test_gather_indexed_f32(gpu, n):
    L = S.oracle()
    src = rnd32(max(n // 2, 1), 13)
    idx = np.random.default_rng(14).integers(-1, len(src), n).astype(np.int32)
    ref = np.zeros(n, f32)
    L.orc_f32_gather_indexed(P(ref), P(src), P(idx), n)
    out = torch.full((n,), 2.0, dtype=torch.float32, device=gpu)
    k("fdd_gather_indexed_f32", out, dev(src, gpu), dev(idx, gpu), n)
    assert np.array_equal(host(out), ref)
    ref64 = np.zeros(n)
    L.orc_f32_gather_indexed_f64(P(ref64), P(src), P(idx), n)
    out64 = torch.full((n,), 2.0, dtype=torch.float64, device=gpu)
    k("fdd_gather_indexed_f32_f64", out64, dev(src, gpu), dev(idx, gpu), n)
    assert np.array_equal(host(out64), ref64)


@pytest.mark.parametrize("n", [1, 3, 129, 4097, 1000003])
def test_diagonal_scaling(gpu, n):
    """z = d .* ((*scale) * u), the point-Jacobi option's kernel, double and float: the two statements it stands for
    (math.okl:29-35 vector_scaling, then AMG/kernels.cu:64-73 vector_multiplication) in that order, bit for bit."""
    L = S.oracle()
    d, u = np.random.default_rng(1).uniform(0.5, 2.0, n), np.random.default_rng(2).uniform(-1, 1, n)
    s = 0.7312
    tmp, ref = np.zeros(n), np.zeros(n)
    L.orc_vector_scaling(P(tmp), ctypes.c_double(s), P(u), n)
    L.orc_amg_vector_multiplication(P(ref), P(d), P(tmp), n)
    z = torch.zeros(n, dtype=torch.float64, device=gpu)
    k("fdd_vector_diagonal_scaling_dev", z, dev(d, gpu), dev(np.array([s]), gpu), dev(u, gpu), n)
    assert np.array_equal(host(z), ref)
    L.orc_amg_vector_multiplication(P(ref), P(d), P(u), n)
    k("fdd_vector_diagonal_scaling_dev", z, dev(d, gpu), None, dev(u, gpu), n)
    assert np.array_equal(host(z), ref)
    d32, u32 = d.astype(f32), u.astype(f32)
    t32 = np.zeros(n, f32)
    L.orc_f32_vector_scaling(P(t32), ctypes.c_double(s), P(u32), n)
    ref32 = (d32 * t32).astype(f32)
    z32 = torch.zeros(n, dtype=torch.float32, device=gpu)
    k("fdd_vector_diagonal_scaling_dev_f32", z32, dev(d32, gpu), dev(np.array([s]), gpu), dev(u32, gpu), n)
    assert np.array_equal(host(z32), ref32)
    k("fdd_vector_diagonal_scaling_dev_f32", z32, dev(d32, gpu), None, dev(u32, gpu), n)
    assert np.array_equal(host(z32), (d32 * u32).astype(f32))


@pytest.mark.parametrize("N", [1, 3, 7, 9])
def test_stiffness_affine_f32_equals_the_streamed_kernel(gpu, N):
    """fdd_stiffness_matrix_affine_f32 (the factors of a point formed from six floats per element and the float GLL
    weights) on factor arrays of exactly that form == the float oracle twin of the streamed kernel, bit for bit."""
    L = S.oracle()
    n = N + 1
    n3 = n**3
    D = S.gll(N)[2].astype(f32)
    w = S.gll(N)[1].astype(f32)
    W = ((w[None, None, :] * w[None, :, None]).astype(f32) * w[:, None, None]).astype(f32).reshape(1, -1)
    for E in (1, 6, 41):
        rng = np.random.default_rng(950 + 13 * N + E)
        c = np.concatenate([rng.uniform(0.5, 1.5, (E, 3)), rng.uniform(-0.2, 0.2, (E, 3))], axis=1).astype(f32)
        G = [np.ascontiguousarray((c[:, g, None] * W).astype(f32).ravel()) for g in range(6)]
        ndof = max(1, (E * n3) // 3)
        pd = rng.integers(-1, ndof, E * n3).astype(np.int32)
        v = rng.uniform(-1, 1, ndof).astype(f32)
        for scale in (None, 0.37251):
            ref = np.zeros(E * n3, f32)
            sc32 = None if scale is None else np.array([scale], np.float64).astype(f32)
            L.orc_f32_sub_stiffness(P(ref), P(v), P(pd), None if sc32 is None else P(sc32), P(D), ptrs(G), E, N)
            dsc = None if scale is None else dev(np.array([scale]), gpu)
            out = torch.full((E * n3,), 3.0, dtype=torch.float32, device=gpu)
            k("fdd_stiffness_matrix_affine_f32", out, dev(v, gpu), dsc, dev(pd, gpu), dev(D, gpu), dev(c.ravel(), gpu), dev(w, gpu), None, E, N)
            assert np.array_equal(host(out), ref), (N, E, scale)
