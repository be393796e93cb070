"""CPU tests of the oracle itself (no GPU): the reference-pinned GLL tables,
the C restatement against independent dense numpy statements of the same
operators, the solvers against the manufactured solution, multi-rank
equivalence, and the committed golden fixture."""
import ctypes
import hashlib
import json
import math
import os

import numpy as np
import pytest

import support as S

P = S._p
vp = ctypes.c_void_p


# ------------------------------------------------------------------ GLL tables
def test_golden_gll_tables_match_reference_speclib():
    """tests/golden/gll_tables.json is what the reference's own
    special_functions.f returns (compiled by oracle/Makefile `ref`).  Runs where
    oracle/_ref exists (build container; the .so also travels to the GPU box)."""
    if not os.path.exists(S.SPECLIB_REF_SO):
        pytest.skip("oracle/_ref/libspeclib_ref.so not built (needs /root/reference)")
    L = ctypes.CDLL(S.SPECLIB_REF_SO)
    L.hgll_.restype = ctypes.c_double
    for N in range(1, 16):
        n = N + 1
        z, w, Dh = S.gll(N)
        zz, ww = np.zeros(n), np.zeros(n)
        cn = ctypes.c_int(n)
        L.zwgll_(P(zz), P(ww), ctypes.byref(cn))
        Dt, D = np.zeros(n * n), np.zeros(n * n)
        L.dgll_(P(Dt), P(D), P(zz), ctypes.byref(cn), ctypes.byref(cn))
        assert np.array_equal(z, zz) and np.array_equal(w, ww) and np.array_equal(Dh, D)
    zc = np.ascontiguousarray(S.gll(3)[0])
    zf = S.gll(7)[0]
    J = S.J_cf(3, 7).reshape(8, 4)
    for i in range(8):
        for j in range(1, 5):
            x = ctypes.c_double(zf[i])
            v = L.hgll_(ctypes.byref(ctypes.c_int(j)), ctypes.byref(x), P(zc), ctypes.byref(ctypes.c_int(4)))
            assert v == J[i, j - 1]


@pytest.mark.parametrize("N", range(1, 16))
def test_gll_table_properties(N):
    """Reference-independent sanity of the tables: quadrature and
    differentiation exactness on polynomials."""
    z, w, D = S.gll(N)
    n = N + 1
    D = D.reshape(n, n)
    assert abs(w.sum() - 2.0) < 1e-13
    assert np.allclose(z, -z[::-1], atol=1e-14) and z[0] == -1.0 and z[-1] == 1.0
    # GLL quadrature integrates x^(2N-1) exactly
    for k in range(0, 2 * N):
        exact = 0.0 if k % 2 else 2.0 / (k + 1)
        assert abs((w * z**k).sum() - exact) < 1e-12
    # D differentiates polynomials of degree <= N exactly: D[i, j] = l_j'(z_i)
    for k in range(0, N + 1):
        d = D @ z**k
        ref = k * z ** max(k - 1, 0) if k else np.zeros(n)
        assert np.abs(d - ref).max() < 1e-10 * max(1, N**2)


@pytest.mark.parametrize("Nc,Nf", [(1, 7), (3, 7), (5, 7), (1, 3), (9, 15), (1, 2)])
def test_interpolator_properties(Nc, Nf):
    J = S.J_cf(Nc, Nf).reshape(Nf + 1, Nc + 1)
    zc, zf = S.gll(Nc)[0], S.gll(Nf)[0]
    assert np.abs(J.sum(1) - 1.0).max() < 1e-12  # partition of unity
    for k in range(Nc + 1):  # exact on degree <= Nc
        assert np.abs(J @ zc**k - zf**k).max() < 1e-11


# ------------------------------------------------- kernels vs dense numpy
def dense_stiffness(u, G, D, n):
    """Au = D^T G D u per element with einsum, independent of the oracle loops."""
    E = len(u) // n**3
    U = u.reshape(E, n, n, n)  # [e, k, j, i]
    g = [a.reshape(E, n, n, n) for a in G]
    ur = np.einsum("ip,ekjp->ekji", D, U)
    us = np.einsum("jp,ekpi->ekji", D, U)
    ut = np.einsum("kp,epji->ekji", D, U)
    wr = g[0] * ur + g[3] * us + g[4] * ut
    ws = g[3] * ur + g[1] * us + g[5] * ut
    wt = g[4] * ur + g[5] * us + g[2] * ut
    out = np.einsum("pi,ekjp->ekji", D, wr) + np.einsum("pj,ekpi->ekji", D, ws) + np.einsum("pk,epji->ekji", D, wt)
    return out.reshape(-1)


@pytest.mark.parametrize("N", [1, 2, 3, 7, 11])
def test_oracle_stiffness_against_dense(N):
    L = S.oracle()
    n = N + 1
    E = 5
    rng = np.random.default_rng(N)
    npts = E * n**3
    u = rng.uniform(-1, 1, npts)
    G = [rng.uniform(0.1, 1, npts) if k < 3 else rng.uniform(-0.3, 0.3, npts) for k in range(6)]
    D = np.ascontiguousarray(S.gll(N)[2])
    GDu = [np.zeros(npts) for _ in range(3)]
    Au = np.zeros(npts)
    gd = (vp * 3)(*[a.ctypes.data for a in GDu])
    gg = (vp * 6)(*[a.ctypes.data for a in G])
    L.orc_dom_stiffness_matrix_1(gd, P(u), P(D), gg, npts, N, 3)
    L.orc_dom_stiffness_matrix_2(P(Au), gd, P(D), npts, N, 3)
    ref = dense_stiffness(u, G, D.reshape(n, n), n)
    assert np.abs(Au - ref).max() <= 1e-12 * np.abs(ref).max()


def test_oracle_restriction_against_dense():
    L = S.oracle()
    Nf, Nc, E = 7, 3, 6
    n_f, n_c = Nf + 1, Nc + 1
    J = np.ascontiguousarray(S.J_cf(Nc, Nf))
    u = np.random.default_rng(0).uniform(-1, 1, E * n_f**3)
    w1 = np.zeros(E * n_f * n_f * n_c)
    w2 = np.zeros(E * n_f * n_c * n_c)
    uc = np.zeros(E * n_c**3)
    L.orc_sub_restriction_1(P(w1), P(J), P(u), len(w1), n_f, n_c, 3)
    L.orc_sub_restriction_2(P(w2), P(J), P(w1), len(w2), n_f, n_c, 3)
    L.orc_sub_restriction_3(P(uc), P(J), P(w2), len(uc), n_f, n_c)
    Jm = J.reshape(n_f, n_c)
    ref = np.einsum("li,mj,nk,enml->ekji", Jm, Jm, Jm, u.reshape(E, n_f, n_f, n_f)).reshape(-1)
    assert np.abs(uc - ref).max() <= 1e-13 * np.abs(ref).max()


def test_oracle_csr_and_assembly():
    import scipy.sparse as sp

    L = S.oracle()
    rng = np.random.default_rng(3)
    rows, cols, n = 40, 30, 500
    r = rng.integers(0, rows, n).astype(np.int32)
    c = rng.integers(0, cols, n).astype(np.int32)
    v = rng.uniform(-1, 1, n)
    v[::17] = 1e-13  # dropped by the 1e-12 tolerance (csr_matrix.tpp:61-64,79)
    A = S.OrcCsr()
    assert L.orc_csr_assemble(ctypes.byref(A), rows, cols, P(r), P(c), P(v), n) == 0
    ptr, col, val = A.to_numpy()
    keep = np.abs(v) > 1e-12
    ref = sp.coo_matrix((v[keep], (r[keep], c[keep])), shape=(rows, cols)).tocsr()
    ref.sum_duplicates()
    ref.sort_indices()
    assert np.array_equal(ptr, ref.indptr) and np.array_equal(col, ref.indices)
    assert np.abs(val - ref.data).max() < 1e-15
    x = rng.uniform(-1, 1, cols)
    y = np.zeros(rows)
    L.orc_csr_multiply(P(y), P(ptr), P(col), P(val), P(x), rows)
    assert np.abs(y - ref @ x).max() < 1e-14
    # out-of-range entry is an error (csr_matrix.tpp:72-76)
    bad = np.array([rows], np.int32)
    B = S.OrcCsr()
    assert L.orc_csr_assemble(ctypes.byref(B), rows, cols, P(bad), P(np.zeros(1, np.int32)), P(np.ones(1)), 1) == -1
    # transpose
    At = S.OrcCsr()
    L.orc_csr_transpose(ctypes.byref(A), ctypes.byref(At))
    tp, tc, tv = At.to_numpy()
    rt = ref.T.tocsr()
    rt.sort_indices()
    assert np.array_equal(tp, rt.indptr) and np.array_equal(tc, rt.indices)
    L.orc_csr_free(ctypes.byref(A))
    L.orc_csr_free(ctypes.byref(At))


@pytest.mark.parametrize("n", [1, 127, 128, 129, 1000, 33333])
def test_oracle_reductions_follow_the_reference_tree(n):
    """128-wide pairwise tree then in-order block sum (domain.okl:125-131,
    domain.tpp:926): within rounding of an exactly rounded sum, and equal to an
    independent numpy statement of the same tree."""
    L = S.oracle()
    rng = np.random.default_rng(n)
    a, b, w = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(0, 1, n)
    nb = (n + 127) // 128
    block = np.zeros(nb)
    L.orc_dom_inner_product(P(block), P(a), P(b), P(w), n, nb)
    got = L.orc_block_sum(P(block), nb)
    t = np.zeros(nb * 128)
    t[:n] = a * b * w
    t = t.reshape(nb, 128).copy()
    alive = 64
    while alive > 0:
        t[:, :alive] += t[:, alive:2 * alive]
        alive //= 2
    ref = 0.0
    for x in t[:, 0]:
        ref += x
    assert got == ref
    assert abs(got - math.fsum(a * b * w)) <= 1e-13 * np.abs(a * b * w).sum()


# ------------------------------------------------------------ solver level
def test_domain_numbering_boundary_first_two_ranks():
    """domain.tpp:236-281: nodes whose local multiplicity differs from the
    global one are numbered first."""
    ms = [S.BoxMesh((4, 2, 2), 3, (2, 1, 1), r) for r in range(2)]
    W = S.OracleWorld(ms, 3)
    try:
        for r in range(2):
            assert W.num_bdary(r) == 7 * 7
            ptr, col, val = W.Q(r)
            assert np.all(np.diff(ptr) == 1) and np.all(val == 1.0)
            # the points on the shared plane x = 0.5 map to the prefix
            on_plane = np.abs(ms[r].x - 0.5) < 1e-12
            assert col[on_plane].max() < 49 and col[~on_plane].min() >= 49
        assert W.L.orc_world_num_interface_slots(W.w) == 49
        # assembled weight = 1 / global multiplicity
        for r in range(2):
            ptr, col, _ = W.Q(r)
            assert np.array_equal(W.assembled_weight(r)[col], 1.0 / ms[r].node_degree)
    finally:
        W.close()


def test_operator_is_symmetric_positive_on_assembled_space():
    m = S.BoxMesh((2, 2, 2), 3)
    W = S.OracleWorld([m], 3)
    try:
        rng = np.random.default_rng(1)
        a = W.dssum([rng.uniform(-1, 1, m.num_local_points)], True, True)
        b = W.dssum([rng.uniform(-1, 1, m.num_local_points)], True, True)
        Aa, Ab = W.stiffness(a), W.stiffness(b)
        assert abs(a[0] @ Ab[0] - b[0] @ Aa[0]) <= 1e-12 * abs(a[0] @ Ab[0])
        assert a[0] @ Aa[0] > 0
    finally:
        W.close()


@pytest.mark.parametrize("ranks", [2, 8])
def test_multi_rank_oracle_equals_single_rank(ranks):
    E, N = (4, 4, 4), 3
    Pg = S.rank_grid(ranks)
    m1 = S.BoxMesh(E, N)
    W1 = S.OracleWorld([m1], N)
    ms = [S.BoxMesh(E, N, Pg, r) for r in range(ranks)]
    WR = S.OracleWorld(ms, N)
    try:
        def rhs(W, meshes):
            us = [np.sin(np.pi * mm.x) * np.sin(2 * np.pi * mm.y) * np.sin(np.pi * mm.z) + mm.x * mm.y for mm in meshes]
            us = W.dssum(us, True, True)
            return us, W.stiffness(us)

        _, f1 = rhs(W1, [m1])
        _, fR = rhs(WR, ms)
        for method in ("fcg", "gmres"):
            u1, i1, h1 = W1.solve(f1, method)
            uR, iR, hR = WR.solve(fR, method)
            assert i1 == iR
            assert np.abs(h1 - hR).max() <= 1e-10 * h1[0]
    finally:
        W1.close()
        WR.close()


def test_golden_fixture_is_reproduced():
    """The committed oracle outputs on config C1 (tests/golden/oracle_c1.json)."""
    with open(os.path.join(S.GOLDEN_DIR, "oracle_c1.json")) as fh:
        gold = json.load(fh)
    E, N, red = tuple(gold["config"]["E"]), gold["config"]["N"], gold["config"]["reduction"]
    m = S.BoxMesh(E, N)
    W = S.OracleWorld([m], N)
    sd = S.OracleSubdomain(E, N, red)
    try:
        u = S.seeded_uniform(m.num_local_points, gold["config"]["seed"])

        def dig(a):
            return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

        assert dig(W.dssum([u], True, True)[0]) == gold["operators"]["dssum_mask_weight"]["sha256"]
        assert dig(W.stiffness([u])[0]) == gold["operators"]["stiffness"]["sha256"]
        assert dig(sd.stiffness(u)) == gold["operators"]["sub_stiffness"]["sha256"]
        assert dig(sd.tree(u)) == gold["operators"]["sub_tree"]["sha256"]
        assert W.residual_norm([u]) == gold["operators"]["residual_norm"]

        us = W.dssum([u], True, True)
        f = W.stiffness(us)
        assert dig(f[0]) == gold["rhs"]["f_sha256"]

        def pre(z, r):
            o, _, _ = sd.solve(r[0], "gmres")
            z[0][:] = o

        for key, kwargs in (("fcg+none", dict(method="fcg")), ("gmres+none", dict(method="gmres")), ("fcg+gmres", dict(method="fcg", precond=pre))):
            uu, its, hist = W.solve(f, **kwargs)
            g = gold["solves"][key]
            assert its == g["iterations"]
            assert np.array_equal(hist, np.array(g["history"]))
            assert np.abs(uu[0] - us[0]).max() <= 5e-6  # manufactured solution recovered
    finally:
        sd.close()
        W.close()


def test_fdd_preconditioner_cuts_iterations():
    with open(os.path.join(S.GOLDEN_DIR, "oracle_c1.json")) as fh:
        gold = json.load(fh)["solves"]
    assert gold["fcg+gmres"]["iterations"] < gold["fcg+none"]["iterations"] / 2
    assert gold["gmres+gmres"]["iterations"] < gold["gmres+none"]["iterations"] / 2
    h = gold["precond_gmres"]["history"]
    assert all(h[i + 1] <= h[i] for i in range(len(h) - 1))  # GMRES residual estimate is monotone


# --------------------------------------------------------------------------
# The full-domain-decomposition composite (oracle/fdd_oracle_composite.c): properties the construction of
# subdomain.tpp:86-2747 must have whatever hierarchy grades the superdomain.  The reference holds no fixture for it
#  and its own grading needs HYPRE; these pin the restatement independently of the product.
# --------------------------------------------------------------------------
@pytest.mark.parametrize("Pg,E,N,red", [((2, 1, 1), (8, 4, 4), 3, 2), ((2, 2, 1), (8, 8, 4), 4, 2), ((2, 2, 2), (8, 8, 8), 3, 1)])
def test_composite_structure_properties(Pg, E, N, red):
    F = S.OracleFdd(E, N, red, Pg)
    try:
        for r in range(F.R):
            info = F.info[r]
            ids, lv = F.region(r)
            # region = own elements at level 0, then rings at non-decreasing level; every global element at most once
            assert len(set(ids.tolist())) == len(ids) and np.all(np.diff(lv[: info["sub_elems"]]) >= 0)
            assert len(set(lv.tolist())) == len(F.deg)
            # the non-conforming Q is conforming in the sense that matters: dof values sampled from a trilinear function
            # (a member of every degree's space on these affine elements) are reproduced at EVERY region point, also at
            # the hanging ones that the J_cf rows interpolate (subdomain.tpp:1179-1585); Dirichlet points get 0
            x, y, z = (F.region_points(r, a) for a in "xyz")
            mask = F.region_points(r, "p_mask")
            fun = (1 + 2 * x - y + 0.5 * z + x * y - 0.3 * y * z + 0.7 * x * z + 0.2 * x * y * z) * mask
            Q = F.matrix(r, "Q")
            pd = F.point_dofs(r)
            assert Q.shape == (info["points"], info["sub_ext_dofs"])
            dof_val = np.full(info["sub_ext_dofs"], np.nan)
            dof_val[pd[pd >= 0]] = fun[pd >= 0]
            assert not np.isnan(dof_val).any()  # every dof sits on some point
            # rows are partitions of unity, except that Dirichlet dofs are not stored: a hanging point next to the
            # boundary keeps only the weights of its interior parents
            rowsum = Q.sum(axis=1).A1
            full = np.abs(rowsum - 1.0) <= 1e-13
            assert np.all(full[pd >= 0]) and np.all(rowsum[mask == 0] == 0)
            hanging = full & (pd < 0)
            assert hanging.sum() > 0 and (Q[hanging] != 0).sum(axis=1).max() >= 2
            assert np.abs(Q @ dof_val - fun)[full].max() <= 1e-13
            assert (F.matrix(r, "Qt") != Q.T).nnz == 0
            # copy-from-owner over interface and extended dofs (subdomain.tpp:2653-2729) is a projection
            M = F.matrix(r, "QQt_int")
            assert abs(M @ M - M).max() <= 1e-15 and abs(F.matrix(r, "Q_int") @ F.matrix(r, "Qt_int") - M).max() <= 1e-15
            # norm_weight marks each unique dof exactly once (subdomain.tpp:2731-2747)
            w = F.norm_weight(r)
            assert set(np.unique(w).tolist()) <= {0.0, 1.0} and int(w.sum()) == info["unique_dofs"]
            # superdomain operator: symmetric, rows of the interpolator's transpose non-negative
            A = F.matrix(r, "A_sup")
            assert abs(A - A.T).max() <= 1e-12 * abs(A).max()
            assert F.composite_levels(r)[0] >= info["sup_dofs"] > 0
    finally:
        F.close()


def test_uncoarsened_degree_one_composite_is_the_global_operator():
    """N = 1 and a superdomain overlap wider than the mesh: nothing is interpolated and nothing aggregated, so every
    rank's composite is the assembled global operator and a converged inner solve returns the global solution on the
    rank's own points -- rings, ring pull, coarse all-gather, Qt_coarse, Pt, A_sup and the interface maps all have to
    be right for that."""
    E, Pg = (4, 4, 4), (2, 2, 1)
    F = S.OracleFdd(E, 1, 1, Pg, 1, 100)
    meshes = [S.BoxMesh(E, 1, Pg, r) for r in range(F.R)]
    W = S.OracleWorld(meshes, 1)
    try:
        us = [(np.sin(3 * m.x + 1) * np.cos(2 * m.y) + m.z * m.x) * m.p_mask for m in meshes]
        r = W.stiffness(us)  # element-local A u*: its assembly Qt r is the global right-hand side of u*
        z, hist = F.precondition(r, "gmres", num_vectors=60, max_iterations=60, tolerance=1e-14)
        for k in range(F.R):
            assert hist[k][-1] <= 1e-12 * hist[k][0]
            assert np.abs(z[k] - us[k]).max() <= 1e-10 * np.abs(us[k]).max(), k
    finally:
        W.close()
        F.close()


# ------------------------------------------------------------------ the float twins (oracle/fdd_oracle_f32.c)
@pytest.mark.parametrize("N", [1, 3, 7, 15])
def test_float_oracle_twins_against_the_double_oracle(N):
    """The IEEE-single restatement of the inner solve's kernels (PTYPE = Float = float) is the double restatement
    rounded: element stiffness with the scatter and the scale fused into the load within single precision of the double
    kernels on the same values, and bit-identical to numpy's float32 arithmetic where that is one rounded operation."""
    L = S.oracle()
    vp = ctypes.c_void_p
    f32 = np.float32
    n3 = (N + 1) ** 3
    E = 5
    rng = np.random.default_rng(40 + N)
    G = [rng.uniform(0.5, 1.5, E * n3) if g < 3 else rng.uniform(-0.2, 0.2, E * n3) for g in range(6)]
    D = np.ascontiguousarray(S.gll(N)[2])
    ndof = E * n3 // 2
    pd = rng.integers(-1, ndof, E * n3).astype(np.int32)
    v = rng.uniform(-1, 1, ndof)
    scale = 0.61
    u = np.where(pd >= 0, scale * v[np.maximum(pd, 0)], 0.0)
    GDu = [np.zeros(E * n3) for _ in range(3)]
    Au = np.zeros(E * n3)
    gd = (vp * 3)(*[a.ctypes.data for a in GDu])
    gg = (vp * 6)(*[a.ctypes.data for a in G])
    L.orc_dom_stiffness_matrix_1(gd, S._p(u), S._p(D), gg, E * n3, N, 3)
    L.orc_dom_stiffness_matrix_2(S._p(Au), gd, S._p(D), E * n3, N, 3)
    G32 = [g.astype(f32) for g in G]
    D32, v32, sc32 = D.astype(f32), v.astype(f32), np.array([scale]).astype(f32)
    Au32 = np.zeros(E * n3, f32)
    gg32 = (vp * 6)(*[a.ctypes.data for a in G32])
    L.orc_f32_sub_stiffness(vp(Au32.ctypes.data), vp(v32.ctypes.data), vp(pd.ctypes.data), vp(sc32.ctypes.data), vp(D32.ctypes.data), gg32, E, N)
    assert np.abs(Au32 - Au).max() <= 2e-5 * np.abs(Au).max()
    assert not np.array_equal(Au32.astype(np.float64), Au)

    n = 10007
    a, b = rng.uniform(-1, 1, n).astype(f32), rng.uniform(-1, 1, n).astype(f32)
    out = np.zeros(n, f32)
    L.orc_f32_vector_vector_addition(vp(out.ctypes.data), ctypes.c_float(1.25), vp(a.ctypes.data), ctypes.c_float(-0.7), vp(b.ctypes.data), n)
    assert np.array_equal(out, (f32(1.25) * a).astype(f32) + (f32(-0.7) * b).astype(f32))
    L.orc_f32_vector_scaling(vp(out.ctypes.data), ctypes.c_double(1 / 3), vp(a.ctypes.data), n)
    assert np.array_equal(out, f32(1 / 3) * a)
    bs = [b, a]
    pp = (vp * 2)(*[x.ctypes.data for x in bs])
    sc = np.array([0.5, 2.0])
    dots = np.zeros(2)
    L.orc_f32_multi_inner_product_scaled(S._p(dots), vp(a.ctypes.data), pp, S._p(sc), 2, n)
    for k in range(2):
        ref = math.fsum(a.astype(np.float64) * (sc[k] * bs[k].astype(np.float64)))
        assert abs(dots[k] - ref) <= 1e-13 * np.sum(np.abs(a.astype(np.float64) * bs[k].astype(np.float64))) * sc[k]
    dst = np.zeros(n, f32)
    c = np.array([0.3, -0.8])
    nrm = L.orc_f32_multi_axpy_norm2_scaled(vp(dst.ctypes.data), vp(a.ctypes.data), S._p(c), ctypes.c_double(-1.0), pp, S._p(sc), 2, n)
    full = a.astype(np.float64) - (c[0] * sc[0]) * b.astype(np.float64) - (c[1] * sc[1]) * a.astype(np.float64)
    assert np.abs(dst - full).max() <= 6e-8 * np.abs(full).max()  # one rounding to float
    assert abs(nrm - np.sum(dst.astype(np.float64) ** 2)) <= 1e-12 * nrm


def test_element_diagonal_closed_form_against_the_kernels():
    """The point-Jacobi option's element diagonal (closed form read off subdomain.okl:4-101) equals the restated
    kernels applied to unit vectors: deformed 3-D elements (all six factors non-zero) and a deformed 2-D mesh."""
    cases = (
        (4, lambda d: S.DeformedMesh((2, 2, 2), d, 0.05)),
        (3, lambda d: S.QuadMeshRanks((3, 2), d, (1, 1), 0, amplitude=0.04)),
    )
    for deg, make in cases:
        sd = S.OracleSubdomain(None, deg, 2, meshes=[make(d) for d in S.level_degrees(deg, 2)])
        assert sd.element_diagonal_check() <= 1e-12
        assert np.all(sd.jacobi_diagonal() > 0)
        sd.close()


def test_kershaw_map_properties():
    """The generalized Kershaw map the meshes of the reference's experiments come from (run.py:25-47: eps = 0.3; here
    tests/support.py: kershaw_map, the numpy statement host/box_mesh.hpp is held to): the identity at eps = 1, x kept, the
    boundary of the cube mapped onto itself, continuous across the six x-layers, monotone in y and z (positive Jacobian),
    and the isoparametric factors of a mesh on it symmetric positive definite point by point with all six live."""
    rng = np.random.default_rng(3)
    x, y, z = rng.uniform(0, 1, (3, 20000))
    X, Y, Z = S.kershaw_map(1.0, 1.0, x, y, z)
    assert np.array_equal(X, x) and np.abs(Y - y).max() <= 1e-15 and np.abs(Z - z).max() <= 1e-15
    for eps in (0.3, 0.6):
        X, Y, Z = S.kershaw_map(eps, eps, x, y, z)
        assert np.array_equal(X, x) and Y.min() >= 0 and Y.max() <= 1 and Z.min() >= 0 and Z.max() <= 1
        for face in (0.0, 1.0):  # y = 0, 1 and z = 0, 1 stay put
            assert np.abs(S.kershaw_map(eps, eps, x, np.full_like(y, face), z)[1] - face).max() <= 1e-15
            assert np.abs(S.kershaw_map(eps, eps, x, y, np.full_like(z, face))[2] - face).max() <= 1e-15
        for k in range(1, 6):  # continuity across the layer boundaries x = k / 6
            lo, hi = S.kershaw_map(eps, eps, np.full_like(y, k / 6.0 - 1e-13), y, z), S.kershaw_map(eps, eps, np.full_like(y, k / 6.0 + 1e-13), y, z)
            assert np.abs(lo[1] - hi[1]).max() <= 1e-11 and np.abs(lo[2] - hi[2]).max() <= 1e-11
        ys = np.sort(y)
        for xv in (0.05, 0.21, 0.4, 0.62, 0.77, 0.95):  # monotone sections
            assert np.all(np.diff(S.kershaw_map(eps, eps, np.full_like(ys, xv), ys, ys)[1]) > 0)
    m = S.KershawMesh((6, 4, 4), 3, 0.3)
    G = np.stack(m.g)  # rr, ss, tt, rs, rt, st
    assert all(np.abs(G[k]).max() > 1e-3 * np.abs(G[0]).max() for k in (3, 4, 5))
    M = np.empty((m.num_local_points, 3, 3))
    for (a, b), k in zip([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)], range(6)):
        M[:, a, b] = M[:, b, a] = G[k]
    assert np.linalg.eigvalsh(M).min() > 0
