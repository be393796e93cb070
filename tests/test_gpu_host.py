"""GPU parity of the C++ host layer (Domain / Subdomain / CSR_Matrix mirrors of
the reference, libfdd_host.so) against the CPU oracle, through the C-ABI of
include/fdd_host.h, on the same seeded inputs.

Tolerances: operators built from bit-exact kernels must be
bit-identical; reductions and therefore Krylov histories are tolerance-based:
identical iteration counts, residual history within 1e-8 relative, solution
within 1e-9 of the oracle's in the max norm.
"""
import numpy as np
import pytest

import support as S
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H

pytestmark = pytest.mark.gpu

E1, N1, RED1 = (4, 4, 4), 3, 2  # BASELINE config C1


@pytest.fixture(scope="module")
def setup(gpu):
    H.init(0)
    H.comm_single()
    H.set_print(False)
    return True


def make_problem(E, N, red, with_sub=True):
    p = H.Problem.box(E, (1, 1, 1), N, red, with_sub)
    if with_sub:
        p.set_flag("sub_use_preconditioner", 0)  # these tests run the inner solves without the V-cycle (test_gpu_amg.py has it)
    # feed the reference's own GLL tables so operator parity is bit-exact
    for lvl in range(p.info["num_levels"]):
        p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
    return p


def oracle_subdomain(p, N, red):
    meshes = [S.ArrayMesh.from_problem(p, lvl) for lvl in range(p.info["num_levels"])]
    return S.OracleSubdomain(None, N, red, meshes=meshes)


def test_gll_tables_of_the_host_layer(setup):
    """host/gll.hpp against the reference's Fortran speclib tables."""
    p = H.Problem.box((2, 2, 2), (1, 1, 1), 15, 1, True)
    p.set_flag("sub_use_preconditioner", 0)
    try:
        assert p.info["num_levels"] == 15
        for lvl in range(15):
            N = p.level_degree(lvl)
            D = p.get_D_hat(lvl)
            ref = S.gll(N)[2]
            assert np.abs(D - ref).max() <= 2e-13 * np.abs(ref).max(), N
    finally:
        p.close()


def test_box_mesh_matches_numpy_statement(setup):
    p = make_problem((4, 6, 2), 3, 2, False)
    m = S.BoxMesh((4, 6, 2), 3)
    try:
        assert np.array_equal(p.mesh_array("glo_num"), m.glo_num)
        assert np.array_equal(p.mesh_array("node_degree"), m.node_degree)
        assert np.array_equal(p.mesh_array("p_mask"), m.p_mask)
        for name, ref in (("x", m.x), ("y", m.y), ("z", m.z)):
            assert np.abs(p.mesh_array(name) - ref).max() < 1e-15
        for g in range(6):
            got = p.mesh_array(f"g_{g + 1}")
            assert np.abs(got - m.g[g]).max() <= 1e-15 * max(np.abs(m.g[g]).max(), 1e-300) + 0.0
        assert p.info["num_total_nodes"] == m.global_nodes
    finally:
        p.close()


def test_domain_setup_and_operators(setup):
    p = make_problem(E1, N1, RED1, False)
    m = S.ArrayMesh.from_problem(p)  # oracle and product share the mesh arrays bit for bit
    W = S.OracleWorld([m], N1)
    try:
        assert p.info["num_local_points"] == 4096 and p.info["num_local_nodes"] == 2197
        assert p.info["num_bdary_nodes"] == 0
        # Q / Qt / weights (domain.tpp:233-302)
        # the product orders its nodes differently (Dirichlet nodes at the ends): same matrices up to that permutation
        _, qp, qc, qv = p.csr(0)
        oq = W.Q(0)
        assert np.array_equal(qp, oq[0]) and np.array_equal(qv, oq[2])
        perm = {}
        for a, b in zip(qc, oq[1]):
            assert perm.setdefault(int(a), int(b)) == int(b)
        assert len(perm) == 2197 and len(set(perm.values())) == 2197
        _, tp, tc, tv = p.csr(1)
        ot = W.Qt(0)
        for n in (0, 1, 57, 1000, 2196):  # a gather row lists the node's points in ascending order on both sides
            o = perm[n]
            assert np.array_equal(tc[tp[n] : tp[n + 1]], ot[1][ot[0][o] : ot[0][o + 1]])
        assert np.array_equal(p.assembled_weight()[qc], W.assembled_weight(0)[oq[1]])

        u = S.seeded_uniform(p.n, 1234)
        for mask, weight in ((True, False), (True, True), (False, False), (False, True)):
            assert np.array_equal(p.dssum(u, mask, weight), W.dssum([u], mask, weight)[0])
        assert np.array_equal(p.stiffness(u), W.stiffness([u])[0])
        assert np.array_equal(p.stiffness(u, dssum=True), W.stiffness([u], True)[0])
        ref = W.residual_norm([u])
        assert abs(p.residual_norm(u) - ref) <= 1e-13 * ref
    finally:
        W.close()
        p.close()


@pytest.mark.parametrize("method", ["fcg", "gmres"])
@pytest.mark.parametrize("rhs", ["sin", "seeded"])
def test_outer_solve_without_preconditioner(setup, method, rhs):
    """Config C1 plumbing: outer solver with use_preconditioner = false
    (domain.tpp:648-651)."""
    p = make_problem(E1, N1, RED1, False)
    m = S.ArrayMesh.from_problem(p)  # oracle and product share the mesh arrays bit for bit
    W = S.OracleWorld([m], N1)
    try:
        if rhs == "sin":
            us = np.sin(np.pi * m.x) * np.sin(np.pi * m.y) * np.sin(np.pi * m.z)
        else:
            us = S.seeded_uniform(p.n, 1234)
        u_star, f = p.make_rhs_from(us)
        o_star = W.dssum([us], True, True)
        o_f = W.stiffness(o_star)
        assert np.array_equal(u_star, o_star[0]) and np.array_equal(f, o_f[0])

        u, its, hist = p.solve(f, method)
        ou, oits, ohist = W.solve(o_f, method)
        assert its == oits and len(hist) == len(ohist)
        assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
        assert np.abs(u - ou[0]).max() <= 1e-9 * np.abs(ou[0]).max()
        # the check the reference omits: the manufactured solution is recovered
        assert np.abs(u - u_star).max() <= 1e-3 * np.abs(u_star).max()
    finally:
        W.close()
        p.close()


def test_subdomain_operators(setup):
    p = make_problem(E1, N1, RED1, True)
    sd = oracle_subdomain(p, N1, RED1)
    try:
        assert p.info["num_levels"] == len(sd.deg) == 2
        assert p.info["sub_num_values"] == sd.num_values
        assert p.info["sub_num_dofs"] == sd.L.orc_subdomain_num_dofs(sd.s) == 11 ** 3
        u = S.seeded_uniform(p.n, 77)
        assert np.array_equal(p.sub_op("tree", u), sd.tree(u))
        assert np.array_equal(p.sub_op("stiffness", u), sd.stiffness(u))
        assert np.array_equal(p.sub_op("dssum", u), sd.dssum(u))
        ref = sd.residual_norm(u)
        assert abs(p.sub_residual_norm(u) - ref) <= 1e-13 * ref
    finally:
        sd.close()
        p.close()


@pytest.mark.parametrize("method", ["gmres", "fcg"])
def test_preconditioner_application(setup, method):
    """Subdomain::generalized_minimum_residual / flexible_conjugate_gradient
    (subdomain.tpp:4309-4489 / 4161-4268), 4 inner iterations."""
    p = make_problem(E1, N1, RED1, True)
    sd = oracle_subdomain(p, N1, RED1)
    try:
        r = S.seeded_uniform(p.n, 5) - 0.5
        z, hist = p.precond_apply(r, method)
        oz, oits, ohist = sd.solve(r, method)
        assert len(hist) == len(ohist) == 5
        assert np.abs(hist - ohist).max() <= 1e-10 * ohist[0]
        assert np.abs(z - oz).max() <= 1e-10 * np.abs(oz).max()
    finally:
        sd.close()
        p.close()


@pytest.mark.parametrize("outer", ["fcg", "gmres"])
@pytest.mark.parametrize("inner", ["gmres", "fcg"])
def test_fdd_preconditioned_solve(setup, outer, inner):
    """Config C1 end to end: outer Krylov + FDD single-subdomain preconditioner
    + stitching dssum (domain.tpp:697-706)."""
    p = make_problem(E1, N1, RED1, True)
    p.set_options(preconditioner_type=0 if inner == "fcg" else 1)
    m = S.ArrayMesh.from_problem(p)  # oracle and product share the mesh arrays bit for bit
    W = S.OracleWorld([m], N1)
    sd = oracle_subdomain(p, N1, RED1)
    try:
        us = S.seeded_uniform(p.n, 1234)
        u_star, f = p.make_rhs_from(us)
        u, its, hist = p.solve(f, outer)

        def pre(z, r):
            out, _, _ = sd.solve(r[0], inner)
            z[0][:] = out

        ou, oits, ohist = W.solve([f], outer, precond=pre)
        assert its == oits and len(hist) == len(ohist)
        assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
        assert np.abs(u - ou[0]).max() <= 1e-9 * np.abs(ou[0]).max()
        assert np.abs(u - u_star).max() <= 1e-3 * np.abs(u_star).max()
    finally:
        sd.close()
        W.close()
        p.close()


def test_reference_shaped_and_restructured_paths_agree(setup):
    """The launch sequences of the reference (Qt/Q SpMV pair per dssum; one
    SpMV pair + dot per Gram-Schmidt coefficient) and the MI355X-first ones
    (gather-scatter dssum; cached assembled basis, multi-dot, multi-axpy;
    inner GMRES entirely on dof vectors) give the same solve: bit-identical
    where only bit-exact kernels changed, within reduction rounding otherwise."""
    out = {}
    for fused, restructured, assembled, outer, dev_book in ((0, 0, 0, 0, 0), (1, 0, 0, 0, 0), (1, 1, 0, 0, 0), (1, 1, 1, 0, 0), (1, 1, 1, 1, 0), (1, 1, 1, 1, 1)):
        p = make_problem(E1, N1, RED1, True)
        p.set_flag("fused_dssum", fused)
        p.set_flag("restructured_inner_solve", restructured)
        p.set_flag("assembled_inner_solve", assembled)
        p.set_flag("assembled_outer_solve", outer)
        p.set_flag("device_bookkeeping", dev_book)
        _, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234))
        r = S.seeded_uniform(p.n, 5) - 0.5
        out[(fused, restructured, assembled, outer, dev_book)] = (p.solve(f, "fcg"), p.precond_apply(r, "gmres"), p.dssum(r, True, True), p.sub_op("dssum", r))
        p.close()
    ref, fus, res, asm, nod, dbk = out[(0, 0, 0, 0, 0)], out[(1, 0, 0, 0, 0)], out[(1, 1, 0, 0, 0)], out[(1, 1, 1, 0, 0)], out[(1, 1, 1, 1, 0)], out[(1, 1, 1, 1, 1)]
    # fused dssum: same bits everywhere
    assert np.array_equal(ref[2], fus[2]) and np.array_equal(ref[3], fus[3])
    assert ref[0][1] == fus[0][1] and np.array_equal(ref[0][2], fus[0][2]) and np.array_equal(ref[0][0], fus[0][0])
    assert np.array_equal(ref[1][0], fus[1][0])
    # restructured / assembled inner solve: same iterates up to the rounding of the dots
    # device-side bookkeeping repeats the host statements: the same bits as the host-driven node-space solve
    assert nod[0][1] == dbk[0][1] and np.array_equal(nod[0][2], dbk[0][2]) and np.array_equal(nod[0][0], dbk[0][0])
    assert np.array_equal(nod[1][0], dbk[1][0]) and np.array_equal(nod[1][1], dbk[1][1])
    for alt in (res, asm, nod, dbk):
        assert ref[0][1] == alt[0][1]
        assert np.abs(ref[0][0] - alt[0][0]).max() <= 1e-9 * np.abs(ref[0][0]).max()
        assert np.abs(ref[0][2] - alt[0][2]).max() <= 1e-10 * ref[0][2][0]
        assert np.abs(ref[1][0] - alt[1][0]).max() <= 1e-12 * np.abs(ref[1][0]).max()
        assert np.abs(ref[1][1] - alt[1][1]).max() <= 1e-12 * ref[1][1][0]


def test_high_order_mfma_path(setup):
    """Config C3's degree (N = 15) at test size: Domain and Subdomain apply the
    stiffness on the fp64 matrix cores.  Operator within 1e-12 of the bit-exact
    scalar kernel and of the oracle; the preconditioned solve keeps the oracle's
    iteration count and residual history."""
    E, N, red = (2, 2, 2), 15, 6
    p = make_problem(E, N, red, True)
    m = S.ArrayMesh.from_problem(p)
    W = S.OracleWorld([m], N)
    sd = oracle_subdomain(p, N, red)
    try:
        assert [p.level_degree(lvl) for lvl in range(p.info["num_levels"])] == [15, 9, 3, 1]
        u = S.seeded_uniform(p.n, 15)
        ref = W.stiffness([u])[0]
        got = p.stiffness(u)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
        p.set_flag("mfma_stiffness", 0)
        assert np.array_equal(p.stiffness(u), ref)  # scalar fused kernel: same bits
        p.set_flag("mfma_stiffness", 1)
        got = p.sub_op("stiffness", u)
        assert np.abs(got - sd.stiffness(u)).max() <= 1e-12 * np.abs(ref).max()

        u_star, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234))
        p.set_options(max_iterations=12, tolerance=0.0)
        uu, its, hist = p.solve(f, "fcg")

        def pre(z, r):
            out, _, _ = sd.solve(r[0], "gmres")
            z[0][:] = out

        ou, oits, ohist = W.solve([f], "fcg", max_iterations=12, tolerance=0.0, precond=pre)
        assert its == oits == 12
        assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
    finally:
        sd.close()
        W.close()
        p.close()


def test_stepwise_pcg_equals_the_solver(setup):
    """fcg_begin + fcg_step (what bench.py times) walks the same iterates as
    flexible_conjugate_gradient."""
    p = make_problem(E1, N1, RED1, True)
    try:
        _, f = p.make_rhs_from(S.seeded_uniform(p.n, 9))
        p.set_options(max_iterations=6, tolerance=0.0)
        u, its, hist = p.solve(f, "fcg")
        p.pcg_begin(f)
        last = p.pcg_steps(6)
        assert its == 6 and last == hist[6]
        assert np.array_equal(p.pcg_solution(), u)
    finally:
        p.close()


def test_mesh_file_roundtrip(setup, tmp_path):
    """The reference's on-disk input (domain.tpp:45-224): write the box mesh as
    a Nek5000-style file set, read it back through Domain::initialize(dir, N)."""
    import ctypes

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    d = str(tmp_path / "mesh")
    E = (ctypes.c_int * 3)(3, 2, 2)
    P = (ctypes.c_int * 3)(1, 1, 1)
    for deg in (3, 1):
        lib.host().call("fddh_write_box_mesh_files", d.encode(), E, P, deg, 0)
    p = H.Problem.from_directory(d, 3, 2)
    q = H.Problem.box((3, 2, 2), (1, 1, 1), 3, 2)
    p.set_flag("sub_use_preconditioner", 0)
    q.set_flag("sub_use_preconditioner", 0)
    try:
        for name in ("x", "glo_num", "node_degree", "p_mask", "g_1", "g_3"):
            assert np.array_equal(p.mesh_array(name), q.mesh_array(name))
        u = S.seeded_uniform(p.n, 3)
        assert np.array_equal(p.stiffness(u, True), q.stiffness(u, True))
    finally:
        p.close()
        q.close()


@pytest.mark.parametrize("nv,mi", [(2, 5), (3, 7), (4, 9), (1, 1), (7, 7), (8, 8), (8, 11)])
@pytest.mark.parametrize("device_bookkeeping", [0, 1])
def test_inner_gmres_restarts_and_stops_inside_a_cycle(setup, nv, mi, device_bookkeeping):
    """Inner GMRES(nv) with max_iterations not a multiple of nv: restart cycles (the residual is
    rebuilt from the dof-space solution) and a stop in the middle of the last cycle -- recorded by the
    device-side bookkeeping while the remaining steps still run -- against the oracle's host loop.  nv = mi = 1, 2, 4, 8 are
    the settings the reference's harness sweeps (run.py:151-152); 8 is the most the device path holds (FDD_MULTI_MAX)."""
    p = make_problem(E1, N1, RED1, True)
    sd = oracle_subdomain(p, N1, RED1)
    try:
        p.set_options(sub_num_vectors=nv, sub_max_iterations=mi)
        p.set_flag("device_bookkeeping", device_bookkeeping)
        r = S.seeded_uniform(p.n, 15) - 0.5
        z, hist = p.precond_apply(r, "gmres")
        oz, oits, ohist = sd.solve(r, "gmres", num_vectors=nv, max_iterations=mi)
        assert len(hist) == len(ohist) == mi + 1
        assert np.abs(hist - ohist).max() <= 1e-9 * ohist[0]
        assert np.abs(z - oz).max() <= 1e-9 * np.abs(oz).max()
    finally:
        sd.close()
        p.close()


def test_curved_mesh_from_files(setup, tmp_path):
    """A deformed (non-affine) mesh read from the reference's file format: all six geometric factors
    are non-zero in the whole solve.  Product vs oracle on the same arrays, and against the
    manufactured solution."""
    E, N, red = (3, 3, 2), 5, 2
    d = str(tmp_path / "curved")
    for deg in S.level_degrees(N, red):
        S.write_mesh_files(d, S.DeformedMesh(E, deg))
    p = H.Problem.from_directory(d, N, red)
    p.set_flag("sub_use_preconditioner", 0)
    for lvl in range(p.info["num_levels"]):
        p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
    meshes = [S.ArrayMesh.from_problem(p, lvl) for lvl in range(p.info["num_levels"])]
    W = S.OracleWorld([meshes[0]], N)
    sd = S.OracleSubdomain(None, N, red, meshes=meshes)
    try:
        for g in ("g_4", "g_5", "g_6"):
            assert np.abs(p.mesh_array(g)).max() > 1e-4
        u = S.seeded_uniform(p.n, 8)
        assert np.array_equal(p.stiffness(u), W.stiffness([u])[0])
        # the operator is symmetric positive on assembled data: <v, A u> = <u, A v>
        us, vs = W.dssum([u], True, True)[0], W.dssum([S.seeded_uniform(p.n, 9)], True, True)[0]
        Au, Av = p.stiffness(us, dssum=True), p.stiffness(vs, dssum=True)
        wgt = 1.0 / meshes[0].node_degree
        assert abs(np.dot(vs * wgt, Au) - np.dot(us * wgt, Av)) <= 1e-12 * abs(np.dot(us * wgt, Au))
        assert np.dot(us * wgt, Au) > 0

        u_star, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234))
        x, its, hist = p.solve(f, "fcg")

        def pre(z, r):
            out, _, _ = sd.solve(r[0], "gmres")
            z[0][:] = out

        ox, oits, ohist = W.solve([f], "fcg", precond=pre)
        assert its == oits and np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
        assert np.abs(x - ox[0]).max() <= 1e-8 * np.abs(ox[0]).max()
        assert np.abs(x - u_star).max() <= 1e-4 * np.abs(u_star).max()
    finally:
        sd.close()
        W.close()
        p.close()


@pytest.mark.parametrize("E,N,red", [((4, 4, 4), 3, 2), ((3, 3, 3), 7, 6), ((6, 4, 4), 5, 2)])
def test_kershaw_mesh(setup, E, N, red):
    """The reference's experiment geometry (run.py:25-47, run.sh:30: Kershaw, eps = 0.3) generated by the host layer
    (host/box_mesh.hpp: all six geometric factors non-zero, level by level at each level's own degree), against the numpy
    statement of the same map and, on the problem's own arrays, against the oracle: the fused stiffness bit for bit, the
    outer PCG and GMRES with the full-domain-decomposition preconditioner iteration for iteration.  4^3 and 3^3 elements
    do not align with the map's six x-layers (the reference's 16^3 and 64^3 meshes do not either); 6 x 4 x 4 does."""
    p = H.Problem.kershaw(E, (1, 1, 1), N, red, 0.3)
    p.set_flag("sub_use_preconditioner", 0)
    for lvl in range(p.info["num_levels"]):
        p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
    meshes = [S.ArrayMesh.from_problem(p, lvl) for lvl in range(p.info["num_levels"])]
    W = S.OracleWorld([meshes[0]], N)
    sd = S.OracleSubdomain(None, N, red, meshes=meshes)
    try:
        for lvl, m in enumerate(meshes):
            twin = S.KershawMesh(E, p.level_degree(lvl), 0.3)
            gmax = max(np.abs(g).max() for g in twin.g)
            for a, b in [(m.x, twin.x), (m.y, twin.y), (m.z, twin.z)]:
                assert np.abs(a - b).max() <= 1e-13
            for k in range(6):
                assert np.abs(m.g[k] - twin.g[k]).max() <= 1e-13 * gmax, (lvl, k)
        assert all(np.abs(meshes[0].g[k]).max() > 1e-3 * np.abs(meshes[0].g[0]).max() for k in (3, 4, 5))  # the off-diagonal factors are live
        p.set_flag("affine_geometry", 1)
        assert not p.affine_info()["fine_domain"] and p.affine_info()["max_deviation"] > 1e-3  # nothing switches to the affine-elements option
        p.set_flag("affine_geometry", 0)
        u = S.seeded_uniform(p.n, 8)
        assert np.array_equal(p.stiffness(u), W.stiffness([u])[0])
        u_star, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234))

        def pre(z, r):
            out, _, _ = sd.solve(r[0], "gmres")
            z[0][:] = out

        for method in ("fcg", "gmres"):
            x, its, hist = p.solve(f, method)
            ox, oits, ohist = W.solve([f], method, precond=pre)
            assert its == oits, (method, its, oits)
            assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
            assert np.abs(x - ox[0]).max() <= 1e-6 * np.abs(ox[0]).max()  # long iterations on a badly conditioned operator (test_cpu_multirank's note)
    finally:
        sd.close()
        W.close()
        p.close()


def test_two_dimensional_mesh_from_files(setup, tmp_path):
    """The `dim == 2` branches of Domain / Subdomain and their kernels, end to end (dim2_checks.py)."""
    import dim2_checks

    its = dim2_checks.check_two_dimensional_solve(H, str(tmp_path / "quad"))
    assert 0 < its < 40


def test_poisson_driver_binary(setup, tmp_path):
    """The compiled driver (host/poisson.cpp, the reference's poisson.cpp:main): a generated box problem, its mesh
    written in the reference's file format, then the same solve from those files -- same iteration count,
    manufactured solution recovered, in a process of its own (no Python, no torch)."""
    import os
    import re
    import subprocess

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    exe = os.path.join(lib.PKG_DIR, "poisson")
    if not os.path.exists(exe):  # build() links the driver next to the libraries
        subprocess.run(["make", "-C", os.path.join(lib.PKG_DIR, "host"), "-s"], check=False)
    if not os.path.exists(exe):
        pytest.skip("driver binary not built")
    d = str(tmp_path / "mesh")
    runs = []
    for args in (["-", "3", "2", "0", "0", "--box", "4", "4", "4", "--write-mesh", d], [d, "3", "2", "0", "0"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        its = int(re.search(r"Iterations: (\d+)", out.stdout).group(1))
        err = float(re.search(r"max \|u - u\*\| on rank 0: (\S+)", out.stdout).group(1))
        assert "Number of dimensions: 3" in out.stdout
        assert 0 < its < 30 and err < 1e-4, out.stdout  # outer tolerance 1e-7 on the residual (domain.hpp:113)
        runs.append((its, err))
    assert runs[0][0] == runs[1][0]


def test_single_precision_preconditioner(gpu):
    """The reference's PTYPE = Float = float (config.hpp:19-20, poisson.cpp:206; run.py:157): the whole inner solve --
    element stiffness, gather, Krylov vectors, V-cycle -- on float data.  Checked against the double path (itself
    held to the oracle): the preconditioner's output to single-precision rounding (1e-5 of its max norm, written
    here), the same outer iteration counts, the same solution to the outer tolerance; switching back is bit-exact."""
    H.comm_single()
    for amg, (E, N, red) in ((0, ((4, 4, 4), 3, 2)), (1, ((4, 4, 4), 3, 2)), (1, ((6, 6, 6), 7, 6))):
        p = H.Problem.box(E, (1, 1, 1), N, red, True)
        for lvl in range(p.info["num_levels"]):
            p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])  # the oracle's tables, before anything is derived from them
        p.set_flag("sub_use_preconditioner", amg)
        if amg:
            p.amg_build(coarsest_size=40)
        _, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234))
        r = S.seeded_uniform(p.n, 5) - 0.5
        z64, h64 = p.precond_apply(r, "gmres")
        u64, its64, hist64 = p.solve(f, "fcg")
        p.set_flag("preconditioner_precision", 32)
        z32, h32 = p.precond_apply(r, "gmres")
        u32, its32, hist32 = p.solve(f, "fcg")
        assert np.abs(z32 - z64).max() <= 1e-5 * np.abs(z64).max()
        assert not np.array_equal(z32, z64)  # it really ran in float
        if not amg:
            # and directly against the CPU oracle's (double) inner solve: the float path is single-precision close to it
            sd = S.OracleSubdomain(E, N, red)
            z_or, _, _ = sd.solve(r, "gmres")
            z32b, _ = p.precond_apply(r, "gmres")
            assert np.abs(z32b - z_or).max() <= 1e-5 * np.abs(z_or).max()
            sd.close()
        assert np.abs(h32 - h64).max() <= 1e-5 * h64[0]
        assert abs(its32 - its64) <= 1, (its32, its64)
        assert hist32[-1] <= 1e-7 * hist32[0] * 1.0001
        assert np.abs(u32 - u64).max() <= 1e-5 * np.abs(u64).max()
        # the stepwise interface bench.py drives (lazy: no host synchronisation inside the inner solves)
        p.pcg_begin(f)
        last32 = p.pcg_steps(2)
        assert abs(last32 - hist32[2]) <= 1e-5 * hist32[0]
        p.set_flag("preconditioner_precision", 64)
        z64b, _ = p.precond_apply(r, "gmres")
        assert np.array_equal(z64b, z64)
        p.close()


@pytest.mark.parametrize("E,N,red", [((4, 4, 4), 3, 2), ((3, 3, 3), 7, 6)])
def test_point_jacobi_option(setup, E, N, red):
    """Point-Jacobi in the inner solver's preconditioner slot ("sub_use_preconditioner" = 2; a labelled option of
    this build, not in the reference): the exact diagonal of the inner iteration's operator against the oracle's and
    against the operator applied to unit vectors, the preconditioner application (GMRES and flexible CG, every launch
    sequence) and the outer solve against the oracle with the same option, the float path against the double one."""
    p = make_problem(E, N, red, True)
    m = S.ArrayMesh.from_problem(p)
    W = S.OracleWorld([m], N)
    sd = oracle_subdomain(p, N, red)
    try:
        assert sd.element_diagonal_check() <= 1e-12
        n_u = p.sub_info()["unique_dofs"]
        dj, od = p.sub_jacobi_diagonal(), sd.jacobi_diagonal()
        pd, opd = p.sub_point_dofs(), sd.point_dofs()
        has = pd >= 0
        assert np.array_equal(has, opd >= 0)
        assert np.abs(dj[pd[has]] - od[opd[has]]).max() <= 1e-12 * od.max()  # numberings differ: compare through the points
        x = np.zeros(n_u)
        for d in np.random.default_rng(1).integers(0, n_u, 40):
            x[d] = 1.0
            assert abs(p.sub_dof_operator(x)[d] - dj[d]) <= 1e-12 * dj.max()
            x[d] = 0.0

        p.set_flag("sub_use_preconditioner", 2)
        r = S.seeded_uniform(p.n, 5) - 0.5
        for method in ("gmres", "fcg"):
            oz, _, ohist = sd.solve(r, method, use_preconditioner=2)
            z, hist = p.precond_apply(r, method)
            assert np.abs(hist - ohist).max() <= 1e-10 * ohist[0], method
            assert np.abs(z - oz).max() <= 1e-10 * np.abs(oz).max(), method
        # the reference-shaped launch sequence (point-space vectors, SpMV chain, host scalars) runs the same option
        for flag in ("assembled_inner_solve", "restructured_inner_solve", "fused_dssum", "device_bookkeeping"):
            p.set_flag(flag, 0)
        z2, hist2 = p.precond_apply(r, "gmres")
        oz, _, ohist = sd.solve(r, "gmres", use_preconditioner=2)
        assert np.abs(hist2 - ohist).max() <= 1e-10 * ohist[0]
        assert np.abs(z2 - oz).max() <= 1e-10 * np.abs(oz).max()
        for flag in ("assembled_inner_solve", "restructured_inner_solve", "fused_dssum", "device_bookkeeping"):
            p.set_flag(flag, 1)

        _, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234))
        u, its, hist = p.solve(f, "fcg")

        def pre(zz, rr):
            out, _, _ = sd.solve(rr[0], "gmres", use_preconditioner=2)
            zz[0][:] = out

        ou, oits, ohist = W.solve([f], "fcg", precond=pre)
        assert its == oits
        assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
        assert np.abs(u - ou[0]).max() <= 1e-9 * np.abs(ou[0]).max()
        p.set_flag("sub_use_preconditioner", 0)
        _, its_plain, _ = p.solve(f, "fcg")
        assert its <= its_plain  # what the option is for
        p.set_flag("sub_use_preconditioner", 2)

        z64, h64 = p.precond_apply(r, "gmres")
        p.set_flag("preconditioner_precision", 32)
        z32, h32 = p.precond_apply(r, "gmres")
        u32, its32, hist32 = p.solve(f, "fcg")
        assert np.abs(z32 - z64).max() <= 1e-5 * np.abs(z64).max() and not np.array_equal(z32, z64)
        assert abs(its32 - its) <= 1 and hist32[-1] <= 1e-7 * hist32[0] * 1.0001
        p.set_flag("preconditioner_precision", 64)
    finally:
        sd.close()
        W.close()
        p.close()


def test_affine_geometry_option(setup, tmp_path):
    """ "affine_geometry" (a labelled option of this build, off by default; the reference always streams the six factor
    arrays): on a box mesh every element passes the check of the mesh's OWN factors against c_f(e) (w_i w_j) w_k, the
    outer operator and the inner solves (double and float) run on the kernel that does not read them, and the solve is
    the streamed one to rounding -- same iteration count against the oracle, solution to 1e-12.  On a deformed mesh no
    element passes, nothing is switched and the results are the streamed ones bit for bit."""
    E, N, red = (4, 3, 3), 7, 6
    p = make_problem(E, N, red, True)
    W = S.OracleWorld([S.ArrayMesh.from_problem(p)], N)
    sd = oracle_subdomain(p, N, red)
    try:
        assert p.affine_info() == {"fine_domain": False, "sub_lists_affine": 0, "sub_lists": 1, "max_deviation": -1.0}
        _, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234))
        u0, its0, h0 = p.solve(f, "fcg")
        r = S.seeded_uniform(p.n, 5) - 0.5
        z0, zh0 = p.precond_apply(r, "gmres")
        q0 = p.stiffness(r, dssum=True)
        p.set_flag("affine_geometry", 1)
        info = p.affine_info()
        assert info["fine_domain"] and info["sub_lists_affine"] == info["sub_lists"] == 1 and 0.0 <= info["max_deviation"] <= 64 * np.finfo(float).eps, info
        z1, zh1 = p.precond_apply(r, "gmres")
        assert np.abs(z1 - z0).max() <= 1e-12 * np.abs(z0).max() and np.abs(zh1 - zh0).max() <= 1e-12 * zh0[0]
        u1, its1, h1 = p.solve(f, "fcg")

        def pre(zz, rr):
            out, _, _ = sd.solve(rr[0], "gmres")
            zz[0][:] = out

        ou, oits, ohist = W.solve([f], "fcg", precond=pre)
        assert its1 == its0 == oits
        assert np.abs(h1 - ohist).max() <= 1e-8 * ohist[0] and np.abs(u1 - ou[0]).max() <= 1e-9 * np.abs(ou[0]).max()
        assert np.abs(u1 - u0).max() <= 1e-12 * np.abs(u0).max()
        # the float inner solve on the same option
        p.set_flag("preconditioner_precision", 32)
        u32, its32, h32 = p.solve(f, "fcg")
        p.set_flag("affine_geometry", 0)
        assert not p.affine_info()["fine_domain"] and p.affine_info()["sub_lists_affine"] == 0
        v32, jts32, g32 = p.solve(f, "fcg")
        assert its32 == jts32 and np.abs(u32 - v32).max() <= 1e-5 * np.abs(v32).max() and h32[-1] <= 1e-7 * h32[0] * 1.0001
        p.set_flag("preconditioner_precision", 64)
        assert np.array_equal(p.stiffness(r, dssum=True), q0)  # the point-space reference sequence never uses the option
    finally:
        W.close()
        sd.close()
        p.close()

    d = str(tmp_path / "curved")
    E, N, red = (3, 2, 2), 5, 4
    for deg in S.level_degrees(N, red):
        S.write_mesh_files(d, S.DeformedMesh(E, deg, 0.05))
    p = H.Problem.from_directory(d, N, red)
    p.set_flag("sub_use_preconditioner", 0)
    try:
        _, f = p.make_rhs_from(S.seeded_uniform(p.n, 99))
        u0, its0, h0 = p.solve(f, "fcg")
        p.set_flag("affine_geometry", 1)
        info = p.affine_info()
        assert not info["fine_domain"] and info["sub_lists_affine"] == 0 and info["max_deviation"] > 1e-6, info
        u1, its1, h1 = p.solve(f, "fcg")
        assert its1 == its0 and np.array_equal(u1, u0) and np.array_equal(h1, h0)
    finally:
        p.close()


def test_unit_stitch_in_place_and_early_gamma_are_bit_identical(setup):
    """One rank: every stitching weight of the dof slice is multiplicity * (1 / multiplicity) = 1.0 exactly, so the
    inner solve writes its correction straight into the outer iteration's z~ instead of a vector that is then
    multiplied by ones ("unit_stitch_in_place", on by default).  Same bits as the reference's sequence, double and
    float inner solve, with and without the V-cycle."""
    p = make_problem((4, 3, 3), 5, 4, True)
    try:
        _, f = p.make_rhs_from(S.seeded_uniform(p.n, 4321))
        for precision in (64, 32):
            p.set_flag("preconditioner_precision", precision)
            p.set_flag("unit_stitch_in_place", 1)
            u1, its1, h1 = p.solve(f, "fcg")
            p.set_flag("unit_stitch_in_place", 0)
            u0, its0, h0 = p.solve(f, "fcg")
            assert its1 == its0 and np.array_equal(u1, u0) and np.array_equal(h1, h0), precision
        p.set_flag("preconditioner_precision", 64)
        # the same for "early_gamma": the flexible dot also forms the next iteration's <z, r+> from the vectors it reads anyway
        # (the projection kernel is left with <p, q>): same sums in the same order
        p.set_flag("unit_stitch_in_place", 1)
        for lazy in (1, 0):
            p.set_flag("lazy_steps", lazy)
            p.set_flag("early_gamma", 1)
            u1, its1, h1 = p.solve(f, "fcg")
            p.set_flag("early_gamma", 0)
            u0, its0, h0 = p.solve(f, "fcg")
            assert its1 == its0 and np.array_equal(u1, u0) and np.array_equal(h1, h0), lazy
        p.set_flag("lazy_steps", 1)
        p.set_flag("early_gamma", 1)
        # and for "skip_last_basis_store": the last Arnoldi step of an inner cycle forms its vector's norm without storing it
        for precision in (64, 32):
            p.set_flag("preconditioner_precision", precision)
            p.set_flag("skip_last_basis_store", 1)
            u1, its1, h1 = p.solve(f, "fcg")
            z1, zh1 = p.precond_apply(f, "gmres")
            p.set_flag("skip_last_basis_store", 0)
            u0, its0, h0 = p.solve(f, "fcg")
            z0, zh0 = p.precond_apply(f, "gmres")
            assert its1 == its0 and np.array_equal(u1, u0) and np.array_equal(h1, h0) and np.array_equal(z1, z0) and np.array_equal(zh1, zh0), precision
        p.set_flag("preconditioner_precision", 64)
        p.set_flag("skip_last_basis_store", 1)
        # "shared_residual_norm": the outer residual norm and the inner solve's first norm are one sum over the dof slice,
        # formed once.  The iterates keep their bits; the recorded norms group their terms differently (a few ulp).
        p.set_flag("shared_residual_norm", 1)
        u1, its1, h1 = p.solve(f, "fcg")
        p.set_flag("shared_residual_norm", 0)
        u0, its0, h0 = p.solve(f, "fcg")
        assert its1 == its0 and np.array_equal(u1, u0) and np.abs(h1 - h0).max() <= 1e-13 * h0[0]
        p.set_flag("shared_residual_norm", 1)
        p.set_flag("sub_use_preconditioner", 1)
        assert p.amg_build(coarsest_size=40) >= 2
        p.set_flag("unit_stitch_in_place", 1)
        u1, its1, h1 = p.solve(f, "fcg")
        p.set_flag("unit_stitch_in_place", 0)
        u0, its0, h0 = p.solve(f, "fcg")
        assert its1 == its0 and np.array_equal(u1, u0) and np.array_equal(h1, h0)
    finally:
        p.close()
