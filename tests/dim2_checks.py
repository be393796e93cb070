"""The reference's `dim == 2` branches end to end (domain.okl:22-31,80-86, subdomain.okl 2-D
restriction, domain.tpp:47,121-138): a deformed quadrilateral mesh written in the reference's
file format, read by the host layer, solved with the FDD preconditioner, against the oracle on
the same arrays.  Shared by the GPU test and the CPU-shim test of the host logic."""
import numpy as np

import support as S


def check_two_dimensional_solve(H, directory):
    E, N, red = (5, 4), 5, 2
    for deg in S.level_degrees(N, red):
        S.write_mesh_files(directory, S.QuadMesh(E, deg, amplitude=0.05))
    p = H.Problem.from_directory(directory, N, red)
    p.set_flag("sub_use_preconditioner", 0)  # the low-order hierarchy is built for 3-D regions only
    assert p.info["dim"] == 2 and p.n == E[0] * E[1] * (N + 1) ** 2
    for lvl in range(p.info["num_levels"]):
        p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
    meshes = [S.ArrayMesh.from_problem(p, lvl) for lvl in range(p.info["num_levels"])]
    W = S.OracleWorld([meshes[0]], N)
    sd = S.OracleSubdomain(None, N, red, meshes=meshes)
    try:
        assert np.abs(p.mesh_array("g_3")).max() > 1e-4  # the cross term is live
        assert p.info["num_total_nodes"] == (E[0] * N + 1) * (E[1] * N + 1)

        u = S.seeded_uniform(p.n, 8)
        assert np.array_equal(p.stiffness(u), W.stiffness([u])[0])
        assert np.array_equal(p.dssum(u, True, True), W.dssum([u], True, True)[0])
        assert np.array_equal(p.sub_op("tree", u), sd.tree(u))
        assert p.info["sub_num_values"] == sd.num_values == p.n
        assert np.array_equal(p.sub_op("stiffness", u), sd.stiffness(u))
        assert np.array_equal(p.sub_op("dssum", u), sd.dssum(u))

        u_star, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234))

        def pre(z, r):
            out, _, _ = sd.solve(r[0], "gmres")
            z[0][:] = out

        for method in ("fcg", "gmres"):
            x, its, hist = p.solve(f, method)
            ox, oits, ohist = W.solve([f], method, precond=pre)
            assert its == oits and np.abs(hist - ohist).max() <= 1e-8 * ohist[0], method
            assert np.abs(x - ox[0]).max() <= 1e-8 * np.abs(ox[0]).max(), method
            assert np.abs(x - u_star).max() <= 1e-4 * np.abs(u_star).max(), method
        return its
    finally:
        sd.close()
        W.close()
        p.close()
