"""Parity of the low-order AMG V-cycle preconditioner (Subdomain::
low_order_preconditioner, subdomain.tpp:3987-4159) between the product's host
layer and the oracle, shared by the CPU-shim test and the GPU test.

The hierarchy is the scipy-built stand-in for HYPRE's (support.low_order_
hierarchy); product and oracle get the same arrays.  Parity of results the
reference would produce with ITS hierarchy is unpinned: HYPRE is not available.

Tolerances: the V-cycle is SpMV + element-wise kernels; the product applies the
polynomial as A*(D*w) (device branch, subdomain.tpp:62-67) and a precomputed
inverse on the coarsest level where the oracle follows the host branch
(val*D*w; Gaussian elimination per application), so agreement is to rounding:
1e-11 of the result's max norm.
"""
import numpy as np

import support as S


def check_amg_f32(p, N, red):
    """`Float = float` (AMG/config.hpp:4): the V-cycle on f32 values and vectors against the oracle's float
    cycle (fdd_oracle_amg_f32.c, same statement order; the coarsest level is a precomputed inverse here and an
    elimination there), against the double cycle, and as the preconditioner of the solves.
    Tolerances are single-precision ones: 2e-5 of the result's max norm between the two float cycles,
    1e-4 between float and double."""
    meshes = [S.ArrayMesh.from_problem(p, lvl) for lvl in range(p.info["num_levels"])]
    dof, nd = p.sub_point_dofs(), p.info["sub_num_dofs"]
    levels = S.low_order_hierarchy(meshes[0], dof, nd)
    p.amg_attach(levels)
    o32 = S.OracleAmgF32(levels)
    try:
        r = S.seeded_uniform(p.n, 5) - 0.5
        z64 = p.amg_apply(r)
        p.set_flag("amg_precision", 32)
        z32 = p.amg_apply(r)
        # Qt r on the dofs (sum over a dof's points), the float cycle, Q back to the points
        has = dof >= 0
        f = np.zeros(nd)
        np.add.at(f, dof[has], r[has])
        u = o32.vcycle(f)
        ref = np.where(has, u[np.maximum(dof, 0)], 0.0)
        scale = np.abs(ref).max()
        assert scale > 0
        assert np.abs(z32 - ref).max() <= 2e-5 * scale, np.abs(z32 - ref).max() / scale
        assert np.abs(z32 - z64).max() <= 1e-4 * np.abs(z64).max()
        assert not np.array_equal(z32, z64)  # it really ran in float
        assert np.array_equal(z32.astype(np.float32).astype(np.float64), z32)  # the correction is float data cast up
        assert np.all(z32[~has] == 0.0)
        assert np.array_equal(p.amg_apply(r), z32)  # replay
        # switching back gives the double cycle again, bit for bit
        p.set_flag("amg_precision", 64)
        assert np.array_equal(p.amg_apply(r), z64)

        # the float cycle as preconditioner: the flexible solves converge as with the double one
        _, rhs = p.make_rhs_from(S.seeded_uniform(p.n, 1234))
        p.set_flag("sub_use_preconditioner", 1)
        p.set_options(preconditioner_type=1)
        u64, its64, h64 = p.solve(rhs, "fcg")
        p.set_flag("amg_precision", 32)
        u32, its32, h32 = p.solve(rhs, "fcg")
        assert abs(its32 - its64) <= 1, (its32, its64)
        assert h32[-1] <= 1e-7 * h32[0] * 1.0001
        assert np.abs(u32 - u64).max() <= 1e-5 * np.abs(u64).max()
        return its32
    finally:
        o32.close()


def check_amg(p, N, red, outer_solve=True, builder="scipy", cheby_order=2, num_vcycles=1):
    meshes = [S.ArrayMesh.from_problem(p, lvl) for lvl in range(p.info["num_levels"])]
    sd = S.OracleSubdomain(None, N, red, meshes=meshes)
    W = S.OracleWorld([meshes[0]], N)
    try:
        # product and oracle number the dofs differently (Domain node order vs the reference's
        # ranking of global ids): the same nodes carry dofs, and each side gets the hierarchy
        # built against its own numbering -- the same operators up to that permutation
        dof, odof = p.sub_point_dofs(), sd.point_dofs()
        assert np.array_equal(dof >= 0, odof >= 0)
        nd = p.info["sub_num_dofs"]
        assert nd == sd.num_dofs() == dof.max() + 1 == odof.max() + 1
        pair = {}
        for a, b in zip(dof[dof >= 0], odof[odof >= 0]):
            assert pair.setdefault(int(a), int(b)) == int(b)  # a bijection between the two numberings

        if builder == "scipy":
            levels = S.low_order_hierarchy(meshes[0], dof, nd)
        else:
            # the host layer's own FEM matrix + smoothed-aggregation hierarchy (host/low_order.hpp)
            # the reference's sweep parameters (subdomain.hpp:236-237, rewritten by run.py:153-155) as run-time switches
            p.set_flag("amg_cheby_order", cheby_order)
            p.set_flag("amg_num_vcycles", num_vcycles)
            assert p.amg_build(coarsest_size=40) >= 2
            levels = p.amg_levels(cheby_order)
        assert len(levels) >= 2 and levels[0]["A"].shape[0] == nd and levels[-1]["P"] is None
        # the oracle's copy: level 0 renumbered (same values, same Chebyshev data), coarse levels shared
        import scipy.sparse as sp

        to_oracle = np.array([pair[a] for a in range(nd)])
        Pm = sp.csr_matrix((np.ones(nd), (to_oracle, np.arange(nd))), shape=(nd, nd))
        fine = dict(levels[0])
        fine["A"] = (Pm @ levels[0]["A"] @ Pm.T).tocsr()
        fine["D"] = np.asarray(Pm @ levels[0]["D"])
        fine["P"] = (Pm @ levels[0]["P"]).tocsr()
        olevels = [fine] + levels[1:]
        if builder == "scipy":
            p.amg_attach(levels)
        sd.attach_amg(olevels, cheby_order=cheby_order, num_vcycles=num_vcycles)

        # one application of the V-cycle preconditioner
        r = S.seeded_uniform(p.n, 5) - 0.5
        z = p.amg_apply(r)
        oz = sd.low_order_preconditioner(r)
        assert np.abs(oz).max() > 0
        assert np.abs(z - oz).max() <= 1e-11 * np.abs(oz).max()
        if builder != "scipy":
            # a geometric level of 8 (or 16) lattice nodes per direction on a conforming 3-D region applies its interpolator
            # matrix-free (fdd_lattice_prolong / _restrict); the CSR interpolator gives the same cycle to rounding
            matrix_free = p.amg_level_transfer(0)
            assert matrix_free == (N in (7, 15) and meshes[0].dim == 3), (N, matrix_free)
            if matrix_free:
                p.set_flag("amg_matrix_free_transfer", 0)
                assert not p.amg_level_transfer(0)
                z_csr = p.amg_apply(r)
                p.set_flag("amg_matrix_free_transfer", 1)
                assert np.abs(z - z_csr).max() <= 1e-13 * np.abs(z_csr).max()
                assert np.array_equal(p.amg_apply(r), z)
                for bits in (32, 64):
                    p.set_flag("amg_precision", bits)
                    if bits == 32:
                        z32 = p.amg_apply(r)
                        p.set_flag("amg_matrix_free_transfer", 0)
                        z32_csr = p.amg_apply(r)
                        p.set_flag("amg_matrix_free_transfer", 1)
                        assert np.abs(z32 - z32_csr).max() <= 2e-5 * np.abs(z32_csr).max()
                        assert np.abs(z32 - z).max() <= 1e-4 * np.abs(z).max() and not np.array_equal(z32, z)
        # points without a dof get nothing; the operator is positive on assembled data
        assert np.all(z[dof < 0] == 0.0)
        assert float(np.dot(sd.dssum(r), z)) > 0.0
        # replay (hipGraph on the GPU): same bits as the first application
        assert np.array_equal(p.amg_apply(r), z)

        # inner solves preconditioned by it
        p.set_flag("sub_use_preconditioner", 1)
        for method in ("gmres", "fcg"):
            zz, hist = p.precond_apply(r, method)
            ozz, oits, ohist = sd.solve(r, method, use_preconditioner=True)
            assert len(hist) == len(ohist) == 5, (method, len(hist), len(ohist))
            assert np.abs(hist - ohist).max() <= 1e-9 * ohist[0], method
            assert np.abs(zz - ozz).max() <= 1e-9 * np.abs(ozz).max(), method
            # and it helps: 4 iterations get further than with the plain dssum
            _, _, plain = sd.solve(r, method, use_preconditioner=False)
            assert ohist[-1] < plain[-1], (method, ohist[-1], plain[-1])

        if outer_solve:
            _, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234))

            def pre(zv, rv):
                out, _, _ = sd.solve(rv[0], "gmres", use_preconditioner=True)
                zv[0][:] = out

            p.set_options(preconditioner_type=1)
            u, its, hist = p.solve(f, "fcg")
            ou, oits, ohist = W.solve([f], "fcg", precond=pre)
            assert its == oits and len(hist) == len(ohist), (its, oits)
            assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
            assert np.abs(u - ou[0]).max() <= 1e-8 * np.abs(ou[0]).max()
            return its
    finally:
        sd.close()
        W.close()
    return None
