"""N ranks of ONE process (one host thread and stream each, sharing one GPU) on the full-domain-decomposition
composite, against the oracle's N-rank world: the in-process communicator of host/comm.hpp (LocalComm: collectives
as device-to-device copies / rank-ordered sums between the ranks' buffers).  2x2x1 puts the edges shared by four
ranks, 2x2x2 the corner shared by eight, through the HIP kernels and Comm::exchange on device buffers -- the pool
allows at most 6 processes on a GPU, so eight ranks cannot be eight processes there.

Used by tests/test_gpu_comm.py on the GPU (product libraries) and, with the CPU stand-in of the kernel C-ABI, by
tests/test_cpu_multirank.py as `python tests/local_world_checks.py --cpu-shim <world> <Ex,Ey,Ez> <N> <red>`.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def run(world, E, N, red, amg=True):
    import support as S
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    Pg = S.rank_grid(world)
    meshes = [S.BoxMesh(E, N, Pg, r) for r in range(world)]
    W = S.OracleWorld(meshes, N)
    F = S.OracleFdd(E, N, red, Pg)
    us = [np.sin(3 * mm.x + 1) * np.cos(2 * mm.y) + mm.z * mm.x for mm in meshes]
    # every oracle result up front, on this thread (the oracle's world is one mutable object)
    o_f = W.stiffness(W.dssum(us, True, True))
    ref = {"tree": F.tree(us), "v": [np.random.default_rng(5 + r).standard_normal(F.info[r]["num_values"]) for r in range(world)]}
    ref["stiffness"] = [F.stiffness(r, ref["v"][r]) for r in range(world)]
    ref["dssum"] = [F.dssum(r, ref["v"][r]) for r in range(world)]
    ref["norm"] = [F.residual_norm(r, ref["v"][r]) for r in range(world)]
    for mode in (0, 2):
        for method in ("gmres", "fcg"):
            ref[("pre", mode, method)] = F.precondition(us, method, use_preconditioner=mode)
    solves = {}
    for mode in (0, 2):
        def pre(z, r, mode=mode):
            out, _ = F.precondition(r, "gmres", use_preconditioner=mode)
            for k in range(world):
                z[k][:] = out[k]

        for method in (("fcg", "gmres") if mode == 0 else ("fcg",)):
            solves[(mode, method)] = W.solve(o_f, method, precond=pre)
    regions = [F.region(r) for r in range(world)]
    levels = [F.composite_levels(r) for r in range(world)]

    def body(rank, size):
        lib.host().call("fddh_comm_selftest", 1000)
        p = H.Problem.box(E, Pg, N, red, True)
        p.set_flag("sub_use_preconditioner", 0)
        for lvl in range(p.info["num_levels"]):
            p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
        _, f = p.make_rhs_from(us[rank])
        assert np.abs(f - o_f[rank]).max() <= 1e-13 * np.abs(o_f[rank]).max()
        si, oi = p.sub_info(), F.info[rank]
        assert si["is_composite"] == 1 and si["num_peers"] >= 1
        for a, b in (("num_elems", "sub_elems"), ("num_ext_elems", "sub_ext_elems"), ("num_points", "points"), ("sub_dofs", "sub_dofs"), ("sub_ext_dofs", "sub_ext_dofs"), ("interface_dofs", "interface_dofs"),
                     ("sup_dofs", "sup_dofs"), ("sup_ext_dofs", "sup_ext_dofs"), ("unique_dofs", "unique_dofs"), ("coarse_dofs", "coarse_dofs"), ("num_values", "num_values")):
            assert si[a] == oi[b], (a, si[a], oi[b])
        ids, lv = p.sub_region()
        assert np.array_equal(ids, regions[rank][0]) and np.array_equal(lv, regions[rank][1]) and len(set(lv.tolist())) > 1
        assert p.sub_composite_levels() == levels[rank]
        for op in ("stiffness", "dssum"):
            got = p.sub_op(op, ref["v"][rank])
            assert np.abs(got - ref[op][rank]).max() <= 1e-12 * np.abs(ref[op][rank]).max(), op
        assert abs(p.sub_residual_norm(ref["v"][rank]) - ref["norm"][rank]) <= 1e-12 * ref["norm"][rank]
        got = p.sub_op("tree", us[rank])  # ring pull by Comm::exchange + coarse all-gather on the ranks' device buffers
        assert np.abs(got - ref["tree"][rank]).max() <= 1e-12 * np.abs(ref["tree"][rank]).max()
        for mode in (0, 2):
            p.set_flag("sub_use_preconditioner", mode)
            for method in ("gmres", "fcg"):
                z, hist = p.precond_apply(us[rank], method)
                oz, oh = ref[("pre", mode, method)]
                assert np.abs(z - oz[rank]).max() <= 1e-9 * np.abs(oz[rank]).max(), (mode, method)
                assert np.abs(hist - oh[rank]).max() <= 1e-9 * oh[rank][0], (mode, method)
            for method in (("fcg", "gmres") if mode == 0 else ("fcg",)):
                u, its, hist = p.solve(f, method)
                ou, oits, ohist = solves[(mode, method)]
                assert its == oits, (mode, method, its, oits)
                assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
                assert np.abs(u - ou[rank]).max() <= 1e-8 * np.abs(ou[rank]).max()
        p.set_flag("sub_use_preconditioner", 0)
        its64 = solves[(0, "fcg")][1]
        # the whole inner solve in single precision (PTYPE = Float = float) on the composite
        p.set_flag("preconditioner_precision", 32)
        u32, its32, h32 = p.solve(f, "fcg")
        assert abs(its32 - its64) <= 1 and h32[-1] <= 1e-7 * h32[0] * 1.0001
        p.set_flag("preconditioner_precision", 64)
        # the affine-elements option on the composite: every level list of the region (own elements, rings at their reduced
        # degrees, the degree-1 far field) is checked on its own and runs without streaming its factor arrays
        p.set_flag("affine_geometry", 1)
        info = p.affine_info()
        assert info["fine_domain"] and info["sub_lists_affine"] == info["sub_lists"] >= 2 and info["max_deviation"] <= 64 * np.finfo(float).eps, info
        ua, itsa, ha = p.solve(f, "fcg")
        assert itsa == its64 and np.abs(ua - solves[(0, "fcg")][0][rank]).max() <= 1e-8 * np.abs(ua).max()
        p.set_flag("affine_geometry", 0)
        its_amg = None
        if amg:
            # the reference's default: the low-order V-cycle inside every inner step, hierarchy built by the host layer
            p.set_flag("sub_use_preconditioner", 1)
            assert p.amg_build(coarsest_size=40) >= 2
            u, its_amg, hist = p.solve(f, "fcg")
            assert hist[-1] <= 1e-7 * hist[0] and its_amg < its64
            assert np.abs(u - solves[(0, "fcg")][0][rank]).max() <= 1e-5 * np.abs(u).max()
            # and the whole preconditioner in float: every rank builds its f32 plans and captures its f32 V-cycle graph
            # while its peers are somewhere else in theirs (setup copies must not touch the legacy default stream)
            p.set_flag("preconditioner_precision", 32)
            u32a, its32a, h32a = p.solve(f, "fcg")
            assert abs(its32a - its_amg) <= 1 and h32a[-1] <= 1e-7 * h32a[0] * 1.0001
            p.set_flag("preconditioner_precision", 64)
        ct = p.comm_time(2)
        assert ct["ring_exchange"]["bytes"] > 0 and ct["coarse_allgather"]["bytes"] > 0
        p.close()
        return its64, its_amg

    out = H.run_local_ranks(world, body)
    W.close()
    F.close()
    assert len(set(out)) == 1, out  # every rank saw the same iteration counts
    return out[0]


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--cpu-shim":
        from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

        # test-only: serve include/fdd_host.h from the CPU build of the host layer (tests/cpu_shim)
        lib._host = lib._Lib(os.path.join(HERE, "cpu_shim", "_build", "libfdd_host_cpu.so"), os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
        args = args[1:]
    world = int(args[0])
    E = tuple(int(x) for x in args[1].split(","))
    print("local world ok:", run(world, E, int(args[2]), int(args[3])))
