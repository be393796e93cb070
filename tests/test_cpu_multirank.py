"""Multi-rank path on CPU: world_size-2 (and 4) `gloo` process groups drive the
product's C++ host layer -- rank partition, boundary-first numbering,
interface-slot exchange (pack -> all-reduce -> unpack), scalar all-reduces and
the torch.distributed communication callbacks -- exactly as bench.py does on
GPUs with backend "nccl" (= RCCL).

No GPU here, so the kernel C-ABI is served by tests/cpu_shim (the CPU oracle
behind include/fdd_hip.h: test infrastructure, never loaded by the product);
the host-layer sources are the product's own.  The reference result is the
oracle's R-rank world simulated in one process.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import support as S

SHIM_DIR = os.path.join(S.HERE, "cpu_shim")
HOST_CPU_SO = os.path.join(SHIM_DIR, "_build", "libfdd_host_cpu.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, E, N, red, with_sub, mesh_dir=None, composite=False, overlaps=(1, 1), reference_shaped=False, curved=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    # test-only: serve include/fdd_host.h from the CPU build of the host layer
    lib._host = lib._Lib(HOST_CPU_SO, os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H.init(0, use_torch_stream=False)
        H.set_print(False)
        H.comm_torch_callbacks(on_gpu=False)
        lib.host().call("fddh_comm_selftest", 1000)  # all-reduce sum/max, all-gather, all-gatherv, grouped send/receive, barrier

        Pg = S.rank_grid(world)

        def mesh_of(deg, r):
            # curved: an isoparametric deformation with all six geometric factors non-zero (sampled per degree, as a Nek5000 export is)
            if curved == "quad":  # the reference's dim == 2 branches: a deformed quadrilateral mesh cut into rank strips
                return S.QuadMeshRanks(E[:2], deg, Pg[:2], r, amplitude=0.04)
            return S.DeformedMesh(E, deg, 0.05, Pg, r) if curved else S.BoxMesh(E, deg, Pg, r)

        if mesh_dir:
            # the reference's per-rank input files (<dir>/lx1_<N+1>/<name>_<rank>.<N>.dat), one set per level degree
            for deg in (S.level_degrees(N, red) if with_sub else [N]):
                S.write_mesh_files(mesh_dir, mesh_of(deg, rank), proc_id=rank)
            dist.barrier()
            p = H.Problem.from_directory(mesh_dir, N, red, overlaps[0], overlaps[1], with_subdomain=with_sub, block_local=not composite)
        else:
            p = H.Problem.box(E, Pg, N, red, with_sub, overlaps[0], overlaps[1], block_local=not composite)
        if with_sub:
            p.set_flag("sub_use_preconditioner", 0)  # the inner solves of this test run without the V-cycle
        if reference_shaped:
            # the reference's own launch sequence: point-space Krylov vectors, the Qt / QQt_int / Q SpMV chain, host scalars
            for flag in ("assembled_inner_solve", "assembled_outer_solve", "restructured_inner_solve", "fused_dssum", "device_bookkeeping"):
                p.set_flag(flag, 0)
        for lvl in range(p.info["num_levels"]):
            p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])

        # oracle: all ranks simulated in this process, from the numpy mesh statement
        meshes = [mesh_of(N, r) for r in range(world)]
        W = S.OracleWorld(meshes, N)
        mine = meshes[rank]

        assert np.array_equal(p.mesh_array("glo_num"), mine.glo_num)
        assert np.array_equal(p.mesh_array("node_degree"), mine.node_degree)
        assert p.info["num_bdary_nodes"] == W.num_bdary(rank)
        assert p.info["num_local_nodes"] == W.num_nodes(rank)
        assert p.info["num_interface_slots"] == W.L.orc_world_num_interface_slots(W.w)
        assert p.info["num_total_nodes"] == mine.global_nodes
        # node numberings differ (the product moves Dirichlet nodes to the ends): compare through the points
        _, _, qc, _ = p.csr(0)
        oq = W.Q(rank)
        assert np.abs(p.assembled_weight()[qc] - W.assembled_weight(rank)[oq[1]]).max() == 0.0

        def field(mm):
            return np.sin(3 * mm.x + 1) * np.cos(2 * mm.y) + mm.z * mm.x

        us = [field(mm) for mm in meshes]
        for mask, weight in ((True, False), (True, True)):
            got = p.dssum(us[rank], mask, weight)
            ref = W.dssum(us, mask, weight)[rank]
            assert np.abs(got - ref).max() <= 1e-14 * np.abs(ref).max()
        ref = W.residual_norm(us)
        assert abs(p.residual_norm(us[rank]) - ref) <= 1e-13 * ref

        o_star = W.dssum(us, True, True)
        o_f = W.stiffness(o_star)
        u_star, f = p.make_rhs_from(us[rank])
        assert np.abs(f - o_f[rank]).max() <= 1e-13 * np.abs(o_f[rank]).max()

        if with_sub and composite:
            # the full-domain-decomposition composite of every rank against the oracle's R-rank world
            # (oracle/fdd_oracle_composite.c): region, sizes, operators on composite vectors, the tree exchange
            F = S.OracleFdd(E, N, red, Pg, overlaps[0], overlaps[1], meshes=[[mesh_of(d, r) for d in S.level_degrees(N, red)] for r in range(world)])
            si, oi = p.sub_info(), F.info[rank]
            assert si["is_composite"] == 1
            assert si["num_peers"] >= 1
            ids, lv = p.sub_region()
            oids, olv = F.region(rank)
            assert np.array_equal(ids, oids) and np.array_equal(lv, olv)
            assert len(set(lv.tolist())) > 1  # mixed degrees in the region
            for a, b in (("num_elems", "sub_elems"), ("num_ext_elems", "sub_ext_elems"), ("num_points", "points"), ("sub_dofs", "sub_dofs"), ("sub_ext_dofs", "sub_ext_dofs"), ("interface_dofs", "interface_dofs"),
                         ("sup_dofs", "sup_dofs"), ("sup_ext_dofs", "sup_ext_dofs"), ("unique_dofs", "unique_dofs"), ("coarse_dofs", "coarse_dofs"), ("num_values", "num_values")):
                assert si[a] == oi[b], (a, si[a], oi[b])
            assert p.sub_composite_levels() == F.composite_levels(rank)
            v = np.random.default_rng(5 + rank).standard_normal(si["num_values"])
            for op in ("stiffness", "dssum"):
                ref = getattr(F, op)(rank, v)
                assert np.abs(p.sub_op(op, v) - ref).max() <= 1e-12 * np.abs(ref).max(), op
            ref = F.residual_norm(rank, v)
            assert abs(p.sub_residual_norm(v) - ref) <= 1e-12 * ref
            ref = F.tree(us)[rank]
            assert np.abs(p.sub_op("tree", us[rank]) - ref).max() <= 1e-12 * np.abs(ref).max()
            for method in ("gmres", "fcg"):
                z, hist = p.precond_apply(us[rank], method)
                oz, oh = F.precondition(us, method)
                assert np.abs(z - oz[rank]).max() <= 1e-9 * np.abs(oz[rank]).max(), method
                assert np.abs(hist - oh[rank]).max() <= 1e-9 * oh[rank][0], method

            # the point-Jacobi option (labelled: not in the reference): the exact diagonal of the inner iteration's
            # operator against the oracle's (formed from Qt_int, Qt, A, Q, Q_int as they stand) and, for a sample of
            # dofs, against the operator itself applied to unit vectors; then the preconditioner application with it
            if not reference_shaped:
                to_oracle = S.composite_dof_permutation(p.sub_point_dofs(), F.point_dofs(rank), si["sub_dofs"], si["unique_dofs"])
                dj, od = p.sub_jacobi_diagonal(), F.jacobi_diagonal(rank)
                assert F.element_diagonal_check(rank) <= 1e-12
                assert np.abs(dj - od[to_oracle]).max() <= 1e-12 * np.abs(od).max()
                n_u = si["unique_dofs"]
                x = np.zeros(n_u)
                for d in [] if curved == "quad" else np.unique(  # 2-D regions run the reference-shaped point-space loops: no dof-space operator to probe
np.concatenate([np.arange(n_u - min(n_u, 40), n_u), np.random.default_rng(3).integers(0, n_u, 60), np.argsort(dj)[-20:]])
                ):
                    x[d] = 1.0
                    assert abs(p.sub_dof_operator(x)[d] - dj[d]) <= 1e-12 * np.abs(dj).max(), d
                    x[d] = 0.0
            p.set_flag("sub_use_preconditioner", 2)
            for method in ("gmres", "fcg"):
                z, hist = p.precond_apply(us[rank], method)
                oz, oh = F.precondition(us, method, use_preconditioner=2)
                assert np.abs(z - oz[rank]).max() <= 1e-9 * np.abs(oz[rank]).max(), ("jacobi", method)
                assert np.abs(hist - oh[rank]).max() <= 1e-9 * oh[rank][0], ("jacobi", method)
            p.set_flag("sub_use_preconditioner", 0)

            # the collectives of the solve path alone (what bench.py reports as comm_us): a collective call
            ct = p.comm_time(2)
            assert ct["ring_exchange"]["bytes"] > 0 and ct["coarse_allgather"]["bytes"] > 0 and ct["interface_pair_allreduce"]["bytes"] == 16 * p.info["num_interface_slots"]
            assert all(v["avg_us"] > 0 for v in ct.values())

            def pre(z, r):
                out, _ = F.precondition(r, "gmres")
                for k in range(world):
                    z[k][:] = out[k]
        else:
            sds = [S.OracleSubdomain(E, N, red, Pg, r) for r in range(world)] if with_sub else None

            def pre(z, r):
                for k in range(world):
                    out, _, _ = sds[k].solve(r[k], "gmres")
                    z[k][:] = out

        for method in ("fcg", "gmres"):
            u, its, hist = p.solve(f, method)
            ou, oits, ohist = W.solve(o_f, method, precond=pre if with_sub else None)
            assert its == oits, (method, its, oits)
            assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
            assert np.abs(u - ou[rank]).max() <= 1e-8 * np.abs(ou[rank]).max()
        p.close()
        W.close()
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def cpu_host_lib():
    subprocess.check_call(["make", "-C", S.ORACLE_DIR, "-s"])
    subprocess.check_call(["make", "-C", SHIM_DIR, "-s"])
    assert os.path.exists(HOST_CPU_SO)
    return HOST_CPU_SO


@pytest.mark.parametrize("world,with_sub", [(2, False), (2, True), (3, True), (4, False), (8, True)])  # 8 = 2x2x2: edges shared by 4 ranks, the centre node by 8 (the scaling run's topology)
def test_host_layer_multirank_gloo(cpu_host_lib, world, with_sub):
    import torch.multiprocessing as mp

    E, N, red = ((6, 4, 4) if world == 3 else (4, 4, 4)), 3, 2  # 3 ranks: the middle one has two interfaces (unequal boundary counts)
    mp.spawn(_worker, args=(world, _free_port(), E, N, red, with_sub), nprocs=world, join=True)


@pytest.mark.parametrize("world,E,N,red,overlaps", [
    (2, (8, 4, 4), 3, 2, (1, 1)),    # two levels; every coarse dof kept
    (2, (16, 4, 4), 3, 2, (1, 1)),   # the far superdomain is aggregated (two composite levels)
    (4, (8, 8, 4), 4, 2, (1, 1)),    # three levels (4, 2, 1): rings at every degree
    (8, (8, 8, 8), 3, 2, (1, 1)),    # 2x2x2: the scaling run's topology
    (2, (16, 4, 4), 3, 1, (1, 2)),   # reduction 1 (levels 3, 2, 1), superdomain overlap 2
    (2, (16, 4, 4), 3, 2, (1, 1, "reference-shaped")),  # the same through the reference's launch sequence (point-space vectors, SpMV chain)
    (2, (16, 4, 4), 3, 2, (2, 1)),   # two rings per polynomial level
    (2, (16, 4, 4), 3, 2, (2, 2)),   # both overlaps 2
    (3, (12, 4, 4), 3, 2, (1, 1)),   # a rank count that is not a power of two: the middle rank has two neighbours
    (3, (6, 2, 2), 3, 2, (1, 1)),    # the middle rank's rings cover the whole domain: it has NO superdomain while its peers do (the coarse all-gather must still be issued by all three)
])
def test_full_domain_decomposition_composite_gloo(cpu_host_lib, world, E, N, red, overlaps):
    """The composite of SURVEY 8(f) next-1 from the host layer under a gloo group -- neighbour rings at reduced
    degree, non-conforming Q, graded superdomain, interface maps, ring pull + coarse all-gather in tree_operator --
    against the oracle's restatement of subdomain.tpp:86-2747 for the same R ranks, then the outer solves
    preconditioned by it (identical iteration counts and histories)."""
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), E, N, red, True, None, True, overlaps[:2], len(overlaps) > 2), nprocs=world, join=True)


def test_composite_on_a_curved_mesh_from_files(cpu_host_lib, tmp_path):
    """The composite on a deformed mesh read from the reference's per-rank files: ring elements bring their owners'
    six geometric factors, the hanging rows interpolate in reference space, the superdomain operator is assembled from
    deformed degree-1 elements."""
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(2, _free_port(), (8, 4, 4), 3, 2, True, str(tmp_path / "curved"), True, (1, 1), False, True), nprocs=2, join=True)


@pytest.mark.parametrize("world,E", [(2, (16, 4, 1)), (4, (8, 8, 1))])
def test_composite_in_two_dimensions_from_files(cpu_host_lib, tmp_path, world, E):
    """The composite's `dim == 2` branches (quadrilaterals, hanging EDGES only, subdomain.tpp:1179-1585 with DIM 2):
    rank strips / a 2x2 rank grid of a deformed quadrilateral mesh read from files, against the oracle."""
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), E, 3, 2, True, str(tmp_path / "quad"), True, (1, 1), False, "quad"), nprocs=world, join=True)


def _amg_worker(rank, world, port, E, N, red):
    """The reference's DEFAULT inner preconditioner on the composite: the AMG V-cycle on the composite's low-order
    operator (subdomain.tpp:2749-3472: P1 elements on the GLL sub-cells of the mixed-degree region, piecewise-linear
    constraints on hanging edges / faces, the superdomain rows), inside the inner GMRES / CG of every rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    lib._host = lib._Lib(HOST_CPU_SO, os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H.init(0, use_torch_stream=False)
        H.set_print(False)
        H.comm_torch_callbacks(on_gpu=False)
        Pg = S.rank_grid(world)
        p = H.Problem.box(E, Pg, N, red, True)  # defaults: composite region, use_preconditioner = true
        for lvl in range(p.info["num_levels"]):
            p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
        si = p.sub_info()
        F = S.OracleFdd(E, N, red, Pg)

        # the low-order operator against the oracle's dense restatement; the hierarchy on top of it is this
        # build's own (HYPRE BoomerAMG in the reference), so the oracle's V-cycle runs on the product's levels
        assert p.amg_build(coarsest_size=40) >= 2
        levels = p.amg_levels()
        to_oracle = S.composite_dof_permutation(p.sub_point_dofs(), F.point_dofs(rank), si["sub_dofs"], si["unique_dofs"])
        olevels = S.permute_hierarchy(levels, to_oracle)
        Ao = F.low_order_matrix(rank)
        assert olevels[0]["A"].nnz == Ao.nnz
        assert abs(olevels[0]["A"] - Ao).max() <= 1e-12 * abs(Ao).max()
        assert abs(Ao - Ao.T).max() <= 1e-12 * abs(Ao).max()
        every = [None] * world
        dist.all_gather_object(every, olevels)
        for r in range(world):
            F.attach_amg(r, every[r])

        meshes = [S.BoxMesh(E, N, Pg, r) for r in range(world)]
        us = [np.sin(3 * m.x + 1) * np.cos(2 * m.y) + m.z * m.x for m in meshes]
        for method in ("gmres", "fcg"):
            z, hist = p.precond_apply(us[rank], method)
            oz, oh = F.precondition(us, method, use_preconditioner=True)
            assert np.abs(z - oz[rank]).max() <= 1e-9 * np.abs(oz[rank]).max(), method
            assert np.abs(hist - oh[rank]).max() <= 1e-9 * oh[rank][0], method
            _, plain = F.precondition(us, method, use_preconditioner=False)
            assert oh[rank][-1] < plain[rank][-1]  # the V-cycle helps

        W = S.OracleWorld(meshes, N)
        f = W.stiffness(W.dssum(us, True, True))

        def pre(z, r):
            out, _ = F.precondition(r, "gmres", use_preconditioner=True)
            for k in range(world):
                z[k][:] = out[k]

        for method in ("fcg", "gmres"):
            u, its, hist = p.solve(f[rank], method)
            ou, oits, ohist = W.solve(f, method, precond=pre)
            assert its == oits, (method, its, oits)
            assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
            assert np.abs(u - ou[rank]).max() <= 1e-8 * np.abs(ou[rank]).max()
        p.close()
        W.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,E,N,red", [(2, (16, 4, 4), 3, 2), (4, (8, 8, 4), 4, 2), (8, (8, 8, 8), 3, 2)])
def test_composite_with_low_order_preconditioner_gloo(cpu_host_lib, world, E, N, red):
    import torch.multiprocessing as mp

    mp.spawn(_amg_worker, args=(world, _free_port(), E, N, red), nprocs=world, join=True)


def _rod_worker(rank, world, port, mesh_dir, N, red, w, out_file):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import json

    import torch.distributed as dist

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    lib._host = lib._Lib(HOST_CPU_SO, os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H.init(0, use_torch_stream=False)
        H.set_print(False)
        if world > 1:
            H.comm_torch_callbacks(on_gpu=False)
        else:
            H.comm_single()
        E, Pg = (w * world, 2, 2), (world, 1, 1)
        for deg in S.level_degrees(N, red):
            S.write_mesh_files(mesh_dir, S.RodMesh(E, deg, Pg, rank), proc_id=rank)
        dist.barrier()
        res = {}
        for name, block_local in (("fdd", False), ("block_local", True)):
            p = H.Problem.from_directory(mesh_dir, N, red, 1, 1, True, block_local=block_local)  # the reference's defaults otherwise: GMRES(4) + V-cycle inside
            p.set_options(max_iterations=60)
            m = S.RodMesh(E, N, Pg, rank)
            _, f = p.make_rhs_from(np.sin(2 * m.x) + S.seeded_uniform(m.num_local_points, 77 + rank))
            _, its, hist = p.solve(f, "fcg")
            res[name] = [its, float(hist[-1] / hist[0])]
            p.close()
        if rank == 0:
            with open(out_file, "w") as fh:
                json.dump(res, fh)
    finally:
        dist.destroy_process_group()


def test_full_domain_decomposition_converges_where_block_local_stalls(cpu_host_lib, tmp_path):
    """Outer PCG iterations to 1e-7 on a rod (Dirichlet ends only) cut into 1, 2, 4, 8 rank strips of 4 elements,
    degree 2, everything else at the reference's defaults.  Block-local (own elements only) stops converging as
    soon as a rank's strip floats; the composite (rings at reduced degree + graded superdomain) keeps converging,
    its count growing slowly with the rod's length.  (On the cube configs with Dirichlet walls and <= 8 ranks both
    stay within 3-4 iterations: there the composite buys nothing yet.)"""
    import json

    import torch.multiprocessing as mp

    counts = {}
    for world in (1, 2, 4, 8):
        d = tmp_path / ("rod%d" % world)
        d.mkdir()
        out = str(tmp_path / ("rod%d.json" % world))
        mp.spawn(_rod_worker, args=(world, _free_port(), str(d), 2, 1, 4, out), nprocs=world, join=True)
        counts[world] = json.load(open(out))
    fdd = [counts[w]["fdd"][0] for w in (1, 2, 4, 8)]
    blk = [counts[w]["block_local"] for w in (1, 2, 4, 8)]
    assert fdd[0] == blk[0][0]                      # one rank: the same preconditioner
    assert all(counts[w]["fdd"][1] < 1e-7 for w in (1, 2, 4, 8))
    assert fdd[3] <= 6 * fdd[0]                     # 4 -> ~19 over an 8 times longer rod
    assert blk[2][0] == 60 and blk[2][1] > 1e-3     # 4 ranks: block-local has not converged after 60 iterations
    assert blk[3][0] == 60 and blk[3][1] > 1e-3


def _island_meshes(N):
    """Three ranks: ranks 0 and 1 share a cut box; rank 2 owns a separate piece that touches nobody (a mesh-file
    partition with a disconnected component), so its boundary-node prefix is EMPTY while its peers' is not."""
    a, b = S.BoxMesh((4, 2, 2), N, (2, 1, 1), 0), S.BoxMesh((4, 2, 2), N, (2, 1, 1), 1)
    c = S.BoxMesh((2, 2, 2), N)
    c.glo_num = np.where(c.glo_num > 0, c.glo_num + 10**6, c.glo_num)
    c.x = c.x + 2.0
    return [a, b, c]


def _island_worker(rank, world, port, mesh_dir, N):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    lib._host = lib._Lib(HOST_CPU_SO, os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H.init(0, use_torch_stream=False)
        H.set_print(False)
        H.comm_torch_callbacks(on_gpu=False)
        meshes = _island_meshes(N)
        S.write_mesh_files(mesh_dir, meshes[rank], proc_id=rank)
        dist.barrier()
        p = H.Problem.from_directory(mesh_dir, N, 2, 1, 1, with_subdomain=False)
        p.set_D_hat(0, S.gll(N)[2])
        W = S.OracleWorld(meshes, N)
        assert p.info["num_bdary_nodes"] == W.num_bdary(rank) and (p.info["num_bdary_nodes"] == 0) == (rank == 2)
        assert p.info["num_interface_slots"] == W.L.orc_world_num_interface_slots(W.w) > 0
        us = [np.sin(3 * m.x + 1) * np.cos(2 * m.y) + m.z * m.x for m in meshes]
        # every path that exchanges the interface: initial function, stiffness with dssum, residual norm, both outer solvers
        o_f = W.stiffness(W.dssum(us, True, True))
        _, f = p.make_rhs_from(us[rank])
        assert np.abs(f - o_f[rank]).max() <= 1e-13 * max(np.abs(o_f[r]).max() for r in range(world))
        for flag in (1, 0):
            p.set_flag("fused_dssum", flag)
            d = p.dssum(us[rank], True, True)
            assert np.abs(d - W.dssum(us, True, True)[rank]).max() <= 1e-14
        p.set_flag("fused_dssum", 1)
        for method in ("fcg", "gmres"):
            u, its, hist = p.solve(f, method)
            ou, oits, ohist = W.solve(o_f, method)
            assert its == oits and np.abs(hist - ohist).max() <= 1e-8 * ohist[0], method
            assert np.abs(u - ou[rank]).max() <= 1e-8 * max(np.abs(ou[r]).max() for r in range(world))
        p.close()
        W.close()
    finally:
        dist.destroy_process_group()


def test_rank_without_shared_nodes_issues_the_same_collectives(cpu_host_lib, tmp_path):
    """A rank whose partition touches no other rank still has to enter every interface all-reduce its peers enter
    (domain.tpp:590-594 calls gs on every rank); a mismatch would hang this test."""
    import torch.multiprocessing as mp

    mp.spawn(_island_worker, args=(3, _free_port(), str(tmp_path / "islands"), 3), nprocs=3, join=True)


def test_single_rank_cpu_shim_equals_oracle(cpu_host_lib):
    """The shim-backed host layer reproduces the oracle's C1 golden histories
    bit for bit (same kernels, same reduction tree): a check of the host
    layer's solver logic that needs no GPU."""
    code = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import support as S
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H, lib
lib._host = lib._Lib(%r, os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
H.init(0, use_torch_stream=False); H.comm_single(); H.set_print(False)
gold = json.load(open(os.path.join(S.GOLDEN_DIR, "oracle_c1.json")))
p = H.Problem.box((4, 4, 4), (1, 1, 1), 3, 2, True)
p.set_flag("sub_use_preconditioner", 0)
for lvl in range(p.info["num_levels"]):
    p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
m = S.ArrayMesh.from_problem(p)
W = S.OracleWorld([m], 3)
us = W.dssum([S.seeded_uniform(p.n, 1234)], True, True)
f = W.stiffness(us)[0]
for key, method, inner in (("fcg+gmres", "fcg", 1), ("gmres+fcg", "gmres", 0)):
    p.set_options(preconditioner_type=inner)
    u, its, hist = p.solve(f, method)
    g = gold["solves"][key]
    assert its == g["iterations"], (key, its, g["iterations"])
    assert np.abs(hist - np.array(g["history"])).max() <= 1e-9 * g["history"][0], key
# launch-sequence switches that leave every bit alone (DESIGN 4.3), one at a time against all on; and the affine option
p.set_options(preconditioner_type=1)
base = p.solve(f, "fcg")
for flag in ("unit_stitch_in_place", "early_gamma", "skip_last_basis_store"):
    p.set_flag(flag, 0)
    u, its, hist = p.solve(f, "fcg")
    assert its == base[1] and np.array_equal(u, base[0]) and np.array_equal(hist, base[2]), flag
    p.set_flag(flag, 1)
p.set_flag("shared_residual_norm", 0)  # the iterates keep their bits, the recorded norms group their terms differently
u, its, hist = p.solve(f, "fcg")
assert its == base[1] and np.array_equal(u, base[0]) and np.abs(hist - base[2]).max() <= 1e-13 * base[2][0]
p.set_flag("shared_residual_norm", 1)
p.set_flag("affine_geometry", 1)
info = p.affine_info()
assert info["fine_domain"] and info["sub_lists_affine"] == info["sub_lists"] == 1, info
u, its, hist = p.solve(f, "fcg")
assert its == base[1] and np.abs(u - base[0]).max() <= 1e-12 * np.abs(base[0]).max()
print("ok")
""" % (S.ROOT, S.HERE, HOST_CPU_SO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_host_layer_multirank_from_mesh_files(cpu_host_lib, tmp_path):
    """Two ranks, each reading its own files of the reference's mesh format (domain.tpp:45-224), then the same
    checks as above against the oracle's 2-rank world."""
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(2, _free_port(), (4, 4, 4), 3, 2, True, str(tmp_path / "mesh")), nprocs=2, join=True)


@pytest.mark.parametrize("world,E", [(4, "8,8,4"), (8, "8,8,8")])
def test_ranks_as_threads_of_one_process(cpu_host_lib, world, E):
    """The in-process communicator (host/comm.hpp LocalComm: N ranks = N host threads of one process, per-rank state
    thread_local, collectives as copies / rank-ordered sums between the ranks' buffers) on the composite against the
    oracle's N-rank world -- the form in which 4 and 8 ranks share the one GPU of the test box (tests/test_gpu_comm.py),
    here on the CPU stand-in of the kernel C-ABI.  A separate interpreter: the stand-in replaces the product library."""
    out = subprocess.run([sys.executable, os.path.join(S.HERE, "local_world_checks.py"), "--cpu-shim", str(world), E, "3", "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "local world ok" in out.stdout
