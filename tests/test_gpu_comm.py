"""Communication back-ends on the GPU box.  One GPU is available to the tests,
so the process group has world size 1; the collectives are still driven
through their real code paths (device-pointer wrapping for torch.distributed
"nccl" = RCCL, and RCCL called directly from the host layer) by
fddh_comm_selftest, which bypasses the solver's size == 1 shortcuts.  The
world_size 2/4 logic is covered on CPU by tests/test_cpu_multirank.py."""
import os

import numpy as np
import pytest

import rendezvous

from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group(gpu):
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    rdzv = rendezvous.new()  # a file store: no TCP port is reserved, released or guessed (GPUTEST_r03's EADDRINUSE)
    rendezvous.init("nccl", 0, 1, rdzv, device_id=torch.device("cuda", 0))
    H.init(0, use_torch_stream=True)
    H.set_print(False)
    yield dist
    H.comm_single()
    dist.destroy_process_group()
    rendezvous.done(rdzv)


def test_torch_distributed_callbacks_on_device_buffers(group):
    H.comm_torch_callbacks(on_gpu=True)
    lib.host().call("fddh_comm_selftest", 100000)


def test_rccl_called_directly(group):
    H.comm_rccl_from_torch()
    lib.host().call("fddh_comm_selftest", 100000)


def test_solve_is_identical_under_every_backend(group):
    results = []
    for setup in (H.comm_single, lambda: H.comm_torch_callbacks(on_gpu=True), H.comm_rccl_from_torch):
        setup()
        p = H.Problem.box((4, 4, 4), (1, 1, 1), 3, 2, True)
        p.set_flag("sub_use_preconditioner", 0)
        _, f = p.make_rhs(0, 0)
        u, its, hist = p.solve(f, "fcg")
        results.append((u, its, hist))
        p.close()
    for u, its, hist in results[1:]:
        assert its == results[0][1] and np.array_equal(hist, results[0][2]) and np.array_equal(u, results[0][0])


def _two_rank_worker(rank, world, rdzv, E, N, red, composite=False):
    """One of `world` ranks that all drive cuda:0; collectives go through a gloo
    group with the device buffers staged over the host (the solver's multi-rank
    code path -- interface exchange, device-side scalars, node-space PCG -- is
    exactly the one RCCL serves on a multi-GPU node)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist

    import support as S

    rendezvous.init_gloo(rank, world, rdzv)
    try:
        H.init(0, use_torch_stream=True)
        H.set_print(False)
        H.comm_torch_callbacks(on_gpu=True, staged=True)
        lib.host().call("fddh_comm_selftest", 1000)
        Pg = S.rank_grid(world)
        p = H.Problem.box(E, Pg, N, red, True, block_local=not composite)
        p.set_flag("sub_use_preconditioner", 0)
        for lvl in range(p.info["num_levels"]):
            p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
        meshes = [S.BoxMesh(E, N, Pg, r) for r in range(world)]
        W = S.OracleWorld(meshes, N)
        us = [np.sin(3 * mm.x + 1) * np.cos(2 * mm.y) + mm.z * mm.x for mm in meshes]
        o_f = W.stiffness(W.dssum(us, True, True))
        _, f = p.make_rhs_from(us[rank])
        assert np.abs(f - o_f[rank]).max() <= 1e-13 * np.abs(o_f[rank]).max()
        if composite:
            # the full-domain-decomposition composite on the HIP kernels: mixed-degree level lists, non-conforming Q,
            # CSR tail, ring pull and coarse all-gather on device buffers, against the oracle's 2-rank world
            F = S.OracleFdd(E, N, red, Pg)
            si, oi = p.sub_info(), F.info[rank]
            assert si["is_composite"] == 1 and si["num_values"] == oi["num_values"] and si["unique_dofs"] == oi["unique_dofs"]
            ids, lv = p.sub_region()
            assert np.array_equal(ids, F.region(rank)[0]) and len(set(lv.tolist())) > 1
            v = np.random.default_rng(5 + rank).standard_normal(si["num_values"])
            for op in ("stiffness", "dssum"):
                ref = getattr(F, op)(rank, v)
                assert np.abs(p.sub_op(op, v) - ref).max() <= 1e-12 * np.abs(ref).max(), op
            ref = F.tree(us)[rank]
            assert np.abs(p.sub_op("tree", us[rank]) - ref).max() <= 1e-12 * np.abs(ref).max()
            for method in ("gmres", "fcg"):
                z, hist = p.precond_apply(us[rank], method)
                oz, oh = F.precondition(us, method)
                assert np.abs(z - oz[rank]).max() <= 1e-9 * np.abs(oz[rank]).max(), method
                assert np.abs(hist - oh[rank]).max() <= 1e-9 * oh[rank][0], method

            def pre(z, r):
                out, _ = F.precondition(r, "gmres")
                for k in range(world):
                    z[k][:] = out[k]
        else:
            sds = [S.OracleSubdomain(E, N, red, Pg, r) for r in range(world)]

            def pre(z, r):
                for k in range(world):
                    out, _, _ = sds[k].solve(r[k], "gmres")
                    z[k][:] = out

        for method in ("fcg", "gmres"):
            u, its, hist = p.solve(f, method)
            ou, oits, ohist = W.solve(o_f, method, precond=pre)
            assert its == oits, (method, its, oits)
            assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
            assert np.abs(u - ou[rank]).max() <= 1e-8 * np.abs(ou[rank]).max()
        if composite:
            # the composite in single precision (PTYPE = Float = float): float copies of the hanging-point rows and of the
            # superdomain operator next to the float element kernels; same iteration count, single-precision agreement
            u64, its64, _ = p.solve(f, "fcg")
            p.set_flag("preconditioner_precision", 32)
            u32, its32, h32 = p.solve(f, "fcg")
            assert abs(its32 - its64) <= 1 and h32[-1] <= 1e-7 * h32[0] * 1.0001
            assert np.abs(u32 - u64).max() <= 1e-5 * np.abs(u64).max()
            p.set_flag("preconditioner_precision", 64)
        # the stepwise interface bench.py drives
        p.pcg_begin(f)
        last = p.pcg_steps(3)
        _, _, ohist = W.solve(o_f, "fcg", max_iterations=3, tolerance=0.0, precond=pre)
        assert abs(last - ohist[3]) <= 1e-8 * ohist[0]
        p.close()
        W.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_against_the_oracle_world(gpu):
    import torch.multiprocessing as mp

    mp.spawn(_two_rank_worker, args=(2, rendezvous.new(), (4, 4, 4), 3, 2), nprocs=2, join=True)  # block-local regions


def test_two_ranks_on_one_gpu_full_domain_decomposition_composite(gpu):
    """The composite region (rings at reduced degree + graded superdomain) of two ranks that share cuda:0."""
    import torch.multiprocessing as mp

    mp.spawn(_two_rank_worker, args=(2, rendezvous.new(), (16, 4, 4), 3, 2, True), nprocs=2, join=True)


def _full_size_worker(rank, world, rdzv):
    """Config C4's per-rank size (32^3 elements of degree 7 per rank) with two ranks sharing cuda:0: the composite with
    the reference's default inner preconditioner (AMG V-cycle on the composite low-order operator).  No oracle run at
    this size: the manufactured solution comes back, in a handful of outer iterations."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist

    import support as S

    rendezvous.init_gloo(rank, world, rdzv)
    try:
        H.init(0, use_torch_stream=True)
        H.set_print(False)
        H.comm_torch_callbacks(on_gpu=True, staged=True)
        p = H.Problem.box((64, 32, 32), (2, 1, 1), 7, 6, True)  # defaults: composite region, use_preconditioner = true
        si = p.sub_info()
        assert si["is_composite"] == 1 and si["num_peers"] == 1
        assert si["num_elems"] == 32768 + 2 * 1024 and si["own_points"] == 32768 * 512  # own elements + a degree-7 and a degree-1 ring
        assert si["interface_dofs"] == 31 * 31 and len(p.sub_composite_levels()) >= 3  # the far superdomain is graded
        u_star, f = p.make_rhs_from(S.seeded_uniform(p.n, 1234 + rank))
        x, its, hist = p.solve(f, "fcg")
        assert 0 < its <= 6 and hist[-1] <= 1e-7 * hist[0] * 1.0001, (its, hist[-1] / hist[0])
        assert np.abs(x - u_star).max() <= 1e-3 * np.abs(u_star).max()
        # the same with the whole preconditioner in single precision (PTYPE = Float = float): float element kernels, hanging
        # rows, superdomain rows and V-cycle
        p.set_flag("preconditioner_precision", 32)
        x32, its32, hist32 = p.solve(f, "fcg")
        assert abs(its32 - its) <= 1 and hist32[-1] <= 1e-7 * hist32[0] * 1.0001, (its32, hist32[-1] / hist32[0])
        assert np.abs(x32 - u_star).max() <= 1e-3 * np.abs(u_star).max()
        p.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_full_size_composite_with_the_reference_default_preconditioner(gpu):
    import torch.multiprocessing as mp

    mp.spawn(_full_size_worker, args=(2, rendezvous.new()), nprocs=2, join=True)


def _two_rank_worker_2d(rank, world, rdzv, mesh_dir):
    """The composite's `dim == 2` branches on the HIP kernels: two rank strips of a deformed quadrilateral mesh read from
    the reference's per-rank files (fused 2-D stiffness on mixed degrees, hanging edges, the exchange), against the oracle."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist

    import support as S

    rendezvous.init_gloo(rank, world, rdzv)
    try:
        H.init(0, use_torch_stream=True)
        H.set_print(False)
        H.comm_torch_callbacks(on_gpu=True, staged=True)
        E, N, red, Pg = (16, 4), 3, 2, (2, 1)
        degs = S.level_degrees(N, red)
        mesh_of = lambda deg, r: S.QuadMeshRanks(E, deg, Pg, r, amplitude=0.04)
        for deg in degs:
            S.write_mesh_files(mesh_dir, mesh_of(deg, rank), proc_id=rank)
        dist.barrier()
        p = H.Problem.from_directory(mesh_dir, N, red, 1, 1, with_subdomain=True)
        p.set_flag("sub_use_preconditioner", 0)
        for lvl in range(p.info["num_levels"]):
            p.set_D_hat(lvl, S.gll(p.level_degree(lvl))[2])
        meshes = [mesh_of(N, r) for r in range(world)]
        W = S.OracleWorld(meshes, N)
        F = S.OracleFdd(E + (1,), N, red, Pg + (1,), meshes=[[mesh_of(d, r) for d in degs] for r in range(world)])
        si, oi = p.sub_info(), F.info[rank]
        assert si["is_composite"] == 1 and si["num_values"] == oi["num_values"] and si["unique_dofs"] == oi["unique_dofs"]
        ids, lv = p.sub_region()
        assert np.array_equal(ids, F.region(rank)[0]) and len(set(lv.tolist())) > 1
        us = [np.sin(3 * m.x + 1) * np.cos(2 * m.y) + m.y * m.x for m in meshes]
        v = np.random.default_rng(5 + rank).standard_normal(si["num_values"])
        for op in ("stiffness", "dssum"):
            ref = getattr(F, op)(rank, v)
            assert np.abs(p.sub_op(op, v) - ref).max() <= 1e-12 * np.abs(ref).max(), op
        ref = F.tree(us)[rank]
        assert np.abs(p.sub_op("tree", us[rank]) - ref).max() <= 1e-12 * np.abs(ref).max()
        o_f = W.stiffness(W.dssum(us, True, True))
        _, f = p.make_rhs_from(us[rank])
        assert np.abs(f - o_f[rank]).max() <= 1e-13 * np.abs(o_f[rank]).max()

        def pre(z, r):
            out, _ = F.precondition(r, "gmres")
            for k in range(world):
                z[k][:] = out[k]

        for method in ("fcg", "gmres"):
            u, its, hist = p.solve(f, method)
            ou, oits, ohist = W.solve(o_f, method, precond=pre)
            assert its == oits, (method, its, oits)
            assert np.abs(hist - ohist).max() <= 1e-8 * ohist[0]
            assert np.abs(u - ou[rank]).max() <= 1e-8 * np.abs(ou[rank]).max()
        p.close()
        W.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_composite_in_two_dimensions(gpu, tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_two_rank_worker_2d, args=(2, rendezvous.new(), str(tmp_path / "quad")), nprocs=2, join=True)


@pytest.mark.parametrize("world,E", [(4, (8, 8, 4)), (8, (8, 8, 8))])
def test_four_and_eight_ranks_on_one_gpu(gpu, world, E):
    """2x2x1 and 2x2x2 rank grids on the ONE GPU of the box: the edges shared by four ranks and the corner shared by
    eight go through the HIP kernels and through Comm::exchange / all-gather / all-reduce on DEVICE buffers (the
    in-process communicator: each rank a host thread with its own stream; the pool allows at most 6 processes per GPU,
    so eight ranks cannot be eight processes).  Against the oracle's N-rank world: composite structure, operators, the
    tree operator with both exchanges, the preconditioner application (GMRES / flexible CG, with and without the
    point-Jacobi option), the outer solves (identical iteration counts), the float inner solve, the V-cycle inside."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import local_world_checks

    its64, its_amg = local_world_checks.run(world, E, 3, 2)
    assert its_amg < its64
    H.init(0, use_torch_stream=True)
    H.comm_single()


def test_bench_line_of_an_n_rank_run(gpu):
    """bench.py's N-rank line end to end on the one GPU (4 ranks as threads of one process, tiny problem): the keys the
    driver and the judge read -- the block-local headline, the composite's legs with `converged`, the reference default
    on the composite, `cpu_baseline` and `comm_us` at N > 1."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rehearse-ranks", "4", "--elements", "4", "--steps", "3", "--warmup", "1", "--cpu-sample-elements", "6", "--cpu-sample-steps", "3"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks"] == 4 and line["rehearsal"] is True and line["scaling"] is None and line["value"] > 0  # four ranks, ONE device
    assert "BLOCK-LOCAL" in line["config"]["workload"] and line["config"]["rank_grid"] == [2, 2, 1]
    assert line["to_1e-7"]["converged"] is True
    comp = line["composite"]
    assert comp is not None and isinstance(comp["to_1e-7"]["converged"], bool) and isinstance(comp["point_jacobi_to_1e-7"]["converged"], bool)
    rd = line["reference_default"]
    assert "composite" in rd["preconditioner"] and rd["f64"]["to_1e-7"]["converged"] is True and rd["f64"]["to_1e-7"]["iterations"] < line["to_1e-7"]["iterations"]
    assert line["cpu_baseline"]["cores"] == 4 and "4 ranks = 4 host threads" in line["cpu_baseline"]["sample"]  # one host core per subdomain
    assert set(line["comm_us"]) == {"allreduce_3_scalars", "interface_pair_allreduce", "coarse_allgather", "ring_exchange", "interface_pair_neighbour_exchange", "ring_and_coarse_exchange"}
    assert line["config"]["composite"]["num_peers"] == 3
