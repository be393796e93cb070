#!/usr/bin/env python3
"""Outer iterations on the rod of DESIGN 5.3 (Dirichlet ends only, strips of 4 elements of degree 2 per rank) for a given superdomain
overlap and number of inner Krylov steps: what the iteration growth with the rank count is made of.  Test infrastructure (CPU
stand-in of the kernel C-ABI under a gloo group).  python tests/rod_iteration_experiment.py <ranks> <superdomain overlap> <inner steps>"""
import os, sys, json, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
def worker(rank, world, rdzv, mesh_dir, N, red, w, sup_ov, inner, omega_env):
    import torch.distributed as dist
    import support as S
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H, lib
    lib._host = lib._Lib(os.path.join(ROOT, "tests/cpu_shim/_build/libfdd_host_cpu.so"), os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
    import rendezvous

    rendezvous.init_gloo(rank, world, rdzv)
    H.init(0, use_torch_stream=False); H.set_print(False)
    if world>1: H.comm_torch_callbacks(on_gpu=False)
    else: H.comm_single()
    E, Pg = (w*world, 2, 2), (world,1,1)
    for deg in S.level_degrees(N, red):
        S.write_mesh_files(mesh_dir, S.RodMesh(E, deg, Pg, rank), proc_id=rank)
    dist.barrier()
    p = H.Problem.from_directory(mesh_dir, N, red, 1, sup_ov, True)
    p.set_options(max_iterations=60, sub_num_vectors=inner, sub_max_iterations=inner)
    m = S.RodMesh(E, N, Pg, rank)
    _, f = p.make_rhs_from(np.sin(2*m.x) + S.seeded_uniform(m.num_local_points, 77+rank))
    _, its, hist = p.solve(f, "fcg")
    si = p.sub_info()
    if rank==0: print("world %d sup_overlap %d inner %d: its %d rel %.2e  sup_dofs %d of %d coarse, levels %s" % (world, sup_ov, inner, its, hist[-1]/hist[0], si["sup_dofs"], si["coarse_dofs"], p.sub_composite_levels()), flush=True)
    p.close(); dist.destroy_process_group()
if __name__=="__main__":
    import torch.multiprocessing as mp
    world=int(sys.argv[1]); sup=int(sys.argv[2]); inner=int(sys.argv[3])
    d=tempfile.mkdtemp()
    import rendezvous

    port = rendezvous.new()
    mp.spawn(worker,args=(world,port,d,2,1,4,sup,inner,None),nprocs=world,join=True)
