#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W

metric : "PCG DOF-updates/sec + SpMV GB/s (%HBM peak), 3D Poisson N=7"
workload: config C2 per GPU -- 3-D Poisson, 32^3 spectral elements of degree
          N=7 per rank (16 777 216 GLL points / GPU), unit cube, homogeneous
          Dirichlet; outer flexible PCG (Domain::flexible_conjugate_gradient)
          preconditioned by the FDD subdomain solve (Subdomain::
          generalized_minimum_residual, 4 inner iterations, polynomial
          reduction 6) + stitching dssum.  N ranks = N rank blocks of the
          cube (2 -> 64x32x32, 4 -> 64x64x32, 8 -> 64^3 = config C4), one
          process per GPU: weak scaling.  With N > 1 `value` is the SAME
          iteration carried to N ranks: every rank preconditions with the
          solve on its own elements (block-local) and the interface is
          exchanged -- the no-V-cycle configuration that converges (at C4's
          size the composite without the V-cycle does not: DESIGN 5.2).  The
          full-domain-decomposition composite of every rank (own elements,
          rings of neighbour elements at degrees 7 and 1, graded superdomain)
          is built next to it and measured in `composite` (no V-cycle,
          flagged) and `reference_default` (V-cycle inside: what the
          reference runs, BASELINE C5).  --composite-headline swaps the two.
headline: `value` is config C2 as BASELINE.json quotes it -- inner GMRES(4)
          WITHOUT the low-order V-cycle inside.  The reference's default has
          the V-cycle on (subdomain.hpp:231); the same JSON line carries that
          configuration in `reference_default` (per step, and time to the
          reference's 1e-7 stopping tolerance for both), so that the cheap
          iterations of the headline cannot be read as the faster solver.
step    : one full outer PCG iteration (operator apply, gamma/theta dots,
          u/r update, assembled residual norm, preconditioner application,
          stitching, flexible dot, search update), vectors resident in HBM.
value   : unique global nodes x K / (max over ranks of the time of exactly K
          steps, bracketed by barrier + stream synchronise).
roofline: the device kernel family with the largest total time in the timed
          region, timed with HIP events on the rank's own stream; algorithmic
          bytes per BASELINE.md section 4.
cpu_baseline: the CPU oracle (serial restatement of the reference's
          OCCA-Serial path) timed on this box's host cores on a bounded sample
          by rank 0, at every N.  One rank: the CPU oracle's own solver on one
          core.  N ranks: one host core per rank (`cores: N`: the ranks are the
          threads of one child process that loads no GPU library), the restated
          kernels under the same host-layer solver (tests/cpu_baseline_ranks.py),
          on a smaller cube with the same rank grid.  A reported baseline, not
          a target.
legs    : every time-to-tolerance object carries `converged` (the reference's
          500-iteration cap can be hit first).  At N > 1 the line also carries
          the block-local comparison point (`block_local`) next to the
          composite, and at every N the headline configuration with
          point-Jacobi in the inner solver's preconditioner slot
          (`point_jacobi`, a labelled option of this build) and with the
          affine-elements option (`affine_geometry`: the six factor arrays
          formed in the kernel instead of streamed -- also a labelled option,
          never the headline).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--elements", type=int, default=32, help="elements per direction PER GPU (32 = config C2)")
    ap.add_argument("--degree", type=int, default=7)
    ap.add_argument("--reduction", type=int, default=6)
    ap.add_argument("--comm", choices=["torch", "rccl"], default="torch", help="N>1: torch.distributed(nccl=RCCL) callbacks, or RCCL called directly")
    ap.add_argument("--no-precond", action="store_true")
    ap.add_argument("--amg", action="store_true", help="make the reference's default preconditioner (low-order AMG V-cycle inside every inner GMRES step, config C5) the headline configuration")
    ap.add_argument("--no-reference-default", action="store_true", help="skip the `reference_default` object (V-cycle on: hierarchy build + solves to tolerance)")
    ap.add_argument("--no-time-to-tolerance", action="store_true", help="skip the full solves to 1e-7")
    ap.add_argument("--no-stencil", action="store_true", help="skip the 27-point-stencil SpMV (3.05e8 non-zeros at the default size)")
    ap.add_argument("--force-composite", action="store_true", help="N=1 diagnostic: run the one-rank problem through the composite code path (no rings, no superdomain) to see what that path costs on identical work")
    ap.add_argument("--block-local", action="store_true", help="N>1: block-local only (no composite is built: no `composite` / composite `reference_default` legs)")
    ap.add_argument("--composite-headline", action="store_true", help="N>1: make the full-domain-decomposition composite WITHOUT the V-cycle the headline `value` (rounds 1-2); default: the block-local iteration, which converges, with the composite in `composite` and `reference_default`")
    ap.add_argument("--no-amg-fusion", action="store_true", help="with --amg: the smoother's element-wise kernels as separate launches (the reference's sequence) instead of SpMV epilogues")
    ap.add_argument("--amg-precision", type=int, choices=[64, 32], default=64, help="the reference's `Float` (config.hpp:19-20, AMG/config.hpp:4): the preconditioner (inner Krylov solve and V-cycle) in double (default) or float")
    ap.add_argument("--no-amg-graph", action="store_true", help="with --amg: launch the V-cycle kernel by kernel (so that --kernel-table shows them) instead of replaying its hipGraph")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N>1 ranks all on cuda:0 with a gloo group staging device buffers over the host: exercises the multi-rank code path on a one-GPU box (not a measurement)")
    ap.add_argument("--rehearse-ranks", type=int, default=0, help="run N ranks as N host threads of THIS process on cuda:0 (no launcher; collectives are device-to-device copies between the ranks' buffers): exercises the N-rank code path, up to 2x2x2 = 8 ranks, on a one-GPU box (not a measurement)")
    ap.add_argument("--kernel-table", action="store_true", help="time every instrumented kernel family in the timed region (fills `kernels`; costs ~8 %% of a step)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record HIP events around the kernels of the timed region (no roofline object): how much the instrumentation costs")
    ap.add_argument("--outer", choices=["fcg", "gmres"], default="fcg", help="outer Krylov method of the timed steps: the metric's PCG (default) or the reference driver's own choice, flexible GMRES(20) (poisson.cpp:224 hard-codes solver_id = 1); `reference_default_gmres` is in the line either way")
    ap.add_argument("--mesh", choices=["box", "kershaw"], default="box", help="geometry of the whole run (default: the uniform box SURVEY 8(d) specifies); the Kershaw leg of the default run is in `kershaw` either way")
    ap.add_argument("--eps", type=float, default=0.3, help="Kershaw map parameter (run.py:25-47: eps_0.3)")
    ap.add_argument("--no-kershaw", action="store_true", help="skip the `kershaw` leg (a second problem on the deformed mesh)")
    ap.add_argument("--print-launch", action="store_true", help="print the launcher command `--gpus N` would start (JSON list) and exit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flag", action="append", default=[], metavar="NAME=VALUE", help="launch-sequence switch of the host layer (fddh_problem_set_flag) applied to every problem of the run, for A/B runs on one box: e.g. --flag fused_arnoldi_step=0")
    ap.add_argument("--cpu-sample-elements", type=int, default=None, help="elements per direction and rank of the CPU baseline's sample (default: the workload itself up to degree 7 -- 32 = config C2 --, 16 beyond: as many points)")
    ap.add_argument("--cpu-sample-steps", type=int, default=3, help="outer PCG iterations of the sample (one rank at C2: about 6 s each on the box's host)")
    return ap.parse_args()


def cpu_baseline(args, world, P):
    """Oracle timed on host cores: same solver structure, same rank grid, on a smaller cube.
    One rank: the oracle's own serial solver (single-subdomain preconditioner), one core.
    N ranks: ONE HOST CORE PER RANK, as BASELINE.md section 2 / SURVEY 8(d) ask (tests/cpu_baseline_ranks.py, a child
    process without any GPU library: every rank a host thread on the serial C restatement of the kernels behind the kernel
    C-ABI, the in-process communicator where MPI stood, the same host-layer solver and preconditioner as `value`); if that
    child run is not possible (no CPU build of the host layer on this box), the oracle's N-rank world simulated serially in
    this process on one core.  `cores` states what was really used."""
    if world > 1:
        try:
            return cpu_baseline_ranks(args, world, P)
        except Exception as exc:  # the fallback below says so in its sample text
            print("bench.py: per-rank CPU baseline unavailable (%s); serial simulation instead" % exc, file=sys.stderr, flush=True)
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support as S

    N, red = args.degree, args.reduction
    composite_sample = world > 1 and args.composite_headline and not args.block_local  # the oracle's composites cost a setup of their own: a smaller cube
    e = args.cpu_sample_elements if world == 1 else max(4, int(round((0.6 if composite_sample else 1.0) * min(args.cpu_sample_elements, 20) / world ** (1.0 / 3.0))))
    E = tuple(e * p for p in P)
    deg = S.level_degrees(N, red)
    t_setup = time.perf_counter()
    meshes = [[S.BoxMesh(E, d, P, r) for d in deg] for r in range(world)]
    W = S.OracleWorld([m[0] for m in meshes], N)
    F = sds = None
    if not args.no_precond:
        if world > 1 and args.composite_headline and not args.block_local:  # the configuration `value` is measured on
            F = S.OracleFdd(E, N, red, P, meshes=meshes)
        else:
            sds = [S.OracleSubdomain(None, N, red, meshes=meshes[r]) for r in range(world)]
    us = W.dssum([S.seeded_uniform(meshes[r][0].num_local_points, 1234 + r) for r in range(world)], True, True)
    f = W.stiffness(us)
    t_setup = time.perf_counter() - t_setup

    def pre(z, r):
        if F is not None:
            out, _ = F.precondition(r, "gmres")
            for k in range(world):
                z[k][:] = out[k]
        else:
            for k in range(world):
                out, _, _ = sds[k].solve(r[k], "gmres")
                z[k][:] = out

    steps = args.cpu_sample_steps
    t0 = time.perf_counter()
    _, its, _ = W.solve(f, "fcg", max_iterations=steps, tolerance=0.0, precond=None if args.no_precond else pre)
    dt = time.perf_counter() - t0
    nodes = meshes[0][0].global_nodes
    W.close()
    if F is not None:
        F.close()
    for sd in sds or []:
        sd.close()
    # orc_world_fcg runs `steps` full iterations plus the start-up residual norm
    # and first preconditioner application, all inside dt (slightly pessimistic)
    kind = "no preconditioner" if args.no_precond else ("full-domain-decomposition composite of every rank" if F is not None else "FDD preconditioner on every rank's own elements")
    return {
        "value": nodes * steps / dt,
        "unit": "DOF-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{E[0]}x{E[1]}x{E[2]} elements ({e}^3 per rank, {world} rank{'s' if world > 1 else ''} simulated serially in one process), N={N}, {steps} outer PCG iterations, "
                  f"{kind} (inner GMRES(4)), serial C oracle, {dt:.1f} s (+ {t_setup:.1f} s of oracle setup outside the clock)",
    }


def cpu_baseline_ranks(args, world, P):
    import subprocess

    composite = args.composite_headline and not args.block_local
    e = max(4, int(round((0.6 if composite else 1.0) * min(args.cpu_sample_elements, 20))))  # per rank: about 10 s of work on every core (N ranks: a 20^3 sample per rank at most, the N cores share the host's memory system)
    steps = args.cpu_sample_steps
    script = os.path.join(ROOT, "tests", "cpu_baseline_ranks.py")
    out = subprocess.run([sys.executable, script, str(world), str(e), str(args.degree), str(args.reduction), str(steps), "0" if composite else "1"], capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-400:])
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    E = r["elements"]
    return {
        "value": r["nodes"] * r["steps"] / r["seconds"],
        "unit": "DOF-updates/s",
        "cores": world,
        "kind": "port",
        "sample": f"{E[0]}x{E[1]}x{E[2]} elements ({e}^3 per rank), N={args.degree}, {steps} outer PCG iterations, {world} ranks = {world} host threads of one child process = {world} host cores (one per subdomain), "
                  f"{'full-domain-decomposition composite' if composite else 'block-local FDD preconditioner'} (inner GMRES(4)), the serial C restatement of the kernels under the host layer's solver, the in-process communicator in place of MPI, "
                  f"{r['seconds']:.1f} s (+ {r['setup_seconds']:.1f} s of setup outside the clock)",
    }


PMC_FILES = ["profiles/r04_pmc_traffic_c2.json", "profiles/r04_pmc_traffic_c3.json"]  # one per profiled workload (tools/profile_bench.sh at HEAD)


def pmc_traffic(kernel_key, elements, degree):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes
    (tools/profile_bench.sh: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this
    same command; FETCH_SIZE doubled on gfx950 per MI355X_MICROARCH.md).  None when the
    committed counters are for another workload or kernel."""
    pmc = source = None
    for name in PMC_FILES:
        path = os.path.join(ROOT, name)
        if not os.path.exists(path):
            continue
        with open(path) as fh:
            cand = json.load(fh)
        if cand.get("workload") == {"elements_per_gpu": elements, "degree": degree}:
            pmc, source = cand, name
            break
    if pmc is None:
        return None, None
    family, gather = kernel_key.split("<")[0], "<gather" in kernel_key
    best = None
    for name, st in pmc["kernels"].items():
        if not name.startswith(family):
            continue
        targs = [a.strip() for a in name[name.index("<") + 1:].rstrip("> ").split(",")]
        if family == "fused_stiffness_kernel":
            # fused_stiffness_kernel_t<T, n, kGather, kNTStore>: the double instance with the same gather flag
            if len(targs) < 3 or targs[0] != "double" or (targs[2] == "true") != gather:
                continue
        if family == "mfma_stiffness_kernel" and (len(targs) < 2 or (targs[1] == "true") != gather):
            continue  # mfma_stiffness_kernel<n, kGather>
        if best is None or st["launches"] > best["launches"]:
            best = st
    return (None, None) if best is None else (best["hbm_bytes_per_launch"], source)


def launch_command(args, argv):
    """The launcher line of an N-rank run: one rank per GPU of this node, rendezvous on 127.0.0.1 (the container's hostname
    may not resolve) at a port the launcher's own agent binds and KEEPS (--standalone: no port number is passed around)."""
    return [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            os.path.abspath(__file__)] + [a for a in argv if a != "--print-launch"]


def launch_ranks(args):
    import subprocess

    cmd = launch_command(args, sys.argv[1:])
    if args.print_launch:
        print(json.dumps(cmd))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or args.gpus) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


class _ThreadRanks:
    """max over the ranks of an in-process rehearsal (ranks are threads of this process)"""

    def __init__(self, n):
        import threading

        self.n, self.vals, self.barrier = n, [0.0] * n, threading.Barrier(n)

    def max(self, rank, x):
        self.vals[rank] = x
        self.barrier.wait()
        m = max(self.vals)
        self.barrier.wait()
        return m


def main():
    args = parse()
    if args.cpu_sample_elements is None:
        args.cpu_sample_elements = min(args.elements, 32 if args.degree <= 7 else 16)
    if (args.gpus > 1 or args.print_launch) and "RANK" not in os.environ and args.rehearse_ranks <= 1:
        # `python bench.py --gpus N` from a bare shell: this process launches its own ranks, as the reference's harness
        # does (run.py:160, `jsrun -n P -a 1 -g 1`) -- one CHILD process per GPU under torch.distributed.run (never an
        # exec: nothing here has touched the GPU, and nothing will), its output relayed, its exit code returned.
        sys.exit(launch_ranks(args))
    import torch

    if args.rehearse_ranks > 1:
        # N ranks as N host threads of THIS process, all on cuda:0, each with its own stream; collectives are
        # device-to-device copies between the ranks' buffers (host/comm.hpp LocalComm).  The multi-rank code path of a
        # rank -- composite, ring pull, coarse all-gather, interface exchange -- is the one RCCL serves on a node; the
        # timings are NOT a scaling measurement (the ranks share one GPU).  More than 6 ranks cannot be rehearsed as
        # processes on the pool's boxes (process guard), hence threads.
        from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H

        torch.cuda.set_device(0)
        R = args.rehearse_ranks
        args.gpus = R
        group = _ThreadRanks(R)
        H.run_local_ranks(R, lambda r, w: run(args, r, w, lambda x, r=r: group.max(r, x), "in-process rehearsal: %d ranks as threads sharing cuda:0 (LocalComm, device-to-device copies)" % w))
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world  # under a launcher the launcher's world size governs

    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H

    H.init(local_rank, use_torch_stream=True)
    H.set_print(False)
    if world == 1:
        H.comm_single()
    elif args.rehearse_on_one_gpu:
        H.comm_torch_callbacks(on_gpu=True, staged=True)
    elif args.comm == "rccl":
        H.comm_rccl_from_torch()
    else:
        H.comm_torch_callbacks(on_gpu=True)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    run(args, rank, world, max_over_ranks, "single" if world == 1 else ("gloo-staged rehearsal on one GPU" if args.rehearse_on_one_gpu else args.comm))
    if world > 1:
        dist.destroy_process_group()


def run(args, rank, world, max_over_ranks, comm_label):
    """one rank's bench: the communicator of the calling thread is set up"""
    import numpy as np
    import torch

    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
    from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import lib

    def progress(what):
        # a line per phase on stderr: long multi-rank runs stay visibly alive (and a hang can be placed)
        if rank == 0:
            print("[bench.py %6.1f s] %s" % (time.perf_counter() - t_start, what), file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    rehearsal = bool(world > 1 and (args.rehearse_ranks > 1 or args.rehearse_on_one_gpu))
    P = H.rank_grid(world)
    e = args.elements
    E = tuple(e * p for p in P)
    N = args.degree

    def create(block_local):
        """A failure on one rank leaves the others inside the composite's setup collectives: report it and leave with a
        failure code so that the launcher tears the job down, instead of moving on to a mismatched collective."""
        try:
            if args.mesh == "kershaw":
                made = H.Problem.kershaw(E, P, N, args.reduction, args.eps, with_subdomain=not args.no_precond, block_local=block_local, force_composite=args.force_composite and world == 1)
            else:
                made = H.Problem.box(E, P, N, args.reduction, with_subdomain=not args.no_precond, block_local=block_local, force_composite=args.force_composite and world == 1)
            for item in args.flag:
                name, _, value = item.partition("=")
                made.set_flag(name, int(value))
            return made
        except Exception as exc:
            print("bench.py rank %d: problem setup failed: %s" % (rank, exc), file=sys.stderr, flush=True)
            os._exit(1)

    t_setup = time.perf_counter()
    # N > 1: two preconditioners of the same outer iteration are built.
    #   block-local  every rank's own elements only -- the one-rank configuration (BASELINE C2) carried to N ranks
    #                unchanged, plus the interface exchange: the HEADLINE `value`, because without the V-cycle it is
    #                the one that converges (C4 size, 8 ranks: 379 iterations; the composite: residual 1.5e-4 after 500,
    #                DESIGN 5.2) and because its per-rank work is the one-rank work, so the N-rank curve measures the
    #                exchanges and not a bigger region;
    #   composite    the full-domain-decomposition region of every rank: carried in `composite` (no V-cycle, flagged)
    #                and in `reference_default` (V-cycle inside: the configuration the reference runs, BASELINE C5).
    # --composite-headline makes the composite without the V-cycle the headline, as in rounds 1-2.
    comp = blk = None
    if world > 1 and not args.no_precond:
        blk = create(True)
        if not args.block_local:
            comp = create(False)
        prob = comp if (args.composite_headline and comp is not None) else blk
    else:
        prob = create(args.block_local)
    progress("problem%s set up" % ("s (block-local and composite)" if comp is not None else ""))
    _, f = prob.make_rhs(function_id=4, seed=1234 + rank)  # rand()/RAND_MAX u*, poisson.cpp:211
    t_setup = time.perf_counter() - t_setup
    sub = prob.sub_info() if not args.no_precond else None
    composite = bool(sub and sub["is_composite"])
    comp_sub = comp.sub_info() if comp is not None else None

    amg_states = {}

    def configure(amg, precision=64, jacobi=False, problem=None):
        """headline: inner GMRES(4) alone; reference default: the low-order V-cycle inside every inner step;
        jacobi: point-Jacobi in that slot (labelled option of this build)"""
        if args.no_precond:
            return
        problem = problem or prob
        st = amg_states.setdefault(id(problem), {"levels": 0, "setup_s": 0.0})
        if amg and st["levels"] == 0:
            t = time.perf_counter()
            st["levels"] = problem.amg_build()
            st["setup_s"] = time.perf_counter() - t
            if args.no_amg_graph:
                problem.set_flag("amg_graph", 0)
            if args.no_amg_fusion:
                problem.set_flag("amg_fused_smoother", 0)
        problem.set_flag("sub_use_preconditioner", 1 if amg else (2 if jacobi else 0))
        # the reference's PTYPE = Float (config.hpp:19-20): the WHOLE inner solve (element stiffness, gather, Krylov vectors,
        # V-cycle) in double or in float
        problem.set_flag("preconditioner_precision", precision)

    info0 = prob.refresh()
    nodes = info0["num_total_nodes"]

    def timed_steps(steps, warmup, kernel_timing, problem=None, rhs=None):
        """`warmup` untimed + exactly `steps` timed outer PCG iterations bracketed by barrier + synchronise"""
        problem = problem or prob
        problem.pcg_begin(f if rhs is None else rhs)
        dominant = None
        if warmup > 0 and kernel_timing:
            # which kernel family dominates is MEASURED during the warm-up (every family timed); the timed region then
            # records HIP events around that family only (two records per launch cost a few microseconds)
            lib.host().call("fddh_profile_enable", 1)
            problem.pcg_steps(warmup)
            wbuf = ctypes.create_string_buffer(1 << 16)
            lib.host().call("fddh_profile_collect", wbuf, len(wbuf))
            wk = json.loads(wbuf.value.decode())
            if wk:
                dominant = max(wk, key=lambda k: wk[k]["ms"])
        else:
            problem.pcg_steps(warmup)
        lib.host().call("fddh_profile_enable", 1 if kernel_timing else 0)
        if dominant and not args.kernel_table:
            lib.host().call("fddh_profile_only", dominant.encode())
        H.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = problem.pcg_steps(steps)
        torch.cuda.synchronize()
        H.barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.host().call("fddh_profile_collect", buf, len(buf))
        lib.host().call("fddh_profile_enable", 0)
        return dt, last, json.loads(buf.value.decode())

    def timed_gmres_steps(steps, warmup, kernel_timing, problem=None, rhs=None):
        """`warmup` untimed + exactly `steps` timed Arnoldi steps of the outer flexible GMRES(20) (domain.tpp:727-914): a solve
        capped at `steps` iterations with the stopping test off, so the time holds the cycle's start (residual norm, first
        basis vector) and end (back-substitution, solution update) as a real solve pays them, bracketed as timed_steps is"""
        problem = problem or prob
        rhs = f if rhs is None else rhs
        problem.set_options(max_iterations=max(warmup, 1), tolerance=0.0)
        problem.solve_timed(rhs, "gmres")
        problem.set_options(max_iterations=steps, tolerance=0.0)
        lib.host().call("fddh_profile_enable", 1 if kernel_timing else 0)
        H.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its, hist, _ = problem.solve_timed(rhs, "gmres")
        torch.cuda.synchronize()
        H.barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.host().call("fddh_profile_collect", buf, len(buf))
        lib.host().call("fddh_profile_enable", 0)
        problem.set_options(max_iterations=500, tolerance=1e-7)  # the reference's values (domain.hpp:116-118)
        if its != steps:  # an exact zero or a NaN ended the cycle early: say so instead of dividing by steps that did not run
            print("bench.py: the outer GMRES ran %d of %d Arnoldi steps" % (its, steps), file=sys.stderr, flush=True)
            dt = dt * steps / max(its, 1)
        return dt, float(hist[-1]), json.loads(buf.value.decode())

    def to_tolerance(problem=None, rhs=None, method="fcg"):
        """the reference's own stopping rule: relative residual 1e-7 within 500 iterations (domain.hpp:116-118), clock
        around the device work; `converged` says which of the two ended the solve"""
        its, hist, sec = (problem or prob).solve_timed(f if rhs is None else rhs, method)
        sec = max_over_ranks(sec)
        rel = float(hist[-1] / hist[0]) if len(hist) else None
        return {"iterations": its, "converged": bool(rel is not None and rel <= 1e-7), "time_ms": sec * 1e3, "relative_residual": rel, "DOF_updates_per_s": nodes * its / sec if sec > 0 else None}

    # ---------------- headline configuration ----------------
    configure(args.amg, args.amg_precision)
    headline_steps = timed_gmres_steps if args.outer == "gmres" else timed_steps
    dt, last_res, kernels = headline_steps(args.steps, args.warmup, not args.no_kernel_timing)
    value = nodes * args.steps / dt
    progress("headline steps timed: %.3f ms per step" % (dt / args.steps * 1e3))

    # the same K steps with a stopping test (one host synchronisation) per step, as a real solve runs them
    if args.outer == "fcg":
        prob.set_flag("lazy_steps", 0)
        dt_tests, _, _ = timed_steps(args.steps, min(args.warmup, 1), False)
        prob.set_flag("lazy_steps", 1)
    else:
        dt_tests = dt  # a GMRES solve reads its residual estimate every step anyway
    headline_tol = None if args.no_time_to_tolerance else to_tolerance(method=args.outer)
    progress("headline solve to tolerance: %s" % (headline_tol,))

    # the headline configuration with the preconditioner in single precision (the reference's Float = float)
    headline_f32 = None
    if not args.no_precond and not args.no_reference_default and args.amg_precision == 64:
        configure(args.amg, 32)
        d32, lr32, _ = timed_steps(args.steps, min(args.warmup, 2), False)
        headline_f32 = {"ms_per_step": d32 / args.steps * 1e3, "value": nodes * args.steps / d32, "last_residual_norm": lr32}
        if not args.no_time_to_tolerance:
            headline_f32["to_1e-7"] = to_tolerance()
        configure(args.amg, args.amg_precision)
        progress("single-precision preconditioner leg done")

    # the headline configuration with point-Jacobi in the inner solver's preconditioner slot (labelled option of this
    # build; DESIGN 5: what four unpreconditioned Krylov steps lack is row scaling, most of all on the composite)
    point_jacobi = None
    if not args.no_precond and not args.no_reference_default and not args.amg:
        configure(False, args.amg_precision, jacobi=True)
        dj, lrj, _ = timed_steps(args.steps, min(args.warmup, 2), False)
        point_jacobi = {"preconditioner": "fdd_gmres4 + point-Jacobi in the inner preconditioner slot (option of this build, not in the reference)",
                        "ms_per_step": dj / args.steps * 1e3, "value": nodes * args.steps / dj, "last_residual_norm": lrj}
        if not args.no_time_to_tolerance:
            point_jacobi["to_1e-7"] = to_tolerance()
        configure(args.amg, args.amg_precision)

    progress("point-Jacobi leg done")

    # the headline configuration with the affine-elements option (a labelled option of this build, NOT the headline: the
    # reference always streams the six factor arrays, 48 of the stiffness kernel's 64 bytes per point).  Every element of
    # the bench's box mesh is an affine image of the reference cube, so its factors are six numbers per element times the
    # GLL weights; the option checks that on the mesh's own arrays and then forms them in the kernel instead of reading them.
    affine_leg = None
    if not args.no_precond and not args.no_reference_default and not args.amg:
        prob.set_flag("affine_geometry", 1)
        ainfo = prob.affine_info()
        if ainfo["fine_domain"] and ainfo["sub_lists_affine"] > 0:
            da, lra, _ = timed_steps(args.steps, min(args.warmup, 2), False)
            affine_leg = {"what": "headline configuration with element-wise constant (affine) geometry: the six factor arrays are not streamed (option of this build, not in the reference; results equal to rounding)",
                          "ms_per_step": da / args.steps * 1e3, "value": nodes * args.steps / da, "last_residual_norm": lra, "max_deviation_of_the_mesh_factors": ainfo["max_deviation"],
                          "sub_lists_affine": ainfo["sub_lists_affine"], "sub_lists": ainfo["sub_lists"]}
            if not args.no_time_to_tolerance:
                affine_leg["to_1e-7"] = to_tolerance()
            if args.amg_precision == 64:
                configure(args.amg, 32)
                d32a, _, _ = timed_steps(args.steps, min(args.warmup, 2), False)
                affine_leg["preconditioner_in_f32"] = {"ms_per_step": d32a / args.steps * 1e3, "value": nodes * args.steps / d32a}
                configure(args.amg, args.amg_precision)
        else:
            affine_leg = {"what": "the mesh's factor arrays are not of the affine form", "max_deviation_of_the_mesh_factors": ainfo["max_deviation"]}
        prob.set_flag("affine_geometry", 0)
    progress("affine-geometry leg done")

    def leg(problem, label):
        """the no-V-cycle configuration on the OTHER region of an N-rank run, next to the headline"""
        configure(False, 64, problem=problem)
        _, rhs = problem.make_rhs(function_id=4, seed=1234 + rank)
        d, lr, _ = timed_steps(args.steps, min(args.warmup, 2), False, problem=problem, rhs=rhs)
        out_leg = {"preconditioner": label, "ms_per_step": d / args.steps * 1e3, "value": nodes * args.steps / d, "last_residual_norm": lr}
        if not args.no_time_to_tolerance:
            out_leg["to_1e-7"] = to_tolerance(problem, rhs)
            configure(False, 64, jacobi=True, problem=problem)
            out_leg["point_jacobi_to_1e-7"] = to_tolerance(problem, rhs)
            configure(False, 64, problem=problem)
        return out_leg

    # N > 1: the other region's figures, so that neither is ever read without the other
    block_local_leg = composite_leg = None
    if comp is not None and not args.no_reference_default:
        if prob is comp:
            block_local_leg = leg(blk, "BLOCK-LOCAL fdd_gmres4 (own elements only: no neighbour rings / superdomain)")
        else:
            composite_leg = leg(comp, "full-domain-decomposition composite, fdd_gmres4 WITHOUT the V-cycle (not a configuration the reference runs: its use_preconditioner is hard-wired true)")
    if world == 1 and not args.no_precond and not args.amg:
        # identical keys at every N: on one rank the region IS the whole domain -- no rings, no
        # superdomain -- so the composite without the V-cycle is the headline configuration itself
        composite_leg = {"preconditioner": "one rank: the subdomain is the whole domain, the full-domain-decomposition composite without the V-cycle IS the headline configuration (same figures as `value`)",
                         "ms_per_step": dt / args.steps * 1e3, "value": value, "last_residual_norm": last_res}
        if headline_tol is not None:
            composite_leg["to_1e-7"] = headline_tol
    for lg in (composite_leg, block_local_leg):
        if lg is not None:
            lg["converged"] = lg["to_1e-7"]["converged"] if "to_1e-7" in lg else None
    progress("comparison leg done")
    table = {}
    for name, st in kernels.items():
        avg_ms = st["ms"] / st["count"]
        gbps = st["bytes"] / (st["ms"] * 1e-3) / 1e9
        table[name] = {"launches": st["count"], "avg_us": avg_ms * 1e3, "total_ms": st["ms"], "bytes_per_launch": st["bytes"] / st["count"], "GBps": gbps}

    # ---------------- the SpMV half of the metric (outside the timed region) ----------------
    spmv = {}
    if not args.no_stencil and world == 1:  # a per-GPU figure: measured in the single-GPU run
        # the general CSR case, SURVEY 8(d)(ii): 27-point stencil on the C2 node grid (225^3 rows, 3.05e8 non-zeros);
        # every byte of the formula moves here (values, columns, row pointers, x, y)
        m = e * N + 1
        if 27 * m**3 >= 2**31:  # int32 non-zero count (csr_matrix.tpp:132-134): larger node grids use SURVEY's 225^3
            m = 225
        us, nbytes, nnz = H.spmv_stencil_time(m, 40)  # 40 launches (32 ms): the average over 10 moved by 3 % from run to run on one box
        spmv["27-point stencil, %d^3 rows" % m] = {"frac_moved_of_hbm_peak": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "GBps_moved": nbytes / (us * 1e-6) / 1e9, "bytes_moved": nbytes, "avg_us": us, "nnz": nnz,
                                                  "kernel": "csr_block_kernel<EpiPlain>"}
    for which, label in ((0, "Q (scatter, 1 nnz/row)"), (1, "Qt (gather, 1-8 nnz/row)")):
        us, nbytes = ctypes.c_double(), ctypes.c_double()
        lib.host().call("fddh_problem_spmv_time", prob.h, which, 100, ctypes.byref(us), ctypes.byref(nbytes))
        # These two matrices are boolean with one entry per point, and the plan knows: the value array (and for Q the row
        # pointers, ptr[i] = i) are not read.  bytes_moved is what the kernels read and write (index 4 B + vector entries
        # 8 B, + 4 B row pointers for Qt) and is the figure to hold against the HBM peak; SURVEY 8(d)'s CSR formula
        # (12 nnz + 12 rows + 8 cols) counts bytes that never move and is kept only as `formula_*`.
        pts, nds = prob.info["num_local_points"], prob.info["num_local_nodes"]
        moved = (12.0 * pts + 8.0 * nds) if which == 0 else (12.0 * pts + 12.0 * nds)
        spmv[label] = {"frac_moved_of_hbm_peak": moved / (us.value * 1e-6) / 1e9 / HBM_PEAK_GBPS, "GBps_moved": moved / (us.value * 1e-6) / 1e9, "bytes_moved": moved, "avg_us": us.value,
                       "formula_bytes": nbytes.value, "formula_GBps": nbytes.value / (us.value * 1e-6) / 1e9}

    roofline = None
    if table:
        dom = max(table, key=lambda k: table[k]["total_ms"])
        traffic, traffic_file = (None, None) if (args.amg or args.no_precond or composite) else pmc_traffic(dom, e, N)  # the committed counters are for the single-rank workloads
        roofline = {
            "bound": "hbm",
            "kernel": dom,
            "achieved": table[dom]["GBps"],
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": table[dom]["GBps"] / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": None if traffic is None else traffic_file + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not re-measured in this run)",
            "launches": table[dom]["launches"],
            "avg_launch_us": table[dom]["avg_us"],
            "algorithmic_bytes_per_launch": table[dom]["bytes_per_launch"],
        }

    # ---------------- the reference's default inner preconditioner (V-cycle on): on the composite when there is one ----------------
    reference_default = reference_default_gmres = None
    ref_prob = comp if comp is not None else prob
    if not args.no_precond and not args.no_reference_default and not args.amg:
        reference_default = {"preconditioner": "fdd_gmres4 + low-order AMG V-cycle in every inner step (Subdomain::use_preconditioner = true, subdomain.hpp:231)"
                                               + (" on the full-domain-decomposition composite of every rank" if comp is not None else "")}
        _, f_ref = ref_prob.make_rhs(function_id=4, seed=1234 + rank)
        for precision in (64, 32):
            configure(True, precision, problem=ref_prob)
            d, lr, _ = timed_steps(args.steps, min(args.warmup, 2), False, problem=ref_prob, rhs=f_ref)
            entry = {"ms_per_step": d / args.steps * 1e3, "value": nodes * args.steps / d, "last_residual_norm": lr}
            if not args.no_time_to_tolerance:
                entry["to_1e-7"] = to_tolerance(ref_prob, f_ref)
            reference_default["f%d" % precision] = entry
            progress("reference-default leg (f%d) done" % precision)
        # ... and under the outer solver the reference's driver actually runs: flexible GMRES(20) (poisson.cpp:224, solver_id = 1;
        # domain.tpp:727-914), classical Gram-Schmidt with (j+1) assembled dots and (j+1) updates in Arnoldi step j
        reference_default_gmres = {"solver": "outer flexible GMRES(20) (poisson.cpp:224: the driver's hard-wired solver_id = 1) + " + reference_default["preconditioner"]}
        for precision in (64, 32):
            configure(True, precision, problem=ref_prob)
            d, lr, _ = timed_gmres_steps(args.steps, min(args.warmup, 2), False, problem=ref_prob, rhs=f_ref)
            entry = {"ms_per_arnoldi_step": d / args.steps * 1e3, "value": nodes * args.steps / d, "steps": args.steps, "last_residual_estimate": lr}
            if not args.no_time_to_tolerance:
                entry["to_1e-7"] = to_tolerance(ref_prob, f_ref, "gmres")
            reference_default_gmres["f%d" % precision] = entry
            progress("reference-default GMRES leg (f%d) done" % precision)
        reference_default["amg_levels"] = amg_states[id(ref_prob)]["levels"]
        reference_default["amg_setup_s"] = amg_states[id(ref_prob)]["setup_s"]
        configure(args.amg if ref_prob is prob else False, args.amg_precision, problem=ref_prob)

    # ---------------- the reference's own experiment geometry: the box under the Kershaw map (run.py:25-47, run.sh:30: eps_0.3) ----------------
    # Every factor array is full there (g4..g6 = 0 on the box), the operator is far worse conditioned, and the preconditioner
    # shows its real behaviour (the box converges in 3 iterations with the V-cycle inside).  One rank: a second problem of the
    # same size; N ranks: run the whole line with --mesh kershaw instead.
    kershaw_leg = None
    if world == 1 and args.mesh == "box" and not args.no_kershaw and not args.no_precond and not args.no_reference_default and not args.amg:
        kp = None
        try:
            kp = H.Problem.kershaw(E, P, N, args.reduction, args.eps, with_subdomain=True)
        except Exception as exc:
            kershaw_leg = {"error": str(exc)[:300]}
        if kp is not None:
            _, kf = kp.make_rhs(function_id=4, seed=1234 + rank)
            kp.set_flag("affine_geometry", 1)
            kdev = kp.affine_info()["max_deviation"]
            kp.set_flag("affine_geometry", 0)
            configure(False, 64, problem=kp)
            dk, lrk, _ = timed_steps(args.steps, min(args.warmup, 2), False, problem=kp, rhs=kf)
            kershaw_leg = {"mesh": f"{E[0]}x{E[1]}x{E[2]} elements under the Kershaw map, eps_y = eps_z = {args.eps} (host/box_mesh.hpp; the reference's meshes are Nek5000 exports of this case)",
                           "max_deviation_of_the_mesh_factors": kdev,
                           "headline": {"preconditioner": "fdd_gmres4 (no V-cycle)", "ms_per_step": dk / args.steps * 1e3, "value": nodes * args.steps / dk, "last_residual_norm": lrk}}
            if not args.no_time_to_tolerance:
                kershaw_leg["headline"]["to_1e-7"] = to_tolerance(kp, kf)
            progress("Kershaw leg: headline done")
            configure(True, 64, problem=kp)
            dk, lrk, _ = timed_steps(args.steps, min(args.warmup, 2), False, problem=kp, rhs=kf)
            kershaw_leg["reference_default"] = {"ms_per_step": dk / args.steps * 1e3, "value": nodes * args.steps / dk, "last_residual_norm": lrk, "amg_levels": amg_states[id(kp)]["levels"], "amg_setup_s": amg_states[id(kp)]["setup_s"]}
            if not args.no_time_to_tolerance:
                kershaw_leg["reference_default"]["to_1e-7"] = to_tolerance(kp, kf)
                kershaw_leg["reference_default"]["gmres_to_1e-7"] = to_tolerance(kp, kf, "gmres")
                configure(True, 32, problem=kp)
                dk32, _, _ = timed_steps(args.steps, 1, False, problem=kp, rhs=kf)  # also builds the float copies outside the solve's clock
                kershaw_leg["reference_default"]["f32_ms_per_step"] = dk32 / args.steps * 1e3
                kershaw_leg["reference_default"]["f32_to_1e-7"] = to_tolerance(kp, kf)
                kershaw_leg["reference_default"]["f32_gmres_to_1e-7"] = to_tolerance(kp, kf, "gmres")
            progress("Kershaw leg: reference default done")
            kp.close()

    info = prob.refresh()
    if args.no_precond:
        pre_name, pre_text = "none", "no preconditioner"
    else:
        pre_name = "fdd_gmres4+amg_vcycle(%d levels, f%d)" % (amg_states[id(prob)]["levels"], args.amg_precision) if args.amg else ("fdd_gmres4" if (composite or world == 1) else "fdd_gmres4 (block-local)")
        if composite:
            pre_text = "full-domain-decomposition preconditioner (per rank: %d own + %d ring/extended elements, %d superdomain dofs of %d coarse; inner GMRES(4), polynomial reduction %d)" % (
                info["num_local_elements"], sub["num_ext_elems"] - info["num_local_elements"], sub["sup_dofs"], sub["coarse_dofs"], args.reduction)
        elif world > 1:
            pre_text = "BLOCK-LOCAL FDD preconditioner (own elements only: no neighbour rings / superdomain; inner GMRES(4), polynomial reduction %d)" % args.reduction
        else:
            pre_text = "FDD preconditioner (single subdomain = whole domain, inner GMRES(4), polynomial reduction %d)" % args.reduction

    out = {
        "metric": "PCG DOF-updates/sec + SpMV GB/s (%HBM peak), 3D Poisson N=7",
        "value": value,
        "unit": "DOF-updates/s",
        # a rehearsal's ranks share ONE device: the structured fields say so, never an N-GPU weak-scaling point
        "n_gpus": 1 if rehearsal else world,
        "ranks": world,
        "rehearsal": rehearsal,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": None if rehearsal else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"3D Poisson, {E[0]}x{E[1]}x{E[2]} elements ({e}^3 per GPU), N={N}, " + ("flexible PCG" if args.outer == "fcg" else "flexible GMRES(20)") + " + " + pre_text
                        + ("" if args.mesh == "box" else f", Kershaw mesh eps = {args.eps}"),
            "outer": args.outer,
            "mesh": args.mesh,
            "elements": list(E),
            "rank_grid": list(P),
            "poly_degree": N,
            "points_per_gpu": info["num_local_points"],
            "unique_nodes": nodes,
            "preconditioner": pre_name,
            "comm": comm_label,
        },
        "points_updates_per_s": info["num_local_points"] * world * args.steps / dt,
        "last_residual_norm": last_res,
        "ms_per_step_with_stopping_tests": dt_tests / args.steps * 1e3,
        "to_1e-7": headline_tol,
        "preconditioner_in_f32": headline_f32,
        "point_jacobi": point_jacobi,
        "affine_geometry": affine_leg,
        "block_local": block_local_leg,
        "composite": composite_leg,
        "setup_s": t_setup,
        "roofline": roofline,
        "spmv": spmv,
        "reference_default": reference_default,
        "reference_default_gmres": reference_default_gmres,
        "kershaw": kershaw_leg,
        "kernels": table,
    }
    if world > 1:
        # the solve path's collectives alone, with the solve's own sizes and buffers (max over ranks)
        comm = (comp if comp is not None else prob).comm_time(20)
        out["comm_us"] = {k: {"avg_us": max_over_ranks(v["avg_us"]), "bytes": v["bytes"]} for k, v in comm.items()}
    if comp is not None:
        out["config"]["composite"] = {k: comp_sub[k] for k in ("num_elems", "num_ext_elems", "num_points", "sub_dofs", "sub_ext_dofs", "interface_dofs", "sup_dofs", "sup_ext_dofs", "unique_dofs", "coarse_dofs", "num_peers")}
        out["config"]["composite"]["superdomain_levels"] = comp.sub_composite_levels()

    progress("device legs done; CPU baseline")
    # rank 0 times the oracle on its host cores while the other ranks wait at the barrier below
    out["cpu_baseline"] = cpu_baseline(args, world, P) if (rank == 0 and not args.no_cpu_baseline) else None
    if world > 1:
        H.barrier()

    if rank == 0:
        print(json.dumps(out))
    for pr in (comp, blk):
        if pr is not None and pr is not prob:
            pr.close()
    prob.close()


if __name__ == "__main__":
    main()
