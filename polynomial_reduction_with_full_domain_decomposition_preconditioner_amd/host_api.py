"""Python plumbing over libfdd_host.so (include/fdd_host.h): numpy vectors in
and out, communicator set-up from torch.distributed.  The solver itself is the
C++ host layer on the HIP kernels; nothing here computes."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import numpy as np

from . import lib

vp = ctypes.c_void_p

INFO_NAMES = [
    "num_local_points",
    "num_local_nodes",
    "num_bdary_nodes",
    "num_interface_slots",
    "num_total_nodes",
    "num_total_elements",
    "num_local_elements",
    "num_levels",
    "sub_num_values",
    "sub_num_dofs",
    "num_iterations",
    "dim",
]

_ALLREDUCE = ctypes.CFUNCTYPE(ctypes.c_int, vp, vp, ctypes.c_longlong)
_ALLGATHER = ctypes.CFUNCTYPE(ctypes.c_int, vp, vp, vp, ctypes.c_longlong)
_BARRIER = ctypes.CFUNCTYPE(ctypes.c_int, vp)
_EXCHANGE = ctypes.CFUNCTYPE(ctypes.c_int, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_longlong))

SUB_INFO_NAMES = [
    "is_composite",
    "num_elems",
    "num_ext_elems",
    "num_points",
    "sub_dofs",
    "sub_ext_dofs",
    "interface_dofs",
    "sup_dofs",
    "sup_ext_dofs",
    "unique_dofs",
    "coarse_dofs",
    "num_values",
    "own_points",
    "num_peers",
]

WITH_SUBDOMAIN, BLOCK_LOCAL, FORCE_COMPOSITE = 1, 2, 4

_keepalive = []


def _H():
    return lib.host()


def init(device: int = 0, use_torch_stream: bool = True) -> None:
    """Bind this process to a GPU.  With use_torch_stream the kernels run on
    torch's current stream, which orders them with torch.distributed
    collectives issued from the communication callbacks."""
    stream = None
    if use_torch_stream:
        import torch

        torch.cuda.set_device(device)
        stream = vp(torch.cuda.current_stream().cuda_stream)  # 0 = the default stream
    _H().call("fddh_init", device, stream, 0 if use_torch_stream else 1)


def set_print(on: bool) -> None:
    _H().call("fddh_set_print", int(on))


def comm_single() -> None:
    _H().call("fddh_comm_single")


def comm_rccl_from_torch() -> None:
    """RCCL called directly from the host layer; the 128-byte unique id is
    shipped through torch.distributed's store (no GPU traffic)."""
    import torch.distributed as dist

    rank, size = dist.get_rank(), dist.get_world_size()
    buf = ctypes.create_string_buffer(128)
    if rank == 0:
        _H().call("fddh_comm_rccl_unique_id", buf)
    obj = [bytes(buf.raw) if rank == 0 else None]
    dist.broadcast_object_list(obj, src=0)
    ident = ctypes.create_string_buffer(obj[0], 128)
    _H().call("fddh_comm_rccl_init", ident, rank, size)


def local_world(size: int):
    """Handle of an in-process world of `size` ranks (one host thread each, one shared GPU): see comm_local."""
    w = vp()
    _H().call("fddh_local_world_create", ctypes.byref(w), int(size))
    return w


def local_world_destroy(world) -> None:
    _H().call("fddh_local_world_destroy", world)


def comm_local(world, rank: int) -> None:
    """This THREAD becomes rank `rank` of the in-process world (call init(..., use_torch_stream=False) first: the
    rank's own stream).  Collectives are device-to-device copies / sums between the ranks' buffers."""
    _H().call("fddh_comm_local", world, int(rank))


def run_local_ranks(size: int, fn, device: int = 0, init_device: bool = True):
    """Run fn(rank, size) on `size` threads of this process, each bound to the GPU as one rank of a local world.
    Returns the list of results; the first exception of any rank is re-raised."""
    import threading

    # the ranks share this process's cores: each one's setup threads (host/host_parallel.hpp) take their share
    shared_cores = "FDD_HOST_THREADS" not in os.environ and size > 1
    if shared_cores:
        os.environ["FDD_HOST_THREADS"] = str(max(1, (os.cpu_count() or size) // size))
    world = local_world(size)
    results, errors = [None] * size, [None] * size

    def body(r):
        try:
            if init_device:
                init(device, use_torch_stream=False)
            set_print(False)
            comm_local(world, r)
            results[r] = fn(r, size)
        except BaseException as exc:  # noqa: BLE001 - reported to the caller below
            errors[r] = exc
            try:
                _H().call("fddh_local_world_fail", world)  # the peers waiting for this rank get an error now, not after the timeout
            except Exception:  # noqa: BLE001
                pass
        finally:
            try:
                _H().call("fddh_rank_finalize")  # the rank's own stream and communicator end with its thread
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=body, args=(r,), name="fdd-rank-%d" % r) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    local_world_destroy(world)
    if shared_cores:
        os.environ.pop("FDD_HOST_THREADS", None)
    first = [e for e in errors if e is not None and "a peer rank failed" not in str(e)] or [e for e in errors if e is not None]
    if first:
        raise first[0]  # the rank that failed first, not a peer's "a peer rank failed"
    return results


class _DevView:
    """Minimal __cuda_array_interface__ holder so torch can wrap a raw device pointer."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def comm_torch_callbacks(on_gpu: bool = True, staged: bool = False) -> None:
    """Collectives through torch.distributed (backend "nccl" = RCCL on GPUs,
    "gloo" on CPU buffers in the CPU test build).  staged: device buffers go
    through a host copy around the collective (a gloo group driving GPU ranks:
    the one-GPU rehearsal of the multi-rank path)."""
    import torch
    import torch.distributed as dist

    rank, size = dist.get_rank(), dist.get_world_size()

    views = {}  # (pointer, length, dtype) -> tensor view: the solve's exchange buffers are the same few on every call

    def wrap(ptr, n, dtype):
        key = (int(ptr or 0), int(n), dtype)
        t = views.get(key)
        if t is not None:
            return t
        if on_gpu:
            typestr = {torch.float64: "<f8", torch.uint8: "|u1"}[dtype]
            t = torch.as_tensor(_DevView(ptr, (n,), typestr), device="cuda")
        else:
            ctype = {torch.float64: ctypes.c_double, torch.uint8: ctypes.c_uint8}[dtype]
            t = torch.from_numpy(np.ctypeslib.as_array((ctype * n).from_address(ptr)))
        if len(views) < 4096:
            views[key] = t
        return t

    def allreduce(op):
        def fn(ctx, buf, n):
            try:
                t = wrap(buf, int(n), torch.float64)
                if staged:
                    h = t.cpu()
                    dist.all_reduce(h, op=op)
                    t.copy_(h)
                else:
                    dist.all_reduce(t, op=op)
                return 0
            except Exception as exc:  # pragma: no cover - surfaced by the C++ side
                print("allreduce callback failed:", exc, flush=True)
                return 1

        return fn

    def allgather(ctx, send, recv, nbytes):
        try:
            nbytes = int(nbytes)
            s = wrap(send, nbytes, torch.uint8)
            r = wrap(recv, nbytes * size, torch.uint8)
            if staged:
                hs = s.cpu()
                hr = [torch.empty_like(hs) for _ in range(size)]
                dist.all_gather(hr, hs)
                r.copy_(torch.cat(hr))
            elif on_gpu:
                dist.all_gather_into_tensor(r, s)
            else:
                dist.all_gather(list(r.chunk(size)), s)
            return 0
        except Exception as exc:  # pragma: no cover
            print("allgather callback failed:", exc, flush=True)
            return 1

    def barrier(ctx):
        try:
            dist.barrier()
            return 0
        except Exception as exc:  # pragma: no cover
            print("barrier callback failed:", exc, flush=True)
            return 1

    def exchange(ctx, n, peers, send, send_bytes, recv, recv_bytes):
        # the composite's ring pull: one message per peer and direction, all in flight together
        try:
            ops, staged_recv = [], []
            for i in range(int(n)):
                peer, ns, nr = int(peers[i]), int(send_bytes[i]), int(recv_bytes[i])
                if ns:
                    t = wrap(send[i], ns, torch.uint8)
                    ops.append(dist.P2POp(dist.isend, t.cpu() if staged else t, peer))
                if nr:
                    t = wrap(recv[i], nr, torch.uint8)
                    if staged:
                        h = torch.empty(nr, dtype=torch.uint8)
                        staged_recv.append((t, h))
                        t = h
                    ops.append(dist.P2POp(dist.irecv, t, peer))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for t, h in staged_recv:
                t.copy_(h)
            return 0
        except Exception as exc:  # pragma: no cover
            print("exchange callback failed:", exc, flush=True)
            return 1

    cbs = (_ALLREDUCE(allreduce(dist.ReduceOp.SUM)), _ALLREDUCE(allreduce(dist.ReduceOp.MAX)), _ALLGATHER(allgather), _BARRIER(barrier), _EXCHANGE(exchange))
    _keepalive.append(cbs)
    _H().call("fddh_comm_callbacks_ex", rank, size, None, *[ctypes.cast(c, vp) for c in cbs])


def rank_grid(num_ranks: int):
    """Same rule as the C++ driver: powers of two go round-robin over x, y, z."""
    P = [1, 1, 1]
    d = 0
    while num_ranks > 1 and num_ranks % 2 == 0:
        P[d] *= 2
        num_ranks //= 2
        d = (d + 1) % 3
    P[0] *= num_ranks
    return tuple(P)


def _arr3(v: Sequence[int]):
    return (ctypes.c_int * 3)(*[int(x) for x in v])


def _dp(a: Optional[np.ndarray]):
    if a is None:
        return vp(0)
    assert a.flags["C_CONTIGUOUS"]
    return vp(a.ctypes.data)


class Problem:
    """One rank's Domains (levels N, N-r, ..., 1) and optional Subdomain."""

    def __init__(self, handle):
        self.h = handle
        self.info = self._info()
        self.n = self.info["num_local_points"]

    @staticmethod
    def _flags(with_subdomain, block_local, force_composite):
        return (WITH_SUBDOMAIN if with_subdomain else 0) | (BLOCK_LOCAL if block_local else 0) | (FORCE_COMPOSITE if force_composite else 0)

    @classmethod
    def box(cls, E, P=(1, 1, 1), poly_degree=7, poly_reduction=2, with_subdomain=True, subdomain_overlap=1, superdomain_overlap=1, block_local=False, force_composite=False):
        """With more than one rank the Subdomain is the full-domain-decomposition composite (own elements, rings at
        reduced degree, coarsened superdomain); block_local keeps the rank's own elements only."""
        h = vp()
        _H().call("fddh_problem_create_box_ex", ctypes.byref(h), _arr3(E), _arr3(P), poly_degree, poly_reduction, subdomain_overlap, superdomain_overlap, cls._flags(with_subdomain, block_local, force_composite))
        return cls(h)

    @classmethod
    def kershaw(cls, E, P=(1, 1, 1), poly_degree=7, poly_reduction=2, eps=0.3, with_subdomain=True, subdomain_overlap=1, superdomain_overlap=1, block_local=False, force_composite=False, eps_z=None):
        """The box under the generalized Kershaw map (the reference's experiment geometry, run.py:25-47: eps = 0.3); all
        six geometric factors are non-zero.  eps = 1 is the uniform box."""
        h = vp()
        _H().call("fddh_problem_create_kershaw_ex", ctypes.byref(h), _arr3(E), _arr3(P), poly_degree, poly_reduction, subdomain_overlap, superdomain_overlap,
                  cls._flags(with_subdomain, block_local, force_composite), ctypes.c_double(eps), ctypes.c_double(eps if eps_z is None else eps_z))
        return cls(h)

    @classmethod
    def from_directory(cls, directory, poly_degree, poly_reduction, subdomain_overlap=1, superdomain_overlap=1, with_subdomain=True, block_local=False, force_composite=False):
        h = vp()
        _H().call("fddh_problem_create_dir_ex", ctypes.byref(h), os.fsencode(directory), poly_degree, poly_reduction, subdomain_overlap, superdomain_overlap, cls._flags(with_subdomain, block_local, force_composite))
        return cls(h)

    def sub_info(self):
        buf = (ctypes.c_longlong * len(SUB_INFO_NAMES))()
        _H().call("fddh_problem_sub_info", self.h, buf, len(SUB_INFO_NAMES))
        return {k: int(buf[i]) for i, k in enumerate(SUB_INFO_NAMES)}

    def sub_region(self):
        n = self.sub_info()["num_ext_elems"]
        ids, lv = np.zeros(n, np.int32), np.zeros(n, np.int32)
        ip = ctypes.POINTER(ctypes.c_int)
        _H().call("fddh_problem_sub_region", self.h, ids.ctypes.data_as(ip), lv.ctypes.data_as(ip), n)
        return ids, lv

    def sub_composite_levels(self):
        kept = np.zeros(32, np.int32)
        nl = ctypes.c_int()
        _H().call("fddh_problem_sub_composite_levels", self.h, kept.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), 32, ctypes.byref(nl))
        return [int(k) for k in kept[: nl.value]]

    def close(self):
        if self.h:
            _H().call("fddh_problem_destroy", self.h)
            self.h = None

    def _info(self):
        buf = (ctypes.c_longlong * len(INFO_NAMES))()
        _H().call("fddh_problem_info", self.h, buf, len(INFO_NAMES))
        return {k: int(buf[i]) for i, k in enumerate(INFO_NAMES)}

    def refresh(self):
        self.info = self._info()
        return self.info

    def level_degree(self, level):
        d = ctypes.c_int()
        _H().call("fddh_problem_level_degree", self.h, level, ctypes.byref(d))
        return d.value

    def mesh_array(self, name, level=0):
        deg = self.level_degree(level)
        npts = self.info["num_local_elements"] * (deg + 1) ** self.info["dim"]
        if name == "z" and self.info["dim"] == 2:
            return np.zeros(npts)  # a 2-D mesh has no z file (domain.tpp:121-138)
        dtype = {"glo_num": np.int64, "node_degree": np.int32}.get(name, np.float64)
        out = np.zeros(npts, dtype)
        _H().call("fddh_problem_mesh_array", self.h, level, name.encode(), _dp(out), out.nbytes)
        return out

    def csr(self, which):
        nr, nc, nz = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _H().call("fddh_problem_csr", self.h, which, ctypes.byref(nr), ctypes.byref(nc), ctypes.byref(nz), None, None, None)
        ptr = np.zeros(nr.value + 1, np.int32)
        col = np.zeros(nz.value, np.int32)
        val = np.zeros(nz.value)
        _H().call("fddh_problem_csr", self.h, which, None, None, None, _dp(ptr), _dp(col), _dp(val))
        return (nr.value, nc.value), ptr, col, val

    def assembled_weight(self):
        out = np.zeros(self.info["num_local_nodes"])
        _H().call("fddh_problem_assembled_weight", self.h, _dp(out), len(out))
        return out

    def set_D_hat(self, level, D):
        D = np.ascontiguousarray(D, dtype=np.float64)
        n = self.level_degree(level) + 1
        _H().call("fddh_problem_set_D_hat", self.h, level, _dp(D), n)

    def get_D_hat(self, level):
        n = self.level_degree(level) + 1
        D = np.zeros(n * n)
        _H().call("fddh_problem_get_D_hat", self.h, level, _dp(D), n)
        return D

    def set_options(self, max_iterations=-1, tolerance=float("nan"), num_vectors=-1, use_preconditioner=-1, preconditioner_type=-1, sub_num_vectors=-1, sub_max_iterations=-1, sub_build_tree=-1):
        _H().call("fddh_problem_set_options", self.h, max_iterations, tolerance, num_vectors, int(use_preconditioner), preconditioner_type, sub_num_vectors, sub_max_iterations, int(sub_build_tree))

    def set_flag(self, name, value):
        _H().call("fddh_problem_set_flag", self.h, name.encode(), int(value))

    def affine_info(self):
        """after set_flag("affine_geometry", 1): which operators run without streaming the factor arrays (an option of
        this build) and how far the mesh's own factors are from the form c_f(e) (w_i w_j) w_k"""
        dom, aff, lists, dev = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_double(-1.0)
        _H().call("fddh_problem_affine_info", self.h, ctypes.byref(dom), ctypes.byref(aff), ctypes.byref(lists), ctypes.byref(dev))
        return {"fine_domain": bool(dom.value), "sub_lists_affine": aff.value, "sub_lists": lists.value, "max_deviation": dev.value}

    def dssum(self, u, mask=True, weight=False):
        out = np.zeros(self.n)
        _H().call("fddh_problem_dssum", self.h, _dp(out), _dp(np.ascontiguousarray(u)), int(mask), int(weight))
        return out

    def stiffness(self, u, dssum=False):
        out = np.zeros(self.n)
        _H().call("fddh_problem_stiffness", self.h, _dp(out), _dp(np.ascontiguousarray(u)), int(dssum))
        return out

    def residual_norm(self, r):
        v = ctypes.c_double()
        _H().call("fddh_problem_residual_norm", self.h, _dp(np.ascontiguousarray(r)), ctypes.byref(v))
        return v.value

    def make_rhs(self, function_id=0, seed=0):
        u_star, f = np.zeros(self.n), np.zeros(self.n)
        _H().call("fddh_problem_make_rhs", self.h, function_id, seed, _dp(u_star), _dp(f))
        return u_star, f

    def make_rhs_from(self, u_star):
        u_star = np.ascontiguousarray(u_star, dtype=np.float64).copy()
        f = np.zeros(self.n)
        _H().call("fddh_problem_make_rhs_from", self.h, _dp(u_star), _dp(f))
        return u_star, f

    def solve(self, f, method="fcg"):
        u = np.zeros(self.n)
        cap = 4096
        hist = np.zeros(cap)
        nh, its = ctypes.c_int(), ctypes.c_int()
        _H().call("fddh_problem_solve", self.h, 0 if method == "fcg" else 1, _dp(np.ascontiguousarray(f)), _dp(u), _dp(hist), cap, ctypes.byref(nh), ctypes.byref(its))
        return u, its.value, hist[: min(nh.value, cap)].copy()

    def solve_timed(self, f, method="fcg", want_solution=False):
        """(iterations, history, seconds): the solve with the clock around the device work only"""
        u = np.zeros(self.n) if want_solution else None
        cap = 4096
        hist = np.zeros(cap)
        nh, its, sec = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        _H().call("fddh_problem_solve_timed", self.h, 0 if method == "fcg" else 1, _dp(np.ascontiguousarray(f)), _dp(u), _dp(hist), cap, ctypes.byref(nh), ctypes.byref(its), ctypes.byref(sec))
        return its.value, hist[: min(nh.value, cap)].copy(), sec.value

    def precond_apply(self, r, method="gmres"):
        z = np.zeros(self.n)
        hist = np.zeros(64)
        nh = ctypes.c_int()
        _H().call("fddh_problem_precond_apply", self.h, 0 if method == "fcg" else 1, _dp(np.ascontiguousarray(r)), _dp(z), _dp(hist), 64, ctypes.byref(nh))
        return z, hist[: nh.value].copy()

    def comm_time(self, iterations=20):
        """the solve path's collectives alone (collective call): dict name -> (avg us, bytes)"""
        us = (ctypes.c_double * 6)()
        nbytes = (ctypes.c_double * 6)()
        _H().call("fddh_problem_comm_time", self.h, int(iterations), us, nbytes)
        # the last two are what the solve issues by default (point-to-point groups: xGMI links every pair of GPUs directly);
        # the dense all-reduce / all-gather forms before them are the flags' other setting, timed for comparison
        names = ["allreduce_3_scalars", "interface_pair_allreduce", "coarse_allgather", "ring_exchange", "interface_pair_neighbour_exchange", "ring_and_coarse_exchange"]
        return {n: {"avg_us": us[k], "bytes": nbytes[k]} for k, n in enumerate(names)}

    def sub_op(self, op, u):
        code = {"tree": 0, "stiffness": 1, "dssum": 2}[op]
        out = np.zeros(self.info["sub_num_values"])
        _H().call("fddh_problem_sub_op", self.h, code, _dp(np.ascontiguousarray(u)), _dp(out))
        return out

    def sub_dof_operator(self, x):
        """y = (Qt A_L Q | A_sup) x on the unique dofs of the inner iteration"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(len(x))
        _H().call("fddh_problem_sub_dof_op", self.h, 0, _dp(x), _dp(y), len(x))
        return y

    def sub_jacobi_diagonal(self):
        n = self.sub_info()["unique_dofs"]
        d = np.zeros(n)
        _H().call("fddh_problem_sub_jacobi_diagonal", self.h, _dp(d), n)
        return d

    def sub_dof_rhs(self, r):
        """the dof-space right-hand side of the inner solve for the outer point vector r (collective on a composite)"""
        n = self.sub_info()["unique_dofs"]
        y = np.zeros(n)
        _H().call("fddh_problem_sub_dof_op", self.h, 1, _dp(np.ascontiguousarray(r, dtype=np.float64)), _dp(y), n)
        return y

    # --- low-order AMG preconditioner of the inner solve (hierarchy handed in) ---
    def sub_point_dofs(self):
        n = self.sub_info()["num_points"]
        dof = np.zeros(n, dtype=np.int32)
        _H().call("fddh_problem_sub_point_dofs", self.h, dof.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n)
        return dof

    def amg_attach(self, levels):
        """levels: finest first, dicts with A (scipy CSR), D (diagonal scaling),
        coefs (Chebyshev coefficients) and P (scipy CSR, None on the coarsest)."""
        ip = ctypes.POINTER(ctypes.c_int)
        for lv in levels:
            A = lv["A"].tocsr()
            A.sort_indices()
            a = (np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32), np.ascontiguousarray(A.data, dtype=np.float64))
            D = np.ascontiguousarray(lv["D"], dtype=np.float64)
            coefs = np.ascontiguousarray(lv["coefs"], dtype=np.float64)
            if lv.get("P") is not None:
                P = lv["P"].tocsr()
                P.sort_indices()
                pp = (np.ascontiguousarray(P.indptr, dtype=np.int32), np.ascontiguousarray(P.indices, dtype=np.int32), np.ascontiguousarray(P.data, dtype=np.float64))
                pargs = (P.shape[1], pp[0].ctypes.data_as(ip), pp[1].ctypes.data_as(ip), _dp(pp[2]))
            else:
                pargs = (0, None, None, None)
            _H().call("fddh_problem_amg_add_level", self.h, A.shape[0], a[0].ctypes.data_as(ip), a[1].ctypes.data_as(ip), _dp(a[2]), _dp(D), _dp(coefs), len(coefs), *pargs)
        _H().call("fddh_problem_amg_finalize", self.h)

    def amg_build(self, coarsest_size=0, strength=0.0, smooth_prolongator=True, verbose=False):
        """Low-order FEM matrix + smoothed-aggregation hierarchy built by the host layer; returns the level count."""
        nl = ctypes.c_int()
        _H().call("fddh_problem_amg_build", self.h, int(coarsest_size), float(strength), int(smooth_prolongator), int(verbose), ctypes.byref(nl))
        return nl.value

    def amg_level_transfer(self, level):
        """True where the interpolator of `level` is applied matrix-free (fdd_lattice_prolong / _restrict)."""
        flag = ctypes.c_int(0)
        _H().call("fddh_problem_amg_level_transfer", self.h, int(level), ctypes.byref(flag))
        return bool(flag.value)

    def amg_levels(self, cheby_order=2):
        """The attached hierarchy as scipy matrices (finest first), as amg_attach takes it."""
        import scipy.sparse as sp

        ip = ctypes.POINTER(ctypes.c_int)
        out = []
        level = 0
        while True:
            n, nnz, nc, nnzp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            try:
                _H().call("fddh_problem_amg_level_info", self.h, level, ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(nc), ctypes.byref(nnzp))
            except lib.FddError:
                break
            ap, ac, av = np.zeros(n.value + 1, np.int32), np.zeros(nnz.value, np.int32), np.zeros(nnz.value)
            D, coefs = np.zeros(n.value), np.zeros(8)
            has_p = nnzp.value > 0
            pp, pc, pv = np.zeros(n.value + 1, np.int32), np.zeros(max(nnzp.value, 1), np.int32), np.zeros(max(nnzp.value, 1))
            _H().call("fddh_problem_amg_level_arrays", self.h, level, ap.ctypes.data_as(ip), ac.ctypes.data_as(ip), _dp(av), _dp(D), _dp(coefs),
                      pp.ctypes.data_as(ip) if has_p else None, pc.ctypes.data_as(ip) if has_p else None, _dp(pv) if has_p else None)
            lv = {"A": sp.csr_matrix((av, ac, ap), shape=(n.value, n.value)), "D": D, "coefs": coefs[:cheby_order].copy(), "P": None}
            if has_p:
                lv["P"] = sp.csr_matrix((pv[: nnzp.value], pc[: nnzp.value], pp), shape=(n.value, nc.value))
            out.append(lv)
            level += 1
        return out

    def amg_apply(self, r):
        z = np.zeros(self.n)
        _H().call("fddh_problem_amg_apply", self.h, _dp(np.ascontiguousarray(r)), _dp(z))
        return z

    def sub_residual_norm(self, r):
        v = ctypes.c_double()
        _H().call("fddh_problem_sub_residual_norm", self.h, _dp(np.ascontiguousarray(r)), ctypes.byref(v))
        return v.value

    def pcg_begin(self, f):
        _H().call("fddh_problem_pcg_begin", self.h, _dp(np.ascontiguousarray(f)))

    def pcg_steps(self, steps):
        v = ctypes.c_double()
        _H().call("fddh_problem_pcg_steps", self.h, steps, ctypes.byref(v))
        return v.value

    def pcg_solution(self):
        u = np.zeros(self.n)
        _H().call("fddh_problem_pcg_solution", self.h, _dp(u))
        return u


def spmv_stencil_time(m, iterations=10):
    """(avg_us, algorithmic_bytes, nnz) of the 27-point-stencil SpMV on an m^3 node grid"""
    us, nbytes, nnz = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
    _H().call("fddh_spmv_stencil_time", int(m), int(iterations), ctypes.byref(us), ctypes.byref(nbytes), ctypes.byref(nnz))
    return us.value, nbytes.value, nnz.value


def sync():
    _H().call("fddh_sync")


def barrier():
    _H().call("fddh_barrier")
