"""Test support: ctypes binding of the CPU oracle, GLL golden tables, and an
independent numpy generator of the synthetic box mesh.

The oracle is the CHECKER only; nothing here is imported by the product.
"""
from __future__ import annotations

import ctypes
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libfdd_oracle.so")
SPECLIB_REF_SO = os.path.join(ORACLE_DIR, "_ref", "libspeclib_ref.so")
GOLDEN_DIR = os.path.join(HERE, "golden")

c_int_p = ctypes.POINTER(ctypes.c_int)
vp = ctypes.c_void_p


def _p(a):
    if a is None:
        return vp(0)
    assert a.flags["C_CONTIGUOUS"]
    return vp(a.ctypes.data)


class OrcCsr(ctypes.Structure):
    _fields_ = [
        ("num_rows", ctypes.c_int),
        ("num_cols", ctypes.c_int),
        ("num_nnz", ctypes.c_int),
        ("ptr", c_int_p),
        ("col", c_int_p),
        ("val", ctypes.POINTER(ctypes.c_double)),
    ]

    def to_numpy(self):
        ptr = np.ctypeslib.as_array(self.ptr, shape=(self.num_rows + 1,)).copy()
        if self.num_nnz:
            col = np.ctypeslib.as_array(self.col, shape=(self.num_nnz,)).copy()
            val = np.ctypeslib.as_array(self.val, shape=(self.num_nnz,)).copy()
        else:
            col = np.zeros(0, np.int32)
            val = np.zeros(0)
        return ptr, col, val


class OrcMesh(ctypes.Structure):
    _fields_ = [
        ("dim", ctypes.c_int),
        ("poly_degree", ctypes.c_int),
        ("num_local_elements", ctypes.c_int),
        ("x", vp),
        ("y", vp),
        ("z", vp),
        ("glo_num", vp),
        ("node_degree", vp),
        ("p_mask", vp),
        ("g", vp * 6),
    ]


PRECOND_FN = ctypes.CFUNCTYPE(None, vp, ctypes.POINTER(vp), ctypes.POINTER(vp))


class OrcSolverOpts(ctypes.Structure):
    _fields_ = [
        ("max_iterations", ctypes.c_int),
        ("num_vectors", ctypes.c_int),
        ("tolerance", ctypes.c_double),
        ("use_relative", ctypes.c_int),
        ("precond", PRECOND_FN),
        ("precond_ctx", vp),
    ]


class OrcSubdomainOpts(ctypes.Structure):
    _fields_ = [
        ("num_vectors", ctypes.c_int),
        ("max_iterations", ctypes.c_int),
        ("tolerance", ctypes.c_double),
        ("use_preconditioner", ctypes.c_int),
    ]


_oracle = None


def oracle():
    """Load (building if needed) the CPU oracle."""
    global _oracle
    if _oracle is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        stale = (not os.path.exists(ORACLE_SO)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs)
        if stale:
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        L = ctypes.CDLL(ORACLE_SO)
        L.orc_block_sum.restype = ctypes.c_double
        L.orc_world_create.restype = vp
        L.orc_world_Q.restype = ctypes.POINTER(OrcCsr)
        L.orc_world_Qt.restype = ctypes.POINTER(OrcCsr)
        L.orc_world_assembled_weight.restype = ctypes.POINTER(ctypes.c_double)
        L.orc_world_residual_norm.restype = ctypes.c_double
        L.orc_world_assembled_inner_product.restype = ctypes.c_double
        L.orc_subdomain_create.restype = vp
        L.orc_subdomain_residual_norm.restype = ctypes.c_double
        L.orc_amg_create.restype = vp
        L.orc_f32_multi_axpy_norm2_scaled.restype = ctypes.c_double
        L.orc_fdd_create.restype = vp
        L.orc_fdd_subdomain.restype = vp
        L.orc_fdd_composite_levels.restype = ctypes.POINTER(ctypes.c_int)
        L.orc_fdd_matrix.restype = ctypes.POINTER(OrcCsr)
        L.orc_fdd_norm_weight.restype = ctypes.POINTER(ctypes.c_double)
        L.orc_fdd_low_order_matrix.restype = ctypes.POINTER(OrcCsr)
        L.orc_fdd_point_dofs.restype = ctypes.POINTER(ctypes.c_int)
        L.orc_fdd_inner_weight.restype = ctypes.POINTER(ctypes.c_double)
        L.orc_amg32_create.restype = vp
        _oracle = L
    return _oracle


# --------------------------------------------------------------------------
# GLL tables
# --------------------------------------------------------------------------
_tables = None


def gll_tables():
    global _tables
    if _tables is None:
        with open(os.path.join(GOLDEN_DIR, "gll_tables.json")) as fh:
            _tables = json.load(fh)
    return _tables


def gll(N):
    t = gll_tables()["levels"][str(N)]
    return np.array(t["z"]), np.array(t["w"]), np.array(t["D_hat"])


def J_cf(N_c, N_f):
    return np.array(gll_tables()["J_cf"][f"{N_c},{N_f}"])


def level_degrees(N, reduction):
    """subdomain.tpp:98-110"""
    deg = [N]
    while deg[-1] > 1:
        deg.append(max(deg[-1] - reduction, 1))
    return deg


# --------------------------------------------------------------------------
# synthetic box mesh (independent numpy statement of SURVEY.md section 8(d))
# --------------------------------------------------------------------------
class BoxMesh:
    """Unit cube, E = (Ex,Ey,Ez) global elements split into (Px,Py,Pz) rank
    blocks; this object is rank `rank`'s part at degree N.  Arrays are
    element-major with x fastest inside an element, exactly what
    Domain::initialize reads (domain.tpp:45-224)."""

    def __init__(self, E, N, P=(1, 1, 1), rank=0):
        E = tuple(E)
        P = tuple(P)
        self.E, self.N, self.P, self.rank = E, N, P, rank
        n = N + 1
        z, w, _ = gll(N)
        Ex, Ey, Ez = E
        Px, Py, Pz = P
        assert Ex % Px == 0 and Ey % Py == 0 and Ez % Pz == 0
        lx, ly, lz = Ex // Px, Ey // Py, Ez // Pz
        rx, ry, rz = rank % Px, (rank // Px) % Py, rank // (Px * Py)
        self.local_E = (lx, ly, lz)
        self.origin = (rx * lx, ry * ly, rz * lz)
        ne = lx * ly * lz
        self.num_local_elements = ne
        self.num_elem_points = n**3
        self.num_local_points = ne * n**3
        hx, hy, hz = 1.0 / Ex, 1.0 / Ey, 1.0 / Ez

        # global node grid
        Gx, Gy, Gz = Ex * N + 1, Ey * N + 1, Ez * N + 1
        self.global_nodes = Gx * Gy * Gz

        ex = np.arange(lx) + self.origin[0]
        ey = np.arange(ly) + self.origin[1]
        ez = np.arange(lz) + self.origin[2]
        # element-major ordering: local element id = ix + lx*(iy + ly*iz)
        EZ, EY, EX = np.meshgrid(ez, ey, ex, indexing="ij")
        EX, EY, EZ = EX.reshape(-1), EY.reshape(-1), EZ.reshape(-1)

        k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        i, j, k = i.reshape(-1), j.reshape(-1), k.reshape(-1)  # x fastest

        gi = EX[:, None] * N + i[None, :]
        gj = EY[:, None] * N + j[None, :]
        gk = EZ[:, None] * N + k[None, :]

        # vertices first (the same id at every degree, as Nek5000 numbers them), then the other nodes
        V = (Ex + 1) * (Ey + 1) * (Ez + 1)
        is_vertex = (gi % N == 0) & (gj % N == 0) & (gk % N == 0)
        self.glo_num = np.where(is_vertex, 1 + gi // N + (Ex + 1) * (gj // N + (Ey + 1) * (gk // N)), V + 1 + gi + Gx * (gj + Gy * gk)).astype(np.int64).reshape(-1)

        def mult(g, G):
            # number of elements sharing a grid line index g along one axis
            m = np.ones_like(g)
            m[(g % N == 0) & (g > 0) & (g < G - 1)] = 2
            return m

        self.node_degree = (mult(gi, Gx) * mult(gj, Gy) * mult(gk, Gz)).astype(np.int32).reshape(-1)
        on_bdry = (gi == 0) | (gi == Gx - 1) | (gj == 0) | (gj == Gy - 1) | (gk == 0) | (gk == Gz - 1)
        self.p_mask = np.where(on_bdry, 0.0, 1.0).reshape(-1)

        xi = 0.5 * (z + 1.0)
        self.x = ((EX[:, None] + xi[i][None, :]) * hx).reshape(-1)
        self.y = ((EY[:, None] + xi[j][None, :]) * hy).reshape(-1)
        self.z = ((EZ[:, None] + xi[k][None, :]) * hz).reshape(-1)

        # geometric factors with quadrature weights folded in: the stiffness
        # integrand of an affine hex, G_rr = w_i w_j w_k * (hy*hz)/(2*hx) etc.;
        # for the cube hx=hy=hz=h this is w_i w_j w_k * h/2.
        www = (w[i] * w[j] * w[k])[None, :] * np.ones((ne, 1))
        self.g = [
            (www * (hy * hz) / (2.0 * hx)).reshape(-1),
            (www * (hx * hz) / (2.0 * hy)).reshape(-1),
            (www * (hx * hy) / (2.0 * hz)).reshape(-1),
            np.zeros(ne * n**3),
            np.zeros(ne * n**3),
            np.zeros(ne * n**3),
        ]

    def orc_mesh(self):
        m = OrcMesh()
        m.dim = getattr(self, "dim", 3)
        m.poly_degree = self.N
        m.num_local_elements = self.num_local_elements
        m.x, m.y, m.z = _p(self.x), _p(self.y), _p(self.z)
        m.glo_num = _p(self.glo_num)
        m.node_degree = _p(self.node_degree)
        m.p_mask = _p(self.p_mask)
        for g in range(6):
            m.g[g] = _p(self.g[g]).value
        return m


class ArrayMesh(BoxMesh):
    """A mesh given by its arrays (e.g. the ones the C++ host layer generated,
    so that oracle and product see bit-identical geometric factors)."""

    def __init__(self, N, num_local_elements, arrays, dim=3):
        self.N = N
        self.dim = dim
        self.num_local_elements = num_local_elements
        self.num_elem_points = (N + 1) ** dim
        self.num_local_points = num_local_elements * self.num_elem_points
        self.x, self.y, self.z = arrays["x"], arrays["y"], arrays["z"]
        self.glo_num = arrays["glo_num"]
        self.node_degree = arrays["node_degree"]
        self.p_mask = arrays["p_mask"]
        self.g = [arrays[f"g_{k + 1}"] for k in range(6)]

    @classmethod
    def from_problem(cls, problem, level=0):
        names = ["x", "y", "z", "glo_num", "node_degree", "p_mask"] + [f"g_{k + 1}" for k in range(6)]
        arrays = {n: problem.mesh_array(n, level) for n in names}
        return cls(problem.level_degree(level), problem.info["num_local_elements"], arrays, problem.info.get("dim", 3))


class RodMesh(BoxMesh):
    """BoxMesh with the Dirichlet condition on the two x ends only (natural condition on the lateral faces): a rod.
    Cut into rank strips along x, the inner strips float (no Dirichlet node of their own), and the coupling along x
    is global: the configuration that separates a full-domain-decomposition preconditioner from block-Jacobi."""

    def __init__(self, E, N, P=(1, 1, 1), rank=0):
        super().__init__(E, N, P, rank)
        on_ends = (np.abs(self.x) < 1e-12) | (np.abs(self.x - 1.0) < 1e-12)
        self.p_mask = np.where(on_ends, 0.0, 1.0)


def rank_grid(num_ranks):
    """Rank blocks for a cube: 1->(1,1,1), 2->(2,1,1), 4->(2,2,1), 8->(2,2,2)."""
    return {1: (1, 1, 1), 2: (2, 1, 1), 3: (3, 1, 1), 4: (2, 2, 1), 6: (6, 1, 1), 8: (2, 2, 2)}[num_ranks]  # as host_api.rank_grid


# --------------------------------------------------------------------------
# oracle world wrapper
# --------------------------------------------------------------------------
class OracleWorld:
    def __init__(self, meshes, N):
        self.L = oracle()
        self.meshes = meshes
        self.R = len(meshes)
        self._cm = (OrcMesh * self.R)(*[m.orc_mesh() for m in meshes])
        _, _, D = gll(N)
        self.D_hat = np.ascontiguousarray(D)
        self.w = vp(self.L.orc_world_create(self.R, self._cm, _p(self.D_hat)))
        self.npts = [self.L.orc_world_num_local_points(self.w, r) for r in range(self.R)]

    def close(self):
        if self.w:
            self.L.orc_world_destroy(self.w)
            self.w = None

    def _pp(self, arrays):
        arr = (vp * self.R)()
        for r, a in enumerate(arrays):
            arr[r] = a.ctypes.data
        return arr

    def zeros(self):
        return [np.zeros(n) for n in self.npts]

    def Q(self, r):
        return self.L.orc_world_Q(self.w, r).contents.to_numpy()

    def Qt(self, r):
        return self.L.orc_world_Qt(self.w, r).contents.to_numpy()

    def num_nodes(self, r):
        return self.L.orc_world_num_local_nodes(self.w, r)

    def num_bdary(self, r):
        return self.L.orc_world_num_bdary_nodes(self.w, r)

    def assembled_weight(self, r):
        p = self.L.orc_world_assembled_weight(self.w, r)
        return np.ctypeslib.as_array(p, shape=(self.num_nodes(r),)).copy()

    def dssum(self, u, mask=True, weight=False):
        out = self.zeros()
        self.L.orc_world_dssum(self.w, self._pp(out), self._pp(u), int(mask), int(weight))
        return out

    def stiffness(self, u, dssum=False):
        out = self.zeros()
        self.L.orc_world_stiffness(self.w, self._pp(out), self._pp(u), int(dssum))
        return out

    def residual_norm(self, r):
        return self.L.orc_world_residual_norm(self.w, self._pp(r))

    def inner(self, u, v):
        return self.L.orc_world_assembled_inner_product(self.w, self._pp(u), self._pp(v))

    def solve(self, f, method="fcg", max_iterations=500, num_vectors=20, tolerance=1e-7, precond=None):
        u = self.zeros()
        opts = OrcSolverOpts()
        opts.max_iterations = max_iterations
        opts.num_vectors = num_vectors
        opts.tolerance = tolerance
        opts.use_relative = 1
        keep = None
        if precond is not None:
            npts = self.npts
            R = self.R

            def cb(ctx, zpp, rpp):
                z = [np.ctypeslib.as_array(ctypes.cast(zpp[r], ctypes.POINTER(ctypes.c_double)), shape=(npts[r],)) for r in range(R)]
                rr = [np.ctypeslib.as_array(ctypes.cast(rpp[r], ctypes.POINTER(ctypes.c_double)), shape=(npts[r],)) for r in range(R)]
                precond(z, rr)

            keep = PRECOND_FN(cb)
            opts.precond = keep
        else:
            opts.precond = ctypes.cast(None, PRECOND_FN)
        hist = np.zeros(max_iterations + 2)
        nh = ctypes.c_int(0)
        fn = self.L.orc_world_fcg if method == "fcg" else self.L.orc_world_gmres
        its = fn(self.w, self._pp(u), self._pp(f), ctypes.byref(opts), _p(hist), len(hist), ctypes.byref(nh))
        return u, its, hist[: nh.value].copy()


class OracleSubdomain:
    """orc_subdomain for one rank's own elements (conforming composite)."""

    def __init__(self, E, N, reduction, P=(1, 1, 1), rank=0, meshes=None):
        self.L = oracle()
        self.deg = level_degrees(N, reduction)
        self.meshes = meshes if meshes is not None else [BoxMesh(E, d, P, rank) for d in self.deg]
        nl = len(self.deg)
        self._cm = (OrcMesh * nl)(*[m.orc_mesh() for m in self.meshes])
        self._D = [np.ascontiguousarray(gll(d)[2]) for d in self.deg]
        self._J = [np.ascontiguousarray(J_cf(self.deg[l + 1], self.deg[l])) for l in range(nl - 1)]
        Dp = (vp * nl)(*[a.ctypes.data for a in self._D])
        Jp = (vp * max(nl, 1))(*([a.ctypes.data for a in self._J] + [0]))
        degs = (ctypes.c_int * nl)(*self.deg)
        self.s = vp(self.L.orc_subdomain_create(nl, degs, Dp, Jp, self._cm))
        self.num_values = self.L.orc_subdomain_num_values(self.s)
        self.num_points = self.meshes[0].num_local_points

    def close(self):
        if self.s:
            self.L.orc_subdomain_destroy(self.s)
            self.s = None

    def tree(self, u):
        out = np.zeros(self.num_values)
        self.L.orc_subdomain_tree_operator(self.s, _p(out), _p(u))
        return out

    def stiffness(self, u):
        out = np.zeros(self.num_values)
        self.L.orc_subdomain_stiffness(self.s, _p(out), _p(u))
        return out

    def dssum(self, u):
        out = np.zeros(self.num_values)
        self.L.orc_subdomain_dssum(self.s, _p(out), _p(u))
        return out

    def residual_norm(self, r):
        return self.L.orc_subdomain_residual_norm(self.s, _p(r))

    def point_dofs(self):
        dof = np.zeros(self.num_points, dtype=np.int32)
        self.L.orc_subdomain_point_dofs(self.s, dof.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        return dof

    def num_dofs(self):
        return self.L.orc_subdomain_num_dofs(self.s)

    def attach_amg(self, levels, cheby_order=2, num_vcycles=1):
        """levels as Problem.amg_attach takes them (finest first)."""
        ip = ctypes.POINTER(ctypes.c_int)
        self.amg = vp(self.L.orc_amg_create(len(levels), cheby_order, num_vcycles))
        self._amg_keep = []
        for l, lv in enumerate(levels):
            A = lv["A"].tocsr()
            A.sort_indices()
            arrs = [np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32), np.ascontiguousarray(A.data, dtype=np.float64),
                    np.ascontiguousarray(lv["D"], dtype=np.float64), np.ascontiguousarray(lv["coefs"], dtype=np.float64)]
            if lv.get("P") is not None:
                P = lv["P"].tocsr()
                P.sort_indices()
                parr = [np.ascontiguousarray(P.indptr, dtype=np.int32), np.ascontiguousarray(P.indices, dtype=np.int32), np.ascontiguousarray(P.data, dtype=np.float64)]
                pargs = (P.shape[1], parr[0].ctypes.data_as(ip), parr[1].ctypes.data_as(ip), _p(parr[2]))
                arrs += parr
            else:
                pargs = (0, None, None, None)
            self._amg_keep.append(arrs)
            self.L.orc_amg_set_level(self.amg, l, A.shape[0], arrs[0].ctypes.data_as(ip), arrs[1].ctypes.data_as(ip), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]), *pargs)
        self.L.orc_subdomain_attach_amg(self.s, self.amg)

    def jacobi_diagonal(self):
        """point-Jacobi option: the diagonal of the inner iteration's operator over the unique dofs (oracle numbering)"""
        d = np.zeros(self.num_dofs())
        self.L.orc_subdomain_jacobi_diagonal(self.s, _p(d))
        return d

    def element_diagonal_check(self):
        self.L.orc_subdomain_element_diagonal_check.restype = ctypes.c_double
        return self.L.orc_subdomain_element_diagonal_check(self.s)

    def low_order_preconditioner(self, r):
        z = np.zeros(self.num_values)
        self.L.orc_subdomain_low_order_preconditioner(self.s, _p(z), _p(np.ascontiguousarray(r)))
        return z

    def solve(self, f, method="gmres", num_vectors=4, max_iterations=4, tolerance=1e-12, use_preconditioner=False):
        u = np.zeros(self.num_points)
        opts = OrcSubdomainOpts(num_vectors, max_iterations, tolerance, int(use_preconditioner))
        hist = np.zeros(max_iterations + 2)
        nh = ctypes.c_int(0)
        fn = self.L.orc_subdomain_gmres if method == "gmres" else self.L.orc_subdomain_fcg
        its = fn(self.s, _p(u), _p(np.ascontiguousarray(f)), ctypes.byref(opts), _p(hist), len(hist), ctypes.byref(nh))
        return u, its, hist[: nh.value].copy()


class OracleFdd:
    """orc_fdd: the full-domain-decomposition composite of every rank of an R-rank run
    (oracle/fdd_oracle_composite.c) and the preconditioner application on it."""

    INFO = ["sub_elems", "sub_ext_elems", "points", "sub_dofs", "sub_ext_dofs", "interface_dofs", "sup_dofs", "sup_ext_dofs", "unique_dofs", "coarse_dofs", "num_values", "own_points"]

    def __init__(self, E, N, reduction, P, subdomain_overlap=1, superdomain_overlap=1, mesh_class=None, meshes=None):
        self.L = oracle()
        self.deg = level_degrees(N, reduction)
        self.R = int(np.prod(P))
        nl = len(self.deg)
        mc = mesh_class or BoxMesh
        # meshes[rank][level]
        self.meshes = meshes if meshes is not None else [[mc(E, d, P, r) for d in self.deg] for r in range(self.R)]
        flat = [m.orc_mesh() for row in self.meshes for m in row]
        self._cm = (OrcMesh * len(flat))(*flat)
        self._D = [np.ascontiguousarray(gll(d)[2]) for d in self.deg]
        self._J = {}
        Jp = (vp * (nl * nl))()
        for lf in range(nl):
            for lc in range(lf + 1, nl):
                self._J[(lf, lc)] = np.ascontiguousarray(J_cf(self.deg[lc], self.deg[lf]))
                Jp[lf * nl + lc] = self._J[(lf, lc)].ctypes.data
        Dp = (vp * nl)(*[a.ctypes.data for a in self._D])
        degs = (ctypes.c_int * nl)(*self.deg)
        self.f = vp(self.L.orc_fdd_create(self.R, nl, degs, Dp, Jp, self._cm, subdomain_overlap, superdomain_overlap))
        self.info = [self._info(r) for r in range(self.R)]
        self.npts = [self.meshes[r][0].num_local_points for r in range(self.R)]

    def _info(self, r):
        buf = (ctypes.c_int * 12)()
        self.L.orc_fdd_info(self.f, r, buf)
        return dict(zip(self.INFO, list(buf)))

    def close(self):
        if self.f:
            self.L.orc_fdd_destroy(self.f)
            self.f = None

    def _pp(self, arrays):
        arr = (vp * self.R)()
        for r, a in enumerate(arrays):
            arr[r] = a.ctypes.data
        return arr

    def region(self, r):
        n = self.info[r]["sub_ext_elems"]
        ids = np.zeros(n, np.int32)
        lv = np.zeros(n, np.int32)
        ip = ctypes.POINTER(ctypes.c_int)
        self.L.orc_fdd_region(self.f, r, ids.ctypes.data_as(ip), lv.ctypes.data_as(ip))
        return ids, lv

    def composite_levels(self, r):
        p = self.L.orc_fdd_composite_levels(self.f, r)
        out = []
        k = 0
        while p[k] >= 0:
            out.append(p[k])
            k += 1
        return out

    def sub(self, r):
        return vp(self.L.orc_fdd_subdomain(self.f, r))

    MATRICES = {"Q": 0, "Qt": 1, "Q_int": 2, "Qt_int": 3, "QQt_int": 4, "A_sup": 5, "Pt": 6, "Qt_coarse": 7}

    def matrix(self, r, name):
        """scipy CSR of one of the composite's operators"""
        import scipy.sparse as sp

        c = self.L.orc_fdd_matrix(self.f, r, self.MATRICES[name]).contents
        ptr, col, val = c.to_numpy()
        return sp.csr_matrix((val, col, ptr), shape=(c.num_rows, c.num_cols))

    def low_order_matrix(self, r):
        """the composite's low-order operator over the unique dofs (subdomain.tpp:2749-3472), scipy CSR"""
        import scipy.sparse as sp

        c = self.L.orc_fdd_low_order_matrix(self.f, r).contents
        ptr, col, val = c.to_numpy()
        return sp.csr_matrix((val, col, ptr), shape=(c.num_rows, c.num_cols))

    def point_dofs(self, r):
        return np.ctypeslib.as_array(self.L.orc_fdd_point_dofs(self.f, r), shape=(self.info[r]["points"],)).copy()

    def attach_amg(self, r, levels, cheby_order=2, num_vcycles=1):
        """hand rank r's composite the hierarchy of its low-order preconditioner (finest level first, in the oracle's numbering)"""
        ip = ctypes.POINTER(ctypes.c_int)
        amg = vp(self.L.orc_amg_create(len(levels), cheby_order, num_vcycles))
        keep = []
        for l, lv in enumerate(levels):
            A = lv["A"].tocsr()
            A.sort_indices()
            arrs = [np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32), np.ascontiguousarray(A.data, dtype=np.float64),
                    np.ascontiguousarray(lv["D"], dtype=np.float64), np.ascontiguousarray(lv["coefs"], dtype=np.float64)]
            if lv.get("P") is not None:
                P = lv["P"].tocsr()
                P.sort_indices()
                parr = [np.ascontiguousarray(P.indptr, dtype=np.int32), np.ascontiguousarray(P.indices, dtype=np.int32), np.ascontiguousarray(P.data, dtype=np.float64)]
                pargs = (P.shape[1], parr[0].ctypes.data_as(ip), parr[1].ctypes.data_as(ip), _p(parr[2]))
                arrs += parr
            else:
                pargs = (0, None, None, None)
            keep.append(arrs)
            self.L.orc_amg_set_level(amg, l, A.shape[0], arrs[0].ctypes.data_as(ip), arrs[1].ctypes.data_as(ip), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]), *pargs)
        self.L.orc_subdomain_attach_amg(self.sub(r), amg)
        self._amg_keep = getattr(self, "_amg_keep", []) + [(amg, keep)]

    def norm_weight(self, r):
        n = self.info[r]["sub_ext_dofs"] + self.info[r]["sup_ext_dofs"]
        return np.ctypeslib.as_array(self.L.orc_fdd_norm_weight(self.f, r), shape=(n,)).copy()

    def region_points(self, r, name):
        """a mesh array (x, y, z, p_mask, ...) of the region's points, pulled from the owners' meshes"""
        ids, lv = self.region(r)
        counts = np.cumsum([0] + [self.meshes[p][0].num_local_elements for p in range(self.R)])
        out = []
        for e, l in zip(ids, lv):
            p = int(np.searchsorted(counts, e, side="right") - 1)
            m = self.meshes[p][l]
            npts = m.num_elem_points
            le = e - counts[p]
            out.append(getattr(m, name)[le * npts : (le + 1) * npts])
        return np.concatenate(out)

    def tree(self, u):
        out = [np.zeros(self.info[r]["num_values"]) for r in range(self.R)]
        self.L.orc_fdd_tree_operator(self.f, self._pp(out), self._pp([np.ascontiguousarray(a) for a in u]))
        return out

    def stiffness(self, r, u):
        out = np.zeros(self.info[r]["num_values"])
        self.L.orc_subdomain_stiffness(self.sub(r), _p(out), _p(np.ascontiguousarray(u)))
        return out

    def dssum(self, r, u):
        out = np.zeros(self.info[r]["num_values"])
        self.L.orc_subdomain_dssum(self.sub(r), _p(out), _p(np.ascontiguousarray(u)))
        return out

    def residual_norm(self, r, v):
        return self.L.orc_subdomain_residual_norm(self.sub(r), _p(np.ascontiguousarray(v)))

    def jacobi_diagonal(self, r):
        d = np.zeros(self.info[r]["unique_dofs"])
        self.L.orc_subdomain_jacobi_diagonal(self.sub(r), _p(d))
        return d

    def element_diagonal_check(self, r):
        self.L.orc_subdomain_element_diagonal_check.restype = ctypes.c_double
        return self.L.orc_subdomain_element_diagonal_check(self.sub(r))

    def precondition(self, r, method="gmres", num_vectors=4, max_iterations=4, tolerance=1e-12, use_preconditioner=False):
        """r: one outer (own points) vector per rank; returns z per rank and the inner histories."""
        z = [np.zeros(n) for n in self.npts]
        opts = OrcSubdomainOpts(num_vectors, max_iterations, tolerance, int(use_preconditioner))
        cap = max_iterations + 2
        hist = np.zeros((self.R, cap))
        nh = (ctypes.c_int * self.R)()
        rr = [np.ascontiguousarray(a) for a in r]
        self.L.orc_fdd_precondition(self.f, self._pp(z), self._pp(rr), 0 if method == "fcg" else 1, ctypes.byref(opts), _p(hist), cap, nh)
        return z, [hist[k, : nh[k]].copy() for k in range(self.R)]


def composite_dof_permutation(product_point_dofs, oracle_point_dofs, num_sub_dofs, num_unique_dofs):
    """to_oracle[u]: the oracle's unique dof of the product's unique dof u.  The two sides number the subdomain's own
    dofs differently (Domain node order vs the reference's ranking of global ids) and everything else alike; the
    points that carry a dof directly pair them up."""
    both = (product_point_dofs >= 0) & (oracle_point_dofs >= 0)
    assert np.array_equal(product_point_dofs >= 0, oracle_point_dofs >= 0)
    to_oracle = np.arange(num_unique_dofs)
    a, b = product_point_dofs[both], oracle_point_dofs[both]
    own = a < num_sub_dofs
    assert np.array_equal(own, b < num_sub_dofs)
    to_oracle[a[own]] = b[own]
    assert len(np.unique(to_oracle)) == num_unique_dofs
    return to_oracle


def permute_hierarchy(levels, to_oracle):
    """level 0 of a hierarchy (A, D, coefs, P) renumbered by to_oracle; coarse levels shared"""
    import scipy.sparse as sp

    nd = len(to_oracle)
    Pm = sp.csr_matrix((np.ones(nd), (to_oracle, np.arange(nd))), shape=(nd, nd))
    fine = dict(levels[0])
    fine["A"] = (Pm @ levels[0]["A"] @ Pm.T).tocsr()
    fine["D"] = np.asarray(Pm @ levels[0]["D"])
    if levels[0].get("P") is not None:
        fine["P"] = (Pm @ levels[0]["P"]).tocsr()
    return [fine] + list(levels[1:])


def seeded_uniform(n, seed=1234):
    """Seeded stand-in for the reference's unseeded rand()/RAND_MAX RHS
    (domain.tpp:572-573)."""
    return np.random.Generator(np.random.MT19937(seed)).random(n)


# --------------------------------------------------------------------------
# A low-order AMG hierarchy for the inner preconditioner.  The reference gets
# it from HYPRE BoomerAMG on its low-order FEM matrix (subdomain.tpp:2749-3549);
# HYPRE is not in this image, so the tests build a geometric-multigrid
# hierarchy of the same shape with scipy: trilinear FEM on the GLL grid of the
# subdomain, coarsened by 2 per direction with Galerkin coarse operators,
# Chebyshev(2) data per level as hypre's ds / coefs arrays hold them.
# --------------------------------------------------------------------------
def _fem_1d(xs):
    import scipy.sparse as sp

    h = np.diff(xs)
    n = len(xs)
    K = sp.lil_matrix((n, n))
    M = sp.lil_matrix((n, n))
    for e in range(n - 1):
        K[e, e] += 1 / h[e]; K[e + 1, e + 1] += 1 / h[e]; K[e, e + 1] -= 1 / h[e]; K[e + 1, e] -= 1 / h[e]
        M[e, e] += h[e] / 2; M[e + 1, e + 1] += h[e] / 2  # lumped: the closest low-order match of the GLL quadrature mass
    return K.tocsr(), M.tocsr()


def _interp_1d(xf, keep):
    """linear interpolation from the coarse grid xf[keep] to xf"""
    import scipy.sparse as sp

    xc = xf[keep]
    P = sp.lil_matrix((len(xf), len(xc)))
    for i, x in enumerate(xf):
        j = min(np.searchsorted(xc, x, side="right") - 1, len(xc) - 2) if len(xc) > 1 else 0
        if len(xc) == 1:
            P[i, 0] = 1.0
            continue
        t = (x - xc[j]) / (xc[j + 1] - xc[j])
        if abs(t) < 1e-14:
            P[i, j] = 1.0
        elif abs(t - 1) < 1e-14:
            P[i, j + 1] = 1.0
        else:
            P[i, j] = 1 - t
            P[i, j + 1] = t
    return P.tocsr()


def chebyshev_coefs(lmax, lo=0.3, hi=1.1):
    """degree-1 Chebyshev polynomial p(x) = c0 + c1 x ~ 1/x on [lo, hi]*lmax: two Chebyshev steps"""
    a, b = lo * lmax, hi * lmax
    theta, delta = (a + b) / 2, (b - a) / 2
    den = 2 * theta * theta - delta * delta
    return np.array([4 * theta / den, -2 / den])


def low_order_hierarchy(mesh, point_dof, num_dofs, min_size=30, max_levels=8):
    """mesh: BoxMesh/ArrayMesh with x, y, z per level-0 point; point_dof: dof per point (-1: none)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    coords = [np.asarray(getattr(mesh, c), dtype=np.float64) for c in ("x", "y", "z")]
    axes, idx = [], []
    for c in coords:
        u, inv = np.unique(np.round(c, 10), return_inverse=True)
        axes.append(u)
        idx.append(inv)
    nx, ny, nz = (len(a) for a in axes)
    grid_of_point = idx[0] + nx * (idx[1] + ny * idx[2])
    has = point_dof >= 0
    grid_of_dof = np.full(num_dofs, -1, dtype=np.int64)
    grid_of_dof[point_dof[has]] = grid_of_point[has]
    assert (grid_of_dof >= 0).all()
    # all points of one dof sit on the same grid node
    assert (grid_of_dof[point_dof[has]] == grid_of_point[has]).all()

    K, M = zip(*[_fem_1d(a) for a in axes])
    A_full = (sp.kron(M[2], sp.kron(M[1], K[0])) + sp.kron(M[2], sp.kron(K[1], M[0])) + sp.kron(K[2], sp.kron(M[1], M[0]))).tocsr()
    active = np.zeros(nx * ny * nz, dtype=bool)
    active[grid_of_dof] = True
    A = A_full[grid_of_dof][:, grid_of_dof].tocsr()

    levels = []
    order = grid_of_dof  # grid node of every dof of the current level, in dof order
    cur_axes = axes
    for _ in range(max_levels):
        n = A.shape[0]
        D = 1.0 / np.sqrt(A.diagonal())
        DAD = sp.diags(D) @ A @ sp.diags(D)
        lmax = float(spla.eigsh(DAD, k=1, which="LA", return_eigenvectors=False, tol=1e-4, v0=np.ones(n))[0]) if n > 2 else float(np.linalg.eigvalsh(DAD.toarray()).max())
        lv = {"A": A, "D": D, "coefs": chebyshev_coefs(lmax), "P": None}
        levels.append(lv)
        if n <= min_size:
            break
        keeps = [np.unique(np.r_[np.arange(0, len(a), 2), len(a) - 1]) for a in cur_axes]
        if all(len(k) == len(a) for k, a in zip(keeps, cur_axes)):
            break
        P1 = [_interp_1d(a, k) for a, k in zip(cur_axes, keeps)]
        P_full = sp.kron(P1[2], sp.kron(P1[1], P1[0])).tocsr()
        fnx, fny = len(cur_axes[0]), len(cur_axes[1])
        cnx, cny, cnz = (len(k) for k in keeps)
        # a coarse node is a dof iff the fine node under it is
        act_f = np.zeros(P_full.shape[0], dtype=bool)
        act_f[order] = True
        ci, cj, ck = np.meshgrid(np.arange(cnx), np.arange(cny), np.arange(cnz), indexing="ij")
        fine_under = keeps[0][ci] + fnx * (keeps[1][cj] + fny * keeps[2][ck])
        cgrid = ci + cnx * (cj + cny * ck)
        sel = act_f[fine_under.ravel()]
        c_order = np.sort(cgrid.ravel()[sel])
        if len(c_order) == 0 or len(c_order) == n:
            break
        P = P_full[order][:, c_order].tocsr()
        P.eliminate_zeros()
        lv["P"] = P
        A = (P.T @ A @ P).tocsr()
        A.sort_indices()
        order = c_order
        cur_axes = [a[k] for a, k in zip(cur_axes, keeps)]
    return levels


# --------------------------------------------------------------------------
# A curved mesh: the unit cube pushed through a smooth map, so that all six
# geometric factors are non-zero (the reference's Kershaw / pebble-bed inputs,
# run.py:36-74, are such meshes; they are not available here).
# --------------------------------------------------------------------------

class OracleAmgF32:
    """The V-cycle with Float = float (oracle/fdd_oracle_amg_f32.c) on a hierarchy given as Problem.amg_attach takes it."""

    def __init__(self, levels, cheby_order=2, num_vcycles=1):
        self.L = oracle()
        ip = ctypes.POINTER(ctypes.c_int)
        self.n = levels[0]["A"].shape[0]
        self.a = vp(self.L.orc_amg32_create(len(levels), cheby_order, num_vcycles))
        for l, lv in enumerate(levels):
            A = lv["A"].tocsr()
            A.sort_indices()
            arrs = [np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32), np.ascontiguousarray(A.data, dtype=np.float64),
                    np.ascontiguousarray(lv["D"], dtype=np.float64), np.ascontiguousarray(lv["coefs"], dtype=np.float64)]
            if lv.get("P") is not None:
                P = lv["P"].tocsr()
                P.sort_indices()
                parr = [np.ascontiguousarray(P.indptr, dtype=np.int32), np.ascontiguousarray(P.indices, dtype=np.int32), np.ascontiguousarray(P.data, dtype=np.float64)]
                pargs = (P.shape[1], parr[0].ctypes.data_as(ip), parr[1].ctypes.data_as(ip), _p(parr[2]))
            else:
                pargs = (0, None, None, None)
            self.L.orc_amg32_set_level(self.a, l, A.shape[0], arrs[0].ctypes.data_as(ip), arrs[1].ctypes.data_as(ip), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]), *pargs)  # copied inside

    def vcycle(self, f):
        u = np.zeros(self.n)
        self.L.orc_amg32_vcycle(self.a, _p(u), _p(np.ascontiguousarray(f, dtype=np.float64)))
        return u

    def close(self):
        if self.a:
            self.L.orc_amg32_destroy(self.a)
            self.a = None


class DeformedMesh(BoxMesh):
    """BoxMesh with x -> x + a*s(x,y,z)*(1, -0.7, 0.5), s = sin(pi x) sin(pi y) sin(pi z): the boundary stays put,
    elements stay conforming (one global map), the Jacobian is full.  Geometric factors are the isoparametric ones
    of degree N: G = w_i w_j w_k |J| J^-1 J^-T with J = dx/dr from the GLL differentiation matrix."""

    def __init__(self, E, N, amplitude=0.06, P=(1, 1, 1), rank=0):
        super().__init__(E, N, P, rank)
        n = N + 1
        _, w, D = gll(N)
        D = np.asarray(D).reshape(n, n)  # D[i, p] = D_hat[p + i*n]
        s = np.sin(np.pi * self.x) * np.sin(np.pi * self.y) * np.sin(np.pi * self.z)
        self._isoparametric([self.x + amplitude * s, self.y - 0.7 * amplitude * s, self.z + 0.5 * amplitude * s])

    def _isoparametric(self, X):
        """move the GLL points to X = [x, y, z] and recompute the six factors at the mesh's own degree"""
        n = self.N + 1
        _, w, D = gll(self.N)
        D = np.asarray(D).reshape(n, n)  # D[i, p] = D_hat[p + i*n]
        self.x, self.y, self.z = [np.ascontiguousarray(c) for c in X]
        ne = self.num_local_elements
        J = np.zeros((ne, n, n, n, 3, 3))
        for a, c in enumerate(X):
            c = c.reshape(ne, n, n, n)  # [e, k, j, i]
            J[..., a, 0] = np.einsum("ip,ekjp->ekji", D, c)
            J[..., a, 1] = np.einsum("jp,ekpi->ekji", D, c)
            J[..., a, 2] = np.einsum("kp,epji->ekji", D, c)
        det = np.linalg.det(J)
        assert (det > 0).all()
        Ji = np.linalg.inv(J)  # dr/dx
        M = np.einsum("...ab,...cb->...ac", Ji, Ji)  # J^-1 J^-T
        www = w[None, :, None, None] * w[None, None, :, None] * w[None, None, None, :]
        sc = www * det
        pairs = [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]
        self.g = [np.ascontiguousarray((sc * M[..., a, b]).reshape(-1)) for a, b in pairs]


def kershaw_map(eps_y, eps_z, x, y, z):
    """The generalized Kershaw map of the unit cube (D. Kershaw, JCP 39 (1981); the 3-D form of the CEED bake-off problems
    and of Nek5000's kershaw case, whose eps = 0.3 exports every experiment of the reference reads, run.py:25-47): x kept,
    six x-layers whose yz-sections go left-left, left-right, right-left (two layers), left-right, right-right.  An
    independent numpy statement of host/box_mesh.hpp's."""
    right = lambda eps, s: np.where(s <= 0.5, (2.0 - eps) * s, 1.0 + eps * (s - 1.0))
    left = lambda eps, s: 1.0 - right(eps, 1.0 - s)
    step = lambda a, b, t: np.where(t <= 0.0, a, np.where(t >= 1.0, b, a + (b - a) * t))
    layer = np.clip(np.floor(x * 6.0).astype(np.int64), 0, 5)
    lam = (x - layer / 6.0) * 6.0

    def section(eps, s):
        L, R = left(eps, s), right(eps, s)
        return np.select([layer == 0, (layer == 1) | (layer == 4), layer == 2, layer == 3], [L, step(L, R, lam), step(R, L, 0.5 * lam), step(R, L, 0.5 * (1.0 + lam))], default=R)

    return x, section(eps_y, y), section(eps_z, z)


class KershawMesh(DeformedMesh):
    """BoxMesh under the Kershaw map: the reference's experiment geometry (all six geometric factors non-zero)."""

    def __init__(self, E, N, eps=0.3, P=(1, 1, 1), rank=0, eps_z=None):
        BoxMesh.__init__(self, E, N, P, rank)
        self._isoparametric(list(kershaw_map(eps, eps if eps_z is None else eps_z, self.x, self.y, self.z)))


class QuadMesh(BoxMesh):
    """2-D counterpart of BoxMesh / DeformedMesh (the reference's `dim == 2` branches): unit square, E = (Ex, Ey)
    quadrilaterals of degree N, optionally deformed by x -> x + a*s*(1, -0.7), s = sin(pi x) sin(pi y).  Geometric
    factors in the 2-D layout of domain.okl:29-30, g_1 = G_rr, g_2 = G_ss, g_3 = G_rs (g_4..g_6 are read by
    Domain::initialize but unused), G = w_i w_j |J| J^-1 J^-T with J from the GLL differentiation matrix."""

    dim = 2

    def __init__(self, E, N, amplitude=0.0):
        Ex, Ey = E
        self.E, self.N, self.P, self.rank = (Ex, Ey, 1), N, (1, 1, 1), 0
        n = N + 1
        z, w, D = gll(N)
        D = np.asarray(D).reshape(n, n)
        ne = Ex * Ey
        self.local_E = (Ex, Ey, 1)
        self.origin = (0, 0, 0)
        self.num_local_elements = ne
        self.num_elem_points = n * n
        self.num_local_points = ne * n * n
        Gx, Gy = Ex * N + 1, Ey * N + 1
        self.global_nodes = Gx * Gy
        EY, EX = np.meshgrid(np.arange(Ey), np.arange(Ex), indexing="ij")
        EX, EY = EX.reshape(-1), EY.reshape(-1)
        j, i = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        i, j = i.reshape(-1), j.reshape(-1)  # x fastest
        gi = EX[:, None] * N + i[None, :]
        gj = EY[:, None] * N + j[None, :]
        V = (Ex + 1) * (Ey + 1)
        is_vertex = (gi % N == 0) & (gj % N == 0)
        self.glo_num = np.where(is_vertex, 1 + gi // N + (Ex + 1) * (gj // N), V + 1 + gi + Gx * gj).astype(np.int64).reshape(-1)

        def mult(g, G):
            m = np.ones_like(g)
            m[(g % N == 0) & (g > 0) & (g < G - 1)] = 2
            return m

        self.node_degree = (mult(gi, Gx) * mult(gj, Gy)).astype(np.int32).reshape(-1)
        on_bdry = (gi == 0) | (gi == Gx - 1) | (gj == 0) | (gj == Gy - 1)
        self.p_mask = np.where(on_bdry, 0.0, 1.0).reshape(-1)
        xi = 0.5 * (z + 1.0)
        x = ((EX[:, None] + xi[i][None, :]) / Ex).reshape(-1)
        y = ((EY[:, None] + xi[j][None, :]) / Ey).reshape(-1)
        s_ = np.sin(np.pi * x) * np.sin(np.pi * y)
        X = [x + amplitude * s_, y - 0.7 * amplitude * s_]
        self.x, self.y = [np.ascontiguousarray(c) for c in X]
        self.z = np.zeros_like(self.x)
        J = np.zeros((ne, n, n, 2, 2))
        for a, c in enumerate(X):
            c = c.reshape(ne, n, n)  # [e, j, i]
            J[..., a, 0] = np.einsum("ip,ejp->eji", D, c)
            J[..., a, 1] = np.einsum("jp,epi->eji", D, c)
        det = np.linalg.det(J)
        assert (det > 0).all()
        Ji = np.linalg.inv(J)
        M = np.einsum("...ab,...cb->...ac", Ji, Ji)
        sc = (w[None, :, None] * w[None, None, :]) * det
        zero = np.zeros(self.num_local_points)
        self.g = [np.ascontiguousarray((sc * M[..., a, b]).reshape(-1)) for a, b in ((0, 0), (1, 1), (0, 1))] + [zero, zero.copy(), zero.copy()]


class QuadMeshRanks(QuadMesh):
    """QuadMesh cut into P = (Px, Py) rank blocks (elements of a block x fastest), the 2-D counterpart of BoxMesh's
    partition: global ids, multiplicities, mask and geometry are those of the global mesh."""

    def __init__(self, E, N, P=(1, 1), rank=0, amplitude=0.0):
        full = QuadMesh(E, N, amplitude)
        Ex, Ey = E
        Px, Py = P
        assert Ex % Px == 0 and Ey % Py == 0
        lx, ly = Ex // Px, Ey // Py
        rx, ry = rank % Px, rank // Px
        ex = rx * lx + np.arange(lx)
        ey = ry * ly + np.arange(ly)
        elems = (ey[:, None] * Ex + ex[None, :]).reshape(-1)
        npts = full.num_elem_points
        pts = (elems[:, None] * npts + np.arange(npts)[None, :]).reshape(-1)
        self.E, self.N, self.P, self.rank = (Ex, Ey, 1), N, (Px, Py, 1), rank
        self.local_E = (lx, ly, 1)
        self.origin = (rx * lx, ry * ly, 0)
        self.num_local_elements = lx * ly
        self.num_elem_points = npts
        self.num_local_points = lx * ly * npts
        self.global_nodes = full.global_nodes
        for name in ("glo_num", "node_degree", "p_mask", "x", "y", "z"):
            setattr(self, name, np.ascontiguousarray(getattr(full, name)[pts]))
        self.g = [np.ascontiguousarray(a[pts]) for a in full.g]


def write_mesh_files(directory, mesh, proc_id=0):
    """The reference's per-rank input files (domain.tpp:45-224): lx1_<N+1>/{size,x,y,z,glo_num,node_degree,p_mask,g_1..g_6}_<rank>.<N>.dat"""
    N = mesh.N
    d = os.path.join(directory, "lx1_%d" % (N + 1))
    os.makedirs(d, exist_ok=True)
    n = N + 1
    dim = getattr(mesh, "dim", 3)
    with open(os.path.join(d, "size_%d.%d.dat" % (proc_id, N)), "w") as fh:
        fh.write("%d %d %d %d %d\n" % (dim, n, n, n if dim == 3 else 1, mesh.num_local_elements))

    def put(stem, arr, dtype):
        np.ascontiguousarray(arr, dtype=dtype).tofile(os.path.join(d, "%s_%d.%d.dat" % (stem, proc_id, N)))

    put("x", mesh.x, np.float64)
    put("y", mesh.y, np.float64)
    if dim == 3:
        put("z", mesh.z, np.float64)  # domain.tpp:121-138: no z file in 2-D
    put("glo_num", mesh.glo_num, np.int64)
    put("node_degree", mesh.node_degree, np.int32)
    put("p_mask", mesh.p_mask, np.float64)
    for g in range(6):
        put("g_%d" % (g + 1), mesh.g[g], np.float64)


def read_mesh_arrays(directory, N, proc_id=0, dim=3):
    """the arrays of one rank's file set (the reader's format, domain.tpp:45-224) as a dict"""
    d = os.path.join(directory, "lx1_%d" % (N + 1))
    get = lambda stem, dtype: np.fromfile(os.path.join(d, "%s_%d.%d.dat" % (stem, proc_id, N)), dtype=dtype)
    out = {"x": get("x", np.float64), "y": get("y", np.float64), "glo_num": get("glo_num", np.int64), "node_degree": get("node_degree", np.int32), "p_mask": get("p_mask", np.float64)}
    out["z"] = get("z", np.float64) if dim == 3 else np.zeros_like(out["x"])
    for g in range(6):
        out["g_%d" % (g + 1)] = get("g_%d" % (g + 1), np.float64)
    return out


def kershaw_mesh_of_the_host_layer(host_lib, directory, E, N, eps, P=(1, 1, 1), rank=0):
    """Rank `rank`'s Kershaw mesh as host/box_mesh.hpp generates it (written through fddh_write_kershaw_mesh_files and read
    back), held against the numpy statement of the same map and returned with the host layer's bits, so that an oracle
    world built from it sees the geometry of the product's ranks exactly (a long Krylov iteration on a strongly deformed
    mesh amplifies a last-bit difference of the factors past the comparison tolerances)."""
    import ctypes as C

    twin = KershawMesh(E, N, eps, P, rank)
    arr3 = lambda v: (C.c_int * 3)(*v)
    host_lib.call("fddh_write_kershaw_mesh_files", os.fsencode(directory), arr3(E), arr3(P), N, rank, C.c_double(eps), C.c_double(eps))
    a = read_mesh_arrays(directory, N, rank)
    gmax = max(np.abs(g).max() for g in twin.g)
    for name, ref in [("x", twin.x), ("y", twin.y), ("z", twin.z)] + [("g_%d" % (k + 1), twin.g[k]) for k in range(6)]:
        assert np.abs(a[name] - ref).max() <= 1e-13 * (gmax if name.startswith("g_") else 1.0), (N, rank, name)
    assert np.array_equal(a["glo_num"], twin.glo_num) and np.array_equal(a["node_degree"], twin.node_degree) and np.array_equal(a["p_mask"], twin.p_mask)
    twin.x, twin.y, twin.z = a["x"], a["y"], a["z"]
    twin.g = [a["g_%d" % (k + 1)] for k in range(6)]
    return twin


def oracle_stiffness(u, G, D, N, dim=3):
    """The oracle's two-kernel element stiffness (domain.okl:5-98) on plain arrays: (Au, GDu)."""
    L = oracle()
    vp = ctypes.c_void_p
    npts = len(u)
    GDu = [np.zeros(npts) for _ in range(3)]
    Au = np.zeros(npts)
    gd = (vp * 3)(*[a.ctypes.data for a in GDu])
    gg = (vp * 6)(*[a.ctypes.data for a in G])
    L.orc_dom_stiffness_matrix_1(gd, _p(u), _p(D), gg, npts, N, dim)
    L.orc_dom_stiffness_matrix_2(_p(Au), gd, _p(D), npts, N, dim)
    return Au, GDu
