#!/usr/bin/env python3
"""Generate tests/golden/gll_tables.json from the REFERENCE's own speclib.

Runs only in the build container: it needs oracle/_ref/libspeclib_ref.so, which
oracle/Makefile (`make -C oracle ref`) compiles from
/root/reference/special_functions.f with flang (-fdefault-real-8, as the
reference Makefile:52).  The fixture holds data only -- GLL nodes, weights,
the derivative matrix exactly as Domain::initialize stores it
(domain.tpp:311-316: second dgll_ argument, read row-major) and the
coarse-to-fine interpolators exactly as Subdomain builds them
(subdomain.tpp:153-159: J[i*n_c + (j-1)] = hgll_(j, xi_f[i], xi_c, n_c)).

    python tests/golden/make_gll_golden.py
"""
import ctypes
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(ROOT, "oracle", "_ref", "libspeclib_ref.so")

MAX_N = 15  # polynomial degrees 1..15


def main():
    L = ctypes.CDLL(SO)
    L.hgll_.restype = ctypes.c_double
    vp = ctypes.c_void_p

    def zw(n):
        z = np.zeros(n)
        w = np.zeros(n)
        L.zwgll_(vp(z.ctypes.data), vp(w.ctypes.data), ctypes.byref(ctypes.c_int(n)))
        return z, w

    tables = {"source": "special_functions.f (zwgll_, dgll_, hgll_) via flang -fdefault-real-8", "levels": {}, "J_cf": {}}
    nodes = {}
    for N in range(1, MAX_N + 1):
        n = N + 1
        z, w = zw(n)
        Dt = np.zeros(n * n)
        D = np.zeros(n * n)
        cn = ctypes.c_int(n)
        L.dgll_(vp(Dt.ctypes.data), vp(D.ctypes.data), vp(z.ctypes.data), ctypes.byref(cn), ctypes.byref(cn))
        nodes[N] = z
        tables["levels"][str(N)] = {"z": z.tolist(), "w": w.tolist(), "D_hat": D.tolist()}

    for N_f in range(2, MAX_N + 1):
        for N_c in range(1, N_f):
            n_f, n_c = N_f + 1, N_c + 1
            J = np.zeros(n_f * n_c)
            zc = nodes[N_c].copy()
            for i in range(n_f):
                for j in range(1, n_c + 1):
                    x = ctypes.c_double(nodes[N_f][i])
                    J[i * n_c + (j - 1)] = L.hgll_(ctypes.byref(ctypes.c_int(j)), ctypes.byref(x), vp(zc.ctypes.data), ctypes.byref(ctypes.c_int(n_c)))
            tables["J_cf"][f"{N_c},{N_f}"] = J.tolist()

    out = os.path.join(HERE, "gll_tables.json")
    with open(out, "w") as fh:
        json.dump(tables, fh)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
