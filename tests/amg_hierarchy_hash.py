#!/usr/bin/env python3
"""SHA-1 of every level (A, P) of the low-order hierarchy of a box and of a deformed mesh, on the CPU stand-in of the
kernel C-ABI (test infrastructure): setup changes must leave these bit for bit alone, whatever the thread count.
python tests/amg_hierarchy_hash.py [tmpdir]"""
import hashlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import support as S
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H, lib
lib._host = lib._Lib(os.path.join(ROOT, "tests/cpu_shim/_build/libfdd_host_cpu.so"), os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
H.init(0, use_torch_stream=False); H.comm_single(); H.set_print(False)


def digest(p):
    h = hashlib.sha1()
    for lv in p.amg_levels():
        for M in (lv["A"], lv.get("P")):
            if M is None:
                continue
            M = M.tocsr()
            for a in (M.indptr, M.indices, M.data):
                h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp()
    p = H.Problem.box((6, 5, 4), (1, 1, 1), 7, 6, True)
    p.amg_build(coarsest_size=40)
    print("box 6x5x4 N=7:", digest(p))
    p.close()
    p = H.Problem.box((5, 5, 5), (1, 1, 1), 3, 2, True)
    p.amg_build(coarsest_size=40)
    print("box 5^3 N=3:", digest(p))
    p.close()
    E, N, red = (3, 2, 2), 7, 6
    for deg in S.level_degrees(N, red):
        S.write_mesh_files(d, S.DeformedMesh(E, deg, 0.05))
    p = H.Problem.from_directory(d, N, red)
    p.amg_build(coarsest_size=30)
    print("deformed 3x2x2 N=7:", digest(p))
    p.close()


main()
