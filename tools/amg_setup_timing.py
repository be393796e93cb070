#!/usr/bin/env python3
"""Host time of the low-order preconditioner's setup on the GPU box: FEM matrix, hierarchy, upload (the lines
`low_order: ...` of a verbose build).  python tools/amg_setup_timing.py [elements per direction] [degree]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
H.init(0); H.comm_single(); H.set_print(False)
e = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 7
t = time.time(); p = H.Problem.box((e, e, e), (1, 1, 1), N, N - 1, True); print("problem %.1f s" % (time.time() - t), flush=True)
t = time.time(); nl = p.amg_build(verbose=True); print("levels", nl, "build %.1f s" % (time.time() - t), flush=True)
L = p.amg_levels()
nnz = [lv["A"].nnz for lv in L]
print("rows / nnz per level:", [(lv["A"].shape[0], lv["A"].nnz) for lv in L], "operator complexity %.3f" % (sum(nnz) / nnz[0]))
