#!/usr/bin/env python3
"""Hierarchy of the low-order preconditioner: operator complexity and outer PCG iterations to 1e-7 (reference-default
preconditioner: V-cycle inside every inner GMRES step) for the geometric leading levels against plain smoothed
aggregation (FDD_TUNE_AMG_GEOMETRIC=0), on the CPU stand-in of the kernel C-ABI (test infrastructure).
python tests/amg_hierarchy_compare.py [elements per direction] [N] [geometric 0/1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
e = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 7
os.environ["FDD_TUNE_AMG_GEOMETRIC"] = sys.argv[3] if len(sys.argv) > 3 else "1"
import numpy as np
import support as S
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H, lib
lib._host = lib._Lib(os.path.join(ROOT, "tests/cpu_shim/_build/libfdd_host_cpu.so"), os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
H.init(0, use_torch_stream=False); H.comm_single(); H.set_print(False)
p = H.Problem.box((e, e, e), (1, 1, 1), N, 6 if N == 7 else 2, True)
t = time.time(); nl = p.amg_build(verbose=True); tb = time.time() - t
L = p.amg_levels()
nnz = [lv["A"].nnz for lv in L]
print("levels", [(lv["A"].shape[0], lv["A"].nnz, round(lv["A"].nnz / lv["A"].shape[0], 1)) for lv in L])
print("operator complexity %.3f, grid complexity %.3f, build %.1f s" % (sum(nnz) / nnz[0], sum(lv["A"].shape[0] for lv in L) / L[0]["A"].shape[0], tb))
_, f = p.make_rhs(function_id=4, seed=1234)
t = time.time(); u, its, hist = p.solve(f, "fcg"); ts = time.time() - t
print("geometric=%s e=%d N=%d: outer iterations %d, history %s, solve %.1f s" % (os.environ["FDD_TUNE_AMG_GEOMETRIC"], e, N, its, ["%.2e" % (h / hist[0]) for h in hist], ts))
