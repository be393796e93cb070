#!/usr/bin/env python3
"""Row-length statistics of the low-order hierarchy's levels (test infrastructure: CPU stand-in of the kernel C-ABI)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H, lib
lib._host = lib._Lib(os.path.join(ROOT, "tests/cpu_shim/_build/libfdd_host_cpu.so"), os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
H.init(0, use_torch_stream=False); H.comm_single(); H.set_print(False)
e = int(sys.argv[1]) if len(sys.argv) > 1 else 12
p = H.Problem.box((e, e, e), (1, 1, 1), 7, 6, True)
p.amg_build()
for l, lv in enumerate(p.amg_levels()):
    A = lv["A"].tocsr()
    lens = np.diff(A.indptr)
    n = len(lens)
    pad_nat = sum(int(lens[i:i + 64].max()) * min(64, n - i) for i in range(0, n, 64)) / max(lens.sum(), 1)
    srt = np.sort(lens)[::-1]
    pad_sorted = sum(int(srt[i:i + 64].max()) * min(64, n - i) for i in range(0, n, 64)) / max(lens.sum(), 1)
    # sorting inside windows of 4096 rows only (keeps locality)
    pad_win = 0
    for w0 in range(0, n, 4096):
        ws = np.sort(lens[w0:w0 + 4096])[::-1]
        pad_win += sum(int(ws[i:i + 64].max()) * min(64, len(ws) - i) for i in range(0, len(ws), 64))
    pad_win /= max(lens.sum(), 1)
    print("level %d rows %8d nnz %9d  len min/mean/max %3d/%6.1f/%4d  SELL padding: natural %.2f  sorted %.2f  window-4096 %.2f" % (l, n, lens.sum(), lens.min(), lens.mean(), lens.max(), pad_nat, pad_sorted, pad_win))
