#!/usr/bin/env python3
"""Host-side cost of building the low-order hierarchy (test infrastructure: runs the host layer on the CPU stand-in of
the kernel C-ABI).  python tests/amg_setup_timing.py [E]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H, lib
lib._host = lib._Lib(os.path.join(ROOT, "tests/cpu_shim/_build/libfdd_host_cpu.so"), os.path.join(lib.INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
H.init(0, use_torch_stream=False); H.comm_single(); H.set_print(False)
e = int(sys.argv[1]) if len(sys.argv) > 1 else 16
p = H.Problem.box((e, e, e), (1, 1, 1), 7, 6, True)
t = time.time(); nl = p.amg_build(verbose=True); print("levels", nl, "build %.1f s" % (time.time() - t))
