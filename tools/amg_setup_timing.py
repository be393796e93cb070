import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H
H.init(0); H.comm_single(); H.set_print(False)
e = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t = time.time(); p = H.Problem.box((e, e, e), (1, 1, 1), 7, 6, True); print("problem %.1f s" % (time.time() - t))
t = time.time(); nl = p.amg_build(verbose=True); print("levels", nl, "build %.1f s" % (time.time() - t))
