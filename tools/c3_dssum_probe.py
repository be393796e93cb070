"""Development probe: the one dssum_block_kernel<0, weighted, masked> launch of a C3 run shows 86 ms in the kernel table (1.3 ms
expected).  Five dssum applications on a C3 problem under rocprofv3 --kernel-trace: is it the first launch, or every one?"""
import sys

import numpy as np

import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H

H.init(0)
H.comm_single()
H.set_print(False)
p = H.Problem.box((32, 32, 32), (1, 1, 1), 15, 6, True)
u = np.random.default_rng(1).uniform(-1, 1, p.n)
for weight in (True, True, False, True, True):
    out = p.dssum(u, True, weight)
    print("dssum weight=%d -> %.6e" % (weight, float(np.abs(out).max())), flush=True)
for _ in range(2):
    us, f = p.make_rhs_from(u.copy())
    print('make_rhs_from', float(np.abs(us).max()), flush=True)
p.close()
