#!/usr/bin/env python3
"""What the composite code path costs on identical work: config C2 on one GPU through the conforming path and through
the composite path (FDDH_FORCE_COMPOSITE: a one-rank composite has no rings and no superdomain, but runs the
dof-space composite solve: one index array for all points, hanging-point and superdomain stages present and empty)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from polynomial_reduction_with_full_domain_decomposition_preconditioner_amd import host_api as H


def run(force, steps=20, e=32):
    p = H.Problem.box((e, e, e), (1, 1, 1), 7, 6, True, force_composite=force)
    p.set_flag("sub_use_preconditioner", 0)
    _, f = p.make_rhs(function_id=4, seed=1234)
    p.pcg_begin(f)
    p.pcg_steps(3)
    torch.cuda.synchronize()
    t = time.perf_counter()
    last = p.pcg_steps(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps * 1e3
    si = p.sub_info()
    p.close()
    return {"force_composite": force, "ms_per_step": dt, "last": last, "is_composite": si["is_composite"]}


if __name__ == "__main__":
    H.init(0)
    H.comm_single()
    H.set_print(False)
    for force in (False, True):
        print(json.dumps(run(force)))
