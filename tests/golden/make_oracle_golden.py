#!/usr/bin/env python3
"""Generate tests/golden/oracle_c1.json: the oracle's outputs on BASELINE
config C1 (4^3 elements, N=3, single subdomain) with seeded inputs.

The reference ships no golden vectors for this path, so
these are the build's own: they freeze the CPU restatement (any later change
of the oracle or of the mesh generator shows up as a diff) and give the GPU
tests a fixture that does not depend on rebuilding the oracle.  Data only:
iteration counts, residual histories, SHA-256 of the bit-exact operator
outputs and a few sample values.

    python tests/golden/make_oracle_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import support as S  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    E, N, red = (4, 4, 4), 3, 2
    m = S.BoxMesh(E, N)
    W = S.OracleWorld([m], N)
    sd = S.OracleSubdomain(E, N, red)
    n = m.num_local_points
    u = S.seeded_uniform(n, 1234)

    out = {"config": {"E": list(E), "N": N, "reduction": red, "seed": 1234}, "operators": {}, "solves": {}}

    ops = {
        "dssum_mask": W.dssum([u], True, False)[0],
        "dssum_mask_weight": W.dssum([u], True, True)[0],
        "dssum_plain": W.dssum([u], False, False)[0],
        "stiffness": W.stiffness([u])[0],
        "stiffness_dssum": W.stiffness([u], True)[0],
        "sub_tree": sd.tree(u),
        "sub_stiffness": sd.stiffness(u),
        "sub_dssum": sd.dssum(u),
    }
    for k, v in ops.items():
        out["operators"][k] = {"sha256": digest(v), "first": v[:4].tolist(), "sum": float(np.sum(v)), "absmax": float(np.abs(v).max())}
    out["operators"]["residual_norm"] = W.residual_norm([u])
    out["operators"]["sub_residual_norm"] = sd.residual_norm(u)

    us = W.dssum([u], True, True)
    f = W.stiffness(us)
    out["rhs"] = {"u_star_sha256": digest(us[0]), "f_sha256": digest(f[0])}

    def pre_factory(inner):
        def pre(z, r):
            o, _, _ = sd.solve(r[0], inner)
            z[0][:] = o

        return pre

    for outer in ("fcg", "gmres"):
        for inner in (None, "gmres", "fcg"):
            uu, its, hist = W.solve(f, outer, precond=None if inner is None else pre_factory(inner))
            out["solves"][f"{outer}+{inner or 'none'}"] = {
                "iterations": its,
                "history": hist.tolist(),
                "error_inf": float(np.abs(uu[0] - us[0]).max()),
                "u_first": uu[0][:4].tolist(),
            }

    r = S.seeded_uniform(n, 5) - 0.5
    for inner in ("gmres", "fcg"):
        z, its, hist = sd.solve(r, inner)
        out["solves"][f"precond_{inner}"] = {"iterations": its, "history": hist.tolist(), "z_first": z[:4].tolist(), "z_absmax": float(np.abs(z).max())}

    path = os.path.join(HERE, "oracle_c1.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
