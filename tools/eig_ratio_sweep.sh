#!/bin/bash
# Development sweep: lower end of the Chebyshev smoothing interval in front of the lattice-coarsening steps (fraction of
# lambda_max), on the box and on the Kershaw mesh (32^3 x N = 7), reference-default preconditioner, GMRES(20) outside.
cd "$(dirname "$0")/.."
for mesh in box kershaw; do
    for r in 0.15 0.08 0.04 0.02 0.01; do
        FDD_TUNE_AMG_GEOMETRIC_EIG_RATIO=$r python3 bench.py --mesh $mesh --outer gmres --amg --steps 4 --warmup 1 --no-stencil --no-cpu-baseline --no-kernel-timing --no-reference-default --no-kershaw > gpurun_out/eig_sweep.json 2> gpurun_out/eig_sweep.err
        python3 - "$mesh ratio $r" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/eig_sweep.json") if l.startswith("{")][-1])
t = d["to_1e-7"]
print("%-24s to 1e-7: %3d its, %6.1f ms, final relative residual %.2e" % (sys.argv[1], t["iterations"], t["time_ms"], t["relative_residual"]), flush=True)
PY
    done
done
