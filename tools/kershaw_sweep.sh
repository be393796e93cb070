#!/bin/bash
# Development sweep on the Kershaw mesh (eps = 0.3, 32^3 x N = 7): the smoothing interval in front of the hierarchy's
# lattice-coarsening steps and the depth of the geometric coarsening, reference-default preconditioner, GMRES(20) outside.
cd "$(dirname "$0")/.."
run() {
    env "$@" python3 bench.py --mesh kershaw --outer gmres --amg --steps 4 --warmup 1 --no-stencil --no-cpu-baseline --no-kernel-timing --no-reference-default > gpurun_out/kershaw_sweep.json 2> gpurun_out/kershaw_sweep.err
    python3 - "$*" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/kershaw_sweep.json") if l.startswith("{")][-1])
print("%-60s %.2f ms/Arnoldi step | to 1e-7: %s its, %.0f ms, converged %s" % (sys.argv[1], d["ms_per_step"], d["to_1e-7"]["iterations"], d["to_1e-7"]["time_ms"], d["to_1e-7"]["converged"]), flush=True)
PY
}
run FDD_NOP=1
run FDD_TUNE_AMG_GEOMETRIC_EIG_RATIO=0.3
run FDD_TUNE_AMG_GEOMETRIC_EIG_RATIO=0.08
run FDD_TUNE_AMG_GEOMETRIC_EIG_RATIO=0.04
run FDD_TUNE_AMG_GEOMETRIC_MIN_NODES=3
run FDD_TUNE_AMG_GEOMETRIC=0
