#!/bin/bash
# The reference-default preconditioner on the Kershaw mesh (eps = 0.3, 32^3 elements of degree 7) with the hierarchy's
# leading levels geometric (default) or left to the smoothed aggregation: iterations and time to 1e-7.
cd "$(dirname "$0")/.."
for geo in 1 0; do
    FDD_TUNE_AMG_GEOMETRIC=$geo python3 bench.py --mesh kershaw --steps 5 --warmup 1 --no-stencil --no-cpu-baseline > gpurun_out/kershaw_geo$geo.json 2> gpurun_out/kershaw_geo$geo.err
    python3 - gpurun_out/kershaw_geo$geo.json "geometric=$geo" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rd, rg = d["reference_default"], d["reference_default_gmres"]
print(sys.argv[2], "levels", rd["amg_levels"], "setup %.1fs" % rd["amg_setup_s"], "| PCG f64: %.2f ms/step, to 1e-7: %s" % (rd["f64"]["ms_per_step"], rd["f64"].get("to_1e-7")), "| GMRES f64: %s" % rg["f64"].get("to_1e-7"), "| headline to 1e-7: %s" % d["to_1e-7"], flush=True)
PY
done
