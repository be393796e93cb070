#!/bin/bash
# A/B of the XCD-windowed row-block order in the plan SpMVs (Qt multiply = short rows, 27-point stencil = wide rows) on ONE box.
cd "$(dirname "$0")/.."
common="--steps 4 --warmup 1 --no-reference-default --no-time-to-tolerance --no-cpu-baseline --no-kershaw"
show() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], " ".join("%s=%.1fus(%.3f)" % (k.split(",")[0].split(" (")[0], v["avg_us"], v["frac_moved_of_hbm_peak"]) for k, v in d["spmv"].items()), flush=True)
PY
}
for w in 0 8 32 128 0 32; do
    FDD_TUNE_CSR_XCD_SHORT=$w FDD_TUNE_CSR_XCD=$w python3 bench.py $common > gpurun_out/ab2_w$w.json 2>/dev/null && show gpurun_out/ab2_w$w.json "csr window $w:"
done
for w in 16 -1 32; do
    FDD_TUNE_MFMA_XCD_WINDOW=$w python3 bench.py --steps 6 --warmup 2 --no-reference-default --no-time-to-tolerance --no-stencil --no-cpu-baseline --no-kershaw --kernel-table --degree 15 > gpurun_out/ab2_c3_w$w.json 2>/dev/null && python3 - gpurun_out/ab2_c3_w$w.json "C3 mfma window $w:" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], " ".join("%s=%.1fus" % (k, v["avg_us"]) for k, v in d["kernels"].items() if "stiffness" in k), flush=True)
PY
done
