#!/bin/bash
# A/B of the XCD-windowed workgroup orders (csrc/fdd_common.h: fdd_xcd_windowed_block) on ONE box, in the solver:
# the Qt gather's row blocks at C2 and the matrix-core stiffness kernel's element order at C3.
# usage (through gpurun): bash tools/xcd_window_ab.sh [c2|c3|both]
what=${1:-both}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
common="--steps 6 --warmup 2 --no-reference-default --no-time-to-tolerance --no-stencil --no-cpu-baseline --no-kershaw --kernel-table"
show() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
keys = [k for k in d["kernels"] if any(s in k for s in ("stiffness", "gather", "dssum"))]
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], " ".join("%s=%.1fus" % (k, d["kernels"][k]["avg_us"]) for k in keys), "Qt_spmv=%.1fus" % d["spmv"]["Qt (gather, 1-8 nnz/row)"]["avg_us"], flush=True)
PY
}
if [ "$what" != "c3" ]; then
    for w in 0 2 8 32 0 8; do
        FDD_TUNE_DSSUM_XCD_WINDOW=$w python3 bench.py $common > gpurun_out/ab_c2_w$w.json 2>/dev/null && show gpurun_out/ab_c2_w$w.json "C2 dssum window $w:"
    done
fi
if [ "$what" != "c2" ]; then
    for w in 0 -1 8 0 -1; do
        FDD_TUNE_MFMA_XCD_WINDOW=$w python3 bench.py $common --degree 15 > gpurun_out/ab_c3_w$w.json 2>/dev/null && show gpurun_out/ab_c3_w$w.json "C3 mfma window $w:"
    done
fi
