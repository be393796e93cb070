#!/bin/bash
# the round's closing measurements on one box: default bench, 2- and 4-rank rehearsals (through gpurun)
python bench.py > gpurun_out/r03_bench_c2.json 2> gpurun_out/r03_bench_c2.err; echo "rc=$?"
timeout -k 10 300 python bench.py --rehearse-ranks 2 --steps 5 --warmup 1 > gpurun_out/r03_bench_2rank_rehearsal.json 2> gpurun_out/r03_bench_2rank_rehearsal.err; echo "rc=$?"
timeout -k 10 500 python bench.py --rehearse-ranks 4 --steps 5 --warmup 1 > gpurun_out/r03_bench_4rank_rehearsal.json 2> gpurun_out/r03_bench_4rank_rehearsal.err; echo "rc=$?"
python3 - <<'PY'
import json
for f in ("c2", "2rank_rehearsal", "4rank_rehearsal"):
    r = json.load(open("gpurun_out/r03_bench_%s.json" % f)); rd = r["reference_default"]
    print(f, r["ms_per_step"], rd["f64"]["ms_per_step"], rd["f64"]["to_1e-7"]["iterations"], rd["f64"]["to_1e-7"]["time_ms"], rd["f32"]["to_1e-7"]["iterations"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["value"])
PY
