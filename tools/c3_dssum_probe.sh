#!/bin/bash
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/dssum_probe -o run -- python3 $root/tools/c3_dssum_probe.py > $root/gpurun_out/dssum_probe.log 2>&1
cd $root && python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/dssum_probe/**/*_kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "dssum" in r["Kernel_Name"] or "sell_fill" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:80], r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
tail -6 gpurun_out/dssum_probe.log; rm -rf gpurun_out/dssum_probe
