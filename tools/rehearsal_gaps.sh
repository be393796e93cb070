#!/bin/bash
# Kernel-gap summary of the 8-rank rehearsal (ranks as threads sharing ONE GPU: what is measured is how the eight ranks'
# streams interleave on one device, not the node): rocprofv3 kernel trace -> tools/kernel_gaps.py.
cd "$(dirname "$0")/.."
root=$(pwd)
out=$root/gpurun_out/r04_gaps
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o run -- python3 $root/bench.py --rehearse-ranks 8 --steps 3 --warmup 1 --no-cpu-baseline --no-reference-default --no-time-to-tolerance > $out/run.log 2>&1
cd $root
csv=$(find $out/trace -name '*kernel_trace.csv' | head -n 1)
python3 tools/kernel_gaps.py $csv > $out/kernel_gaps_8rank_rehearsal.md
head -40 $out/kernel_gaps_8rank_rehearsal.md
rm -rf $out/trace
