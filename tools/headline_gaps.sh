#!/bin/bash
# Kernel-gap / overlap summary of the headline PCG step on one GPU: rocprofv3 kernel trace -> tools/kernel_gaps.py
# (how much of the one-workgroup launches' dispatch-to-completion time is hidden under their neighbours).
cd "$(dirname "$0")/.."
root=$(pwd)
out=$root/gpurun_out/r04_headline_gaps
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o run -- python3 $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-default --no-time-to-tolerance --no-stencil --no-kershaw --no-kernel-timing > $out/run.log 2>&1
cd $root
csv=$(find $out/trace -name '*kernel_trace.csv' | head -n 1)
python3 tools/kernel_gaps.py $csv > $out/kernel_gaps_headline.md
cat $out/kernel_gaps_headline.md
grep '^{' $out/run.log | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step under the trace: %.4f' % d['ms_per_step'])" | tee -a $out/kernel_gaps_headline.md
rm -rf $out/trace
