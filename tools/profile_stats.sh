#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace statistics of one bench configuration.
# usage: tools/profile_stats.sh <tag> [bench args...]
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/bench.py --no-cpu-baseline --kernel-table "$@" > $out/stats.log 2>&1
cd $root
python3 tools/rocprof_summary.py stats $out/stats $out/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --kernel-table $*"
grep '^{' $out/stats.log | tail -n 1 > $out/bench_under_rocprof.json || true
rm -rf $out/stats
