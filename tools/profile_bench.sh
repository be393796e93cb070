#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + two PMC passes of the default bench.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-reference-default --no-time-to-tolerance --no-stencil --kernel-table "$@" > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-default --no-time-to-tolerance --no-stencil "$@" > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-default --no-time-to-tolerance --no-stencil "$@" > $out/pmc_write.log 2>&1
cd $root
# the workload the counters belong to (bench.py --elements / --degree; defaults = config C2)
export FDD_PROFILE_ELEMENTS=32 FDD_PROFILE_DEGREE=7
args=("$@")
for ((i = 0; i < ${#args[@]}; i++)); do
    [ "${args[$i]}" = "--elements" ] && FDD_PROFILE_ELEMENTS=${args[$((i + 1))]}
    [ "${args[$i]}" = "--degree" ] && FDD_PROFILE_DEGREE=${args[$((i + 1))]}
done
python3 tools/rocprof_summary.py stats $out/stats $out/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --kernel-table $*"
python3 tools/rocprof_summary.py pmc $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json
# keep only the small summaries (the traces hold torch's kilobyte-long kernel names)
cp $out/stats/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || cp $(find $out/stats -name '*kernel_stats.csv' | head -n 1) $out/kernel_stats.csv
grep '^{' $out/stats.log | tail -n 1 > $out/bench_under_rocprof.json || true
rm -rf $out/stats $out/pmc_fetch $out/pmc_write
