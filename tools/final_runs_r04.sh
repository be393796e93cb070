#!/bin/bash
# Round 4's closing measurements on ONE box (through gpurun): the driver's command, its rocprofv3 stats + PMC passes at C2
# and C3, the outer-GMRES kernel tables, the 8-rank rehearsal.  Summaries land in gpurun_out/r04/ and are copied to profiles/.
cd "$(dirname "$0")/.."
out=gpurun_out/r04
mkdir -p $out
python3 bench.py > $out/bench_c2.json 2> $out/bench_c2.err; echo "bench c2 rc=$?"
bash tools/profile_bench.sh r04_c2 --no-kershaw; echo "profile c2 rc=$?"
python3 bench.py --degree 15 --no-kershaw > $out/bench_c3_n15.json 2> $out/bench_c3_n15.err; echo "bench c3 rc=$?"
bash tools/profile_bench.sh r04_c3 --degree 15 --no-kershaw; echo "profile c3 rc=$?"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for tag in gmres gmres_amg; do
    extra="--outer gmres"; [ $tag = gmres_amg ] && extra="--outer gmres --amg --no-amg-graph"
    rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/stats_$tag -o run -- python3 $root/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-reference-default --no-time-to-tolerance --no-stencil --no-kershaw --kernel-table $extra > $root/$out/stats_$tag.log 2>&1
    (cd $root && python3 tools/rocprof_summary.py stats $out/stats_$tag $out/bench_c2_${tag}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --kernel-table $extra" && grep '^{' $out/stats_$tag.log | tail -n 1 > $out/bench_c2_${tag}_under_rocprof.json; rm -rf $out/stats_$tag)
done
cd $root
timeout -k 10 900 python3 bench.py --rehearse-ranks 8 --steps 5 --warmup 1 > $out/bench_8rank_rehearsal.json 2> $out/bench_8rank_rehearsal.err; echo "rehearsal rc=$?"
ls -la $out gpurun_out/r04_c2 gpurun_out/r04_c3
