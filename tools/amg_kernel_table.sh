#!/bin/bash
# development: rocprofv3 kernel table of the reference-default configuration under outer GMRES (V-cycle launched kernel by kernel),
# with extra environment settings given as NAME=VALUE arguments; summary in gpurun_out/amg_table_<tag>.md
tag=$1; shift
for kv in "$@"; do export "$kv"; done
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -o run -- python3 $root/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-reference-default --no-time-to-tolerance --no-stencil --no-kershaw --kernel-table --outer gmres --amg --no-amg-graph > $out/stats_$tag.log 2>&1
cd $root && python3 tools/rocprof_summary.py stats gpurun_out/stats_$tag gpurun_out/amg_table_$tag.md "amg kernel table $tag $*" && python3 tools/rocprof_summary.py bygrid gpurun_out/stats_$tag gpurun_out/amg_bygrid_$tag.md "amg launches by grid size $tag $*" && rm -rf gpurun_out/stats_$tag
head -16 gpurun_out/amg_table_$tag.md; head -40 gpurun_out/amg_bygrid_$tag.md
