#!/bin/bash
# Run on the GPU box: HBM read/write bytes per launch (two PMC passes) of one microbench section.
# usage: tools/pmc_microbench.sh <tag> <section>
set -e
tag=$1; section=$2
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 $root/tools/microbench.py --only $section > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 $root/tools/microbench.py --only $section > $out/pmc_write.log 2>&1
cd $root
python3 tools/rocprof_summary.py pmc $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json
rm -rf $out/pmc_fetch $out/pmc_write
