#!/bin/bash
# Run on the GPU box: one rocprofv3 --pmc pass of a microbench section with the given counters.
# usage: tools/pmc_counters.sh <tag> "<counters>" <microbench args...>
set -e
tag=$1; counters=$2; shift 2
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/pmc -o run -- python3 $root/tools/microbench.py "$@" > $out/run.log 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for path in f:
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
with open(out + "/counters.txt", "w") as fh:
    for k in agg:
        if "stiffness" in k or "csr" in k or "dssum" in k:
            fh.write(k + "\n")
            for c, v in agg[k].items():
                fh.write("   %-28s %.4g per launch (%d launches)\n" % (c, v / cnt[(k, c)], cnt[(k, c)]))
print(open(out + "/counters.txt").read())
PY
rm -rf $out/pmc
