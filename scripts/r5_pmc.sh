#!/bin/bash
# GPU box: one rocprofv3 --pmc pass per counter group of a bench.py command; prints per-kernel means.  Usage: r5_pmc.sh <tag> "<counters ...>" <bench args>
TAG=$1; CTR=$2; shift 2
O=$PWD/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $O/raw -- python3 bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 5 --repeats 2 "$@" > /dev/null 2> $O/err.txt
python3 - "$O" <<'PY'
import sys, glob, csv, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/raw/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    if len(next(iter(c.values()))) < 5: continue
    print(k[:60], {a: round(sum(v) / len(v), 1) for a, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
rm -rf $O/raw; tail -2 $O/err.txt
