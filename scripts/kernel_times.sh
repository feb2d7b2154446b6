#!/bin/bash
# GPU box: rocprofv3 kernel statistics (mean duration per kernel) of bench.py for a list of configs.
# Usage: scripts/kernel_times.sh <tag> "C2" "C4 --walkers 512" ...   -> gpurun_out/<tag>/<CFG>_kernel_stats.csv
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for spec in "$@"; do
  read -r -a a <<< "$spec"; cfg=${a[0]}
  rm -rf $OUT/tr
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python3 bench.py --config "${a[@]}" --no-cpu-baseline --no-extras --steps 30 --warmup 5 > $OUT/$cfg.json 2> $OUT/$cfg.err
  f=$(find $OUT/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${cfg}_kernel_stats.csv
  rm -rf $OUT/tr
  echo "== $spec"; head -6 $OUT/${cfg}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
