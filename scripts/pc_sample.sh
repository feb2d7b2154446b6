#!/bin/bash
# GPU box, round 5: PC-sampling histogram of one bench.py command (rocprofv3 --pc-sampling-beta-enabled), for the per-phase
# instruction budget of walker_kernel / tile_kernel1.  The library is the -gline-tables-only build (same ISA as the product's:
# scripts/README.md), so that every sampled instruction carries its source line.
# Usage: scripts/pc_sample.sh <tag> <method: host_trap|stochastic> <unit> <interval> <bench args...>
TAG=$1; METHOD=$2; UNIT=$3; IVAL=$4; shift 4
REPO=$PWD
O=$REPO/gpurun_out/pcs_$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
export RBVFIT_AMD_LIB=$REPO/rbvfit_amd/lib/exp/librbvfit_amd_g.so
cd /tmp
timeout -k 10 ${PCS_TIMEOUT:-240} rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $IVAL \
    --kernel-trace --output-format csv -d $O/raw -- python3 $REPO/bench.py "$@" > $O/bench.json 2> $O/err.txt
rc=$?
echo "rocprofv3 rc=$rc" > $O/status.txt
find $O/raw -type f | head -20 >> $O/status.txt
f=$(find $O/raw -name "*pc_sampling*.csv" | head -1)
k=$(find $O/raw -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then
  head -3 "$f" > $O/sample_head.txt
  python3 $REPO/scripts/pc_hist.py "$f" "$k" > $O/hist.txt 2>> $O/err.txt
  head -40 $O/hist.txt
fi
rm -rf $O/raw
tail -3 $O/err.txt; cat $O/status.txt | head -5
exit 0
