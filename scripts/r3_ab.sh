#!/bin/bash
# GPU box helper of round 3: optional GPU test run, then A/B of library builds on C1.
# Usage: scripts/r3_ab.sh <tag> <tests: 0|1> label=lib ...   (lib paths relative to rbvfit_amd/lib)
TAG=$1; TESTS=$2; shift 2
mkdir -p gpurun_out/$TAG
if [ "$TESTS" = "1" ]; then
  python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.txt 2>&1
  tail -5 gpurun_out/$TAG/pytest.txt
fi
specs=()
for s in "$@"; do specs+=("${s%%=*}=$PWD/rbvfit_amd/lib/${s#*=}"); done
WS="${WS:-256 512}" CFGS="${CFGS:- }" scripts/exp_ab.sh gpurun_out/$TAG/ab.txt "${specs[@]}" > /dev/null
cat gpurun_out/$TAG/ab.txt
