#!/bin/bash
# GPU box: bench lines of C2..C4 (CFGS) for library builds under rbvfit_amd/lib (labels = paths relative to it), one round.
# Usage: scripts/exp_cfg_libs.sh <tag> label=lib ...
TAG=$1; shift
mkdir -p gpurun_out/$TAG
specs=()
for s in "$@"; do specs+=("${s%%=*}=$PWD/rbvfit_amd/lib/${s#*=}"); done
scripts/exp_cfg.sh gpurun_out/$TAG/cfg.txt "${specs[@]}" > /dev/null
cat gpurun_out/$TAG/cfg.txt
