#!/bin/bash
# GPU box helper of round 3: C1 bench lines for a list of environment settings (one library).
# Usage: scripts/r3_env.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...    ("-" = no extra environment)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
OUT=gpurun_out/$TAG/env.txt
: > $OUT
for round in 1 2; do
for spec in "$@"; do
  [ "$spec" = "-" ] && envs=() || read -r -a envs <<< "$spec"
  for w in ${WS:-256 512}; do
    env "${envs[@]}" python bench.py --no-cpu-baseline --no-extras --walkers $w --steps 400 2>>$OUT.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$spec]', d['config']['walkers_per_gpu'], round(d['value']), 'us/step', round(1e3*d['ms_per_step'],2))" >> $OUT
  done
done
done
cat $OUT
