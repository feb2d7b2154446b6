#!/bin/bash
# GPU box: bench lines of C2..C4 (CFGS) for a list of environment settings, two rounds (one library).
# Usage: scripts/cfg_env.sh <tag> "ENV1=a ENV2=b" "-" ...    ("-" = no extra environment)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
OUT=gpurun_out/$TAG/cfg_env.txt
: > $OUT
for round in 1 2; do
for spec in "$@"; do
  [ "$spec" = "-" ] && envs=() || read -r -a envs <<< "$spec"
  for cfg in ${CFGS:-C2 C3 C4}; do
    extra=""
    [ "$cfg" = "C3" ] && extra="--walkers ${C3W:-2048}"
    [ "$cfg" = "C4" ] && extra="--walkers ${C4W:-512}"
    env "${envs[@]}" python bench.py --no-cpu-baseline --no-extras --config $cfg $extra --steps ${STEPS:-40} --warmup 5 2>>$OUT.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$spec]', '$cfg', d['config']['walkers_per_gpu'], round(d['value']), 'us/step', round(1e3*d['ms_per_step'],1), 'tile_us', round(1e3*r['avg_kernel_ms'],1), 'frac', round(r['frac'],4))" >> $OUT
  done
done
done
cat $OUT
