#!/bin/bash
# GPU box: A/B of environment settings of ONE library on C1 by walker count.  Usage: scripts/r4_env_ab.sh <tag> "label:ENV=V ENV2=V" ...
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O; : > $O/ab.txt
for round in 1 2; do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    for w in ${WS-256 512}; do
      env $envs python bench.py --no-cpu-baseline --no-extras --min-seconds ${MINS:-0.4} --steps 200 --walkers $w ${BARGS:-} 2>>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', d['config']['workload'][:2], d['config']['walkers_per_gpu'], round(d['value']), 'us/step', round(1e3*d['ms_per_step'],2))" >> $O/ab.txt
      tail -1 $O/ab.txt
    done
  done
done
