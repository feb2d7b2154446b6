#!/bin/bash
# GPU box: A/B of library builds on C1 walker counts + C2..C4.  Usage: scripts/exp_ab.sh out.txt label=lib ...
OUT=$1; shift
: > $OUT
for round in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  for w in ${WS:-256 512 8192}; do
    env RBVFIT_AMD_LIB=$lib python bench.py --no-cpu-baseline --no-extras --walkers $w --steps 400 2>>$OUT.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', 'C1', d['config']['walkers_per_gpu'], round(d['value']), 'us/step', round(1e3*d['ms_per_step'],2), 'tile', round(1e3*r['avg_kernel_ms'],2), 'prep', round(1e3*r['prep_ms'],2), 'fin', round(1e3*r['finalize_ms'],2))" >> $OUT
    tail -1 $OUT
  done
done
done
CFGS="${CFGS:-C2 C3 C4}" scripts/exp_cfg.sh $OUT.cfg "$@"
cat $OUT.cfg >> $OUT
