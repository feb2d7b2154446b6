#!/bin/bash
# GPU box: bench lines of C1..C4 for a list of library builds.  Usage: scripts/exp_cfg.sh out.txt label=libpath ...
OUT=$1; shift
: > $OUT
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  for cfg in ${CFGS:-C2 C3 C4}; do
    extra=""
    [ "$cfg" = "C3" ] && extra="--walkers ${C3W:-2048}"
    [ "$cfg" = "C4" ] && extra="--walkers ${C4W:-512}"
    env RBVFIT_AMD_LIB=$lib python bench.py --no-cpu-baseline --no-extras --config $cfg $extra --steps ${STEPS:-30} --warmup 5 2>>$OUT.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', '$cfg', d['config']['walkers_per_gpu'], round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'tile_ms', round(r['avg_kernel_ms'],4), 'prep_ms', round(r['prep_ms'],4), 'frac', round(r['frac'],4))" >> $OUT
    tail -1 $OUT
  done
done
