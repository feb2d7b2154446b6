#!/bin/bash
# GPU box: A/B of library builds on the model_flux entry.  Usage: mf_ab.sh label=lib ...
for round in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%=*}; lib=$PWD/rbvfit_amd/lib/${spec#*=}
    for cfg in C1 C2; do
      env RBVFIT_AMD_LIB=$lib python bench.py --entry model_flux --config $cfg --no-cpu-baseline --steps 100 --warmup 5 --repeats 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label $cfg', round(d['value']), 'us/step', round(1e3*d['ms_per_step'],2))"
    done
  done
done
