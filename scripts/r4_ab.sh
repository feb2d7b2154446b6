#!/bin/bash
# GPU box, round 4: A/B of library builds.  Usage: scripts/r4_ab.sh <tag> label=lib ...   (lib paths relative to rbvfit_amd/lib)
# env: WS="256 512" (C1 walker counts), CFGS="C2 C3 C4" (or "" for none), ROUNDS=2, TESTS=1 to run the GPU tests first with the LAST lib
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
: > $O/ab.txt
line() {   # label lib config-args...
  label=$1; lib=$2; shift 2
  env RBVFIT_AMD_LIB=$lib python bench.py --no-cpu-baseline --no-extras --min-seconds ${MINS:-0.4} --steps ${STEPS:-200} "$@" 2>>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', d['config']['workload'][:2], d['config']['walkers_per_gpu'], round(d['value']), 'us/step', round(1e3*d['ms_per_step'],2), 'kernel', round(1e3*r['avg_kernel_ms'],2), 'prep', round(1e3*r['prep_ms'],2), 'fin', round(1e3*r['finalize_ms'],2))" >> $O/ab.txt
  tail -1 $O/ab.txt
}
if [ "${TESTS:-0}" = "1" ]; then
  last="${@: -1}"; RBVFIT_AMD_LIB=$PWD/rbvfit_amd/lib/${last#*=} python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
fi
for round in $(seq 1 ${ROUNDS:-2}); do
  for spec in "$@"; do
    label=${spec%%=*}; lib=$PWD/rbvfit_amd/lib/${spec#*=}
    for w in ${WS-256 512}; do line $label $lib --walkers $w; done
    for cfg in ${CFGS-}; do
      extra=""; [ "$cfg" = "C3" ] && extra="--walkers ${C3W:-2048}"; [ "$cfg" = "C4" ] && extra="--walkers ${C4W:-512}"
      line $label $lib --config $cfg $extra --steps 30
    done
  done
done
