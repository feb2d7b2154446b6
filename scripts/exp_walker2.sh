#!/bin/bash
# walker kernel A/B: production build (waves_per_eu 6, 28 B scratch) vs a scratch-free build (87 VGPRs, 5 waves/SIMD)
OUT=${1:-gpurun_out/exp_walker2.txt}
: > $OUT
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for w in "$@"; do
    env "${envs[@]}" python bench.py --no-cpu-baseline --no-extras --walkers $w --steps 400 2>>gpurun_out/exp_walker2.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', d['config']['walkers_per_gpu'], round(d['value']), round(1e3*d['ms_per_step'],2), round(1e3*r['avg_kernel_ms'],2), round(1e3*r['prep_ms'],2), round(1e3*r['finalize_ms'],2))" >> $OUT
    tail -1 $OUT
  done
}
WS="${WS:-64 256 512}"
run launches RBVFIT_AMD_WALKER=0 -- $WS
run walker_wpe6 RBVFIT_AMD_WALKER=1 -- $WS
run walker_wpe4 RBVFIT_AMD_WALKER=1 RBVFIT_AMD_LIB=$PWD/rbvfit_amd/lib/exp/lib_wpe4.so -- $WS
