#!/bin/bash
# GPU box: walker_kernel (one launch per batch; plain and with cooperative line cores) vs prep + tile + finalize launches on C1.
# Usage: scripts/exp_walker.sh <out-file> ; each line: label W evals/s us/step kernel_us prep_us fin_us
OUT=${1:-gpurun_out/exp_walker.txt}
: > $OUT
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for w in "$@"; do
    env "${envs[@]}" python bench.py --no-cpu-baseline --no-extras --walkers $w --steps 400 2>>gpurun_out/exp_walker.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', d['config']['walkers_per_gpu'], round(d['value']), round(1e3*d['ms_per_step'],2), round(1e3*r['avg_kernel_ms'],2), round(1e3*r['prep_ms'],2), round(1e3*r['finalize_ms'],2))" >> $OUT
    tail -1 $OUT
  done
}
WS="${WS:-64 128 256 512 1024 2048 8192}"
run launches RBVFIT_AMD_WALKER=0 -- $WS
run walker RBVFIT_AMD_WALKER=1 -- $WS

