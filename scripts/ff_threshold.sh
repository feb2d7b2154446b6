#!/bin/bash
# GPU box: far-field expansions forced on (1) / off (0) / automatic (-1) on C2 by walker count: where the extra launch starts to pay.
OUT=${1:-gpurun_out/ff_threshold.txt}; : > $OUT
for w in 128 256 512 1024 2048; do for ff in 0 1 -1; do
  RBVFIT_AMD_FARFIELD=$ff python bench.py --config C2 --no-cpu-baseline --no-extras --walkers $w --steps 60 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('C2 walkers $w farfield $ff', round(1e3*d['ms_per_step'],1), 'us/step', r['kernel'])" >> $OUT
done; done
cat $OUT
