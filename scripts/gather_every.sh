#!/bin/bash
# pipelined-gather overhead vs chunk length (one rank): per step or per collective?
for e in "$@"; do
  BENCH_GATHER_EVERY=$e BENCH_FORCE_DIST=1 timeout -k 5 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gather']
print('every', $e, 'pipelined', round(d['ms_per_step']*1e3,2), 'blocking', round(g['ms_per_step_blocking_gather']*1e3,2), 'none', round(g['ms_per_step_without_gather']*1e3,2))"
done
