#!/bin/bash
# walker-count sweep of bench.py (GPU box): W, evals/s, ms/step, tile/prep/finalize ms
for w in "$@"; do python bench.py --no-cpu-baseline --walkers $w --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['config']['walkers_per_gpu'], round(d['value']), round(d['ms_per_step'],5), round(r['avg_kernel_ms'],5), round(r['prep_ms'],5), 'frac=%.3f'%r['frac'])"; done
