#!/bin/bash
# GPU box: the two bench lines per config that profiles/ keeps next to the rocprofv3 / PMC summaries (which this run reads for
# roofline.valu_issue): the profiling command's own form and the driver's.  Usage: scripts/bench_lines.sh <tag>
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
for spec in "C1" "C2" "C3 --walkers 2048" "C4 --walkers 512"; do
  read -r -a a <<< "$spec"; cfg=${a[0]}
  st=400; [ "$cfg" = "C2" ] && st=60; [ "$cfg" = "C3" ] && st=40; [ "$cfg" = "C4" ] && st=30
  python3 bench.py --config "${a[@]}" --no-cpu-baseline --no-extras --steps $st --warmup 5 --repeats 3 > $OUT/${cfg}_bench.json 2> $OUT/${cfg}_bench.err
  python3 bench.py --config $cfg --gpus 1 --steps 20 --warmup 5 > $OUT/${cfg}_bench_driverform.json 2> $OUT/${cfg}_bench_driverform.err
  python3 - <<PY
import json
for f in ("$OUT/${cfg}_bench.json", "$OUT/${cfg}_bench_driverform.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print("$cfg", d["config"]["walkers_per_gpu"], round(d["value"]), "us/step", round(1e3 * d["ms_per_step"], 1), "kernel us", round(1e3 * r["avg_kernel_ms"], 1), "frac", round(r["frac"], 4), "valu_issue", round(r["valu_issue"]["frac"], 3) if r.get("valu_issue") else None, d.get("mcmc_steps_per_sec"), (d.get("slice_sampler") or {}).get("steps_per_sec"))
PY
done
