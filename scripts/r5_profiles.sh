#!/bin/bash
# GPU box, round 5: the profile set kept under profiles/ -- per config rocprofv3 kernel statistics + PMC passes (profile_config.sh),
# then (second pass, so that bench.py finds the fresh pmc.json) the two bench lines per config, the model_flux entry, the default line.
# Usage: scripts/r4_profiles.sh <stage>   stage 1: profile passes; stage 2: bench lines (after profiles/r05_C*_pmc.json are in place)
STAGE=${1:-1}
if [ "$STAGE" = "1" ]; then
  for spec in "C1 400" "C2 60" "C3 40 --walkers 2048" "C4 30 --walkers 512"; do
    read -r -a a <<< "$spec"; cfg=${a[0]}; st=${a[1]}
    scripts/profile_config.sh $cfg $cfg $st "${a[@]:2}" > /dev/null 2>&1
    head -7 gpurun_out/prof_$cfg/summary.txt | cut -c1-180
  done
  # the model_flux entry under rocprofv3 (kernel statistics only)
  export TMPDIR=/tmp
  for cfg in C1 C2; do
    O=$PWD/gpurun_out/prof_mf_$cfg; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --entry model_flux --config $cfg --no-cpu-baseline --steps 100 --warmup 5 --repeats 3 > $O/bench_trace.json 2> $O/trace.err
    f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv; rm -rf $O/trace
    head -6 $O/kernel_stats.csv | cut -c1-160
  done
else
  O=gpurun_out/r5_lines; mkdir -p $O
  for spec in "C1" "C2" "C3 --walkers 2048" "C4 --walkers 512"; do
    read -r -a a <<< "$spec"; cfg=${a[0]}
    st=400; [ "$cfg" = "C2" ] && st=60; [ "$cfg" = "C3" ] && st=40; [ "$cfg" = "C4" ] && st=30
    python3 bench.py --config "${a[@]}" --no-cpu-baseline --no-extras --steps $st --warmup 5 > $O/${cfg}_bench.json 2> $O/${cfg}_bench.err
    python3 bench.py --config $cfg --gpus 1 --steps 20 --warmup 5 > $O/${cfg}_bench_driverform.json 2> $O/${cfg}_bench_driverform.err
    python3 - <<PY
import json
for f in ("$O/${cfg}_bench.json", "$O/${cfg}_bench_driverform.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print("$cfg", d["config"]["walkers_per_gpu"], round(d["value"]), "us/step", round(1e3 * d["ms_per_step"], 2), "kernel us", round(1e3 * r["avg_kernel_ms"], 2), "frac", round(r["frac"], 4), "valu_issue", round(r["valu_issue"]["frac"], 3) if r.get("valu_issue") else None, "host_entry", d.get("value_host_entry"), "stretch", d.get("mcmc_steps_per_sec"), "slice", (d.get("slice_sampler") or {}).get("steps_per_sec"))
PY
  done
  for cfg in C1 C2; do
    python3 bench.py --entry model_flux --config $cfg --steps 20 --warmup 5 > $O/model_flux_$cfg.json 2> $O/model_flux_$cfg.err
    python3 -c "
import json; d=json.loads(open('$O/model_flux_$cfg.json').read().strip().splitlines()[-1]); r=d['roofline']
print('model_flux $cfg', round(d['value']), 'us/step', round(1e3*d['ms_per_step'],2), 'frac', round(r['frac'],4), r['ms_per_step_by_flux_farfield'], 'host', round(d['value_host_entry']), 'cpu', round(d['cpu_baseline']['value']))"
  done
  ( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
fi
