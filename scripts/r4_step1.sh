#!/bin/bash
# GPU box, round 4 step 1: tests, default bench line (host entry first-class), host_spin A/B, model_flux entry, walker timeline.
O=gpurun_out/r4_step1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; tail -c 600 $O/bench_driverform.json
for hs in 1 2; do
  RBVFIT_AMD_HOST_SPIN=$hs BENCH_HOST_ENTRY=1 python bench.py --no-cpu-baseline --no-extras --steps 200 --min-seconds 0.5 > $O/bench_hostspin$hs.json 2>> $O/err.txt
  python - <<PY
import json
d=json.loads(open("$O/bench_hostspin$hs.json").read().strip().splitlines()[-1])
print("host_spin=$hs", round(d["value"]), round(d["value_host_entry"]), d["host_entry_latency"])
PY
done
python bench.py --entry model_flux --steps 20 --warmup 5 > $O/bench_model_flux_C1.json 2>> $O/err.txt; tail -c 1500 $O/bench_model_flux_C1.json
python bench.py --entry model_flux --config C2 --steps 20 --warmup 5 > $O/bench_model_flux_C2.json 2>> $O/err.txt; tail -c 1500 $O/bench_model_flux_C2.json
RBVFIT_AMD_LIB=$PWD/rbvfit_amd/lib/exp/lib_stamps.so python scripts/walker_timeline.py 256 512 > $O/timeline.txt 2>&1
