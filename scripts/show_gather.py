"""Prints the gather figures of a bench line (bench.py --gpus N, or BENCH_FORCE_DIST=1): step time, the exchange's cost."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d["gather"]
print(f"{d['value']:.0f} evals/s, {1e3 * d['ms_per_step']:.2f} us per pass; direct gather: {g['direct_gather']} "
      f"({g['direct_gather_unavailable_because']}); without exchange {1e3 * g['ms_per_step_without_gather']:.2f} us, "
      f"exchange costs {g['blocking_gather_cost_us']:.2f} us; with the blocking RCCL all_gather: "
      f"{g['ms_per_step_blocking_rccl_all_gather']}")
