"""Condense rocprofv3 CSV output (kernel stats + PMC passes of scripts/profile_config.sh) into a short text
summary and a machine-readable pmc.json (HBM traffic of the dominant kernel per launch, for bench.py's
`roofline.traffic`).  Usage: summarize_prof.py <dir> <config>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "C1"


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


bench = {}
try:
    bench = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
    r = bench["roofline"]
    print(f"== bench line (un-profiled run): {bench['value']:.0f} evals/s, {1e3 * bench['ms_per_step']:.2f} us/step, "
          f"{bench['config']['walkers_per_gpu']} walkers; dominant kernel {1e3 * r['avg_kernel_ms']:.2f} us (HIP events), "
          f"roofline frac {r['frac']:.4f} ==")
except Exception as e:
    print("no bench line:", e)

print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
dominant = None
for f in find("trace", "*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:10]:
        print(f"{r.get('Name','')[:70]:70s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} "
              f"min={r.get('MinNs')} max={r.get('MaxNs')} pct={r.get('Percentage')}")
    vp = [r for r in rows if "vp::" in r.get("Name", "")]
    if vp and dominant is None:
        dominant = max(vp, key=lambda r: float(r.get("TotalDurationNs", r.get("Percentage", 0)) or 0))
dom_name = dominant["Name"] if dominant else "tile_kernel<0, 0, false>"
dom_key = dom_name.split("(")[0].replace("void ", "").strip()
print(f"== dominant kernel: {dom_key}, rocprofv3 mean {float(dominant['AverageNs']) / 1e3:.2f} us over {dominant['Calls']} calls ==" if dominant else "")

means = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")
                grid = r.get("Grid_Size")
                acc[(k, grid)][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
                meta[(k, grid)] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), grid, r.get("Workgroup_Size"))
    print(f"== {sub} (per-dispatch mean; vp:: kernels, by grid size) ==")
    for (k, grid), cs in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
        if "vp::" not in k:
            continue
        n = len(next(iter(cs.values())))
        if n < 3:
            continue
        print(f"{k[:72]}  vgpr/sgpr/lds/grid/wg={meta[(k, grid)]} n={n}")
        for c, v in cs.items():
            print(f"    {c}: {sum(v) / len(v):.6g}")
            if dom_key in k:
                means.setdefault(c, []).append((n, sum(v) / len(v)))

def best(c):
    v = means.get(c)
    return max(v)[1] if v else None     # the grid with the most dispatches = the benchmarked launches

fetch_kb, write_kb = best("FETCH_SIZE"), best("WRITE_SIZE")
pm = {"config": cfg, "walkers_per_gpu": bench.get("config", {}).get("walkers_per_gpu"), "kernel": dom_key,
      "rocprofv3_mean_kernel_us": float(dominant["AverageNs"]) / 1e3 if dominant else None,
      "FETCH_SIZE_KB_mean": fetch_kb, "WRITE_SIZE_KB_mean": write_kb,
      "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B; 8-B/lane loads uncalibrated), WRITE_SIZE x1",
      "tile_kernel_hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024.0 if fetch_kb is not None and write_kb is not None else None,
      "sq": {c: best(c) for c in means if c not in ("FETCH_SIZE", "WRITE_SIZE")}}
json.dump(pm, open(os.path.join(out, "pmc.json"), "w"), indent=1)
print("pmc.json:", json.dumps(pm))
