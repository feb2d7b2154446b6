"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace", "*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        print(f"{r.get('Name','')[:70]:70s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} "
              f"min={r.get('MinNs')} max={r.get('MaxNs')} pct={r.get('Percentage')}")

for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    print(f"== {sub} (per-dispatch mean by kernel) ==")
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")[:60]
                acc[k][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
                meta[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    for k, cs in acc.items():
        print(f"{k}  vgpr/sgpr/lds/grid/wg={meta[k]}")
        for c, v in cs.items():
            print(f"    {c}: mean={sum(v)/len(v):.6g} n={len(v)}")

# machine-readable traffic figure for bench.py (tile kernel, lnprob variant)
import json
def mean_counter(sub, counter, kernel_sub="tile_kernel<0, 0"):
    vals = []
    for f in find(sub, "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel_sub in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                    vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None
fetch_kb, write_kb = mean_counter("pmc_fetch", "FETCH_SIZE"), mean_counter("pmc_write", "WRITE_SIZE")
if fetch_kb is not None and write_kb is not None:
    meta = {}
    try:
        meta = json.loads(open(os.path.join(out, "bench_trace.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    pm = {"config": "C1", "walkers_per_gpu": meta.get("config", {}).get("walkers_per_gpu"),
          "FETCH_SIZE_KB_mean": fetch_kb, "WRITE_SIZE_KB_mean": write_kb,
          "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B; 8-B/lane loads uncalibrated), WRITE_SIZE x1",
          "tile_kernel_hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024.0}
    json.dump(pm, open(os.path.join(out, "pmc.json"), "w"), indent=1)
    print("pmc.json:", pm)
