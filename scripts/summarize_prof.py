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
