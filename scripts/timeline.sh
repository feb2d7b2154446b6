#!/bin/bash
# kernel timeline (start/end ns) of a few bench steps
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.getcwd()+"/gpurun_out/timeline/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "vp::" in r["Kernel_Name"]]
t0 = int(rows[60]["Start_Timestamp"])
prev_end = None
for r in rows[60:75]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end else 0
    print(f"{r['Kernel_Name'][:40]:40s} start={(s-t0)/1e3:9.2f}us dur={(e-s)/1e3:7.2f}us gap_before={gap/1e3:6.2f}us")
    prev_end = e
PY
