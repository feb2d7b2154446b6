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
rows = [r for r in rows if "vp::" in r["Kernel_Name"] or os.environ.get("TL_ALL")]
N0 = int(os.environ.get("TL_FROM", "60")); N0 = N0 if N0 >= 0 else len(rows) + N0; N1 = N0 + int(os.environ.get("TL_COUNT", "15"))
t0 = int(rows[N0]["Start_Timestamp"])
prev_end = None
for r in rows[N0:N1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end else 0
    print(f"{r['Kernel_Name'][:40]:40s} start={(s-t0)/1e3:9.2f}us dur={(e-s)/1e3:7.2f}us gap_before={gap/1e3:6.2f}us")
    prev_end = e
PY
