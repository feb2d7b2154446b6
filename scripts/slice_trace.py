"""Reads the rocprofv3 database of a slice-sampler run (rocprofv3 --kernel-trace -d DIR -o NAME -- python3 scripts/slice_probe.py):
kernel statistics and the launch sequence of the last iterations.  Usage: slice_trace.py <results.db> [rows]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print("kernel, launches, mean / min / max us, total ms")
for r in c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by 6 desc"):
    print(f"  {r[0][:60]:60s} {r[1]:6d} {r[2]/1e3:7.2f} {r[3]/1e3:7.2f} {r[4]/1e3:7.2f} {r[5]/1e6:8.2f}")
rows = c.execute("select name,start,end from kernels order by start").fetchall()
prev = None
print("last launches: kernel, duration us, gap to the previous one us")
for nm, s, e in rows[-nshow:]:
    print(f"  {nm[:44]:44s} {(e-s)/1e3:7.2f} {0.0 if prev is None else (s-prev)/1e3:7.2f}")
    prev = e
