"""Diagnostic: per-workgroup time stamps of one tile-kernel launch (needs lib_stamp.so, -DVP_STAMP)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RBVFIT_AMD_LIB"] = os.path.join(ROOT, "rbvfit_amd", "lib", "ablate", os.environ.get("STAMP_LIB", "lib_stamp.so"))
from rbvfit_amd.workloads import make_workload
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wl = make_workload("C1", walkers=W)
eng = wl.engine
lib = eng._lib
ntiles = int(os.environ.get("NTILES", "12"))
n = W * ntiles
for _ in range(3):
    eng.lnprob(wl.thetas)
lib.vp_debug_stamps_alloc.argtypes = [C.c_void_p, C.c_int]
lib.vp_debug_stamps_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
lib.vp_debug_stamps_alloc(eng._ctx, n)
eng.lnprob(wl.thetas)
buf = np.zeros((n, 8), dtype=np.uint64)
lib.vp_debug_stamps_read(eng._ctx, n, buf.ctypes.data_as(C.c_void_p))
ok = buf[:, 0] > 0
st = buf[ok].astype(np.int64)
t0 = st[:, 0].min()
T = (st[:, :6] - t0) / 100.0          # us (100 MHz)
print("workgroups", ok.sum(), "of", n)
print("kernel span (first start -> last end): %.2f us" % T[:, 5].max())
print("start times: p0 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(T[:, 0], [0, 50, 90, 99, 100])))
print("end   times: p0 %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(T[:, 5], [0, 10, 50, 90, 100])))
d = T[:, 5] - T[:, 0]
print("wave lifetime: mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % (d.mean(), *np.percentile(d, [10, 50, 90, 100])))
for k, name in enumerate(["prologue", "phase A", "phase B", "exp/LSF", "reduce+ticket"]):
    seg = T[:, k + 1] - T[:, k]
    print("  %-14s mean %.2f p50 %.2f p90 %.2f max %.2f" % (name, seg.mean(), *np.percentile(seg, [50, 90, 100])))
# concurrency over time
edges = np.arange(0, T[:, 5].max() + 1, 1.0)
act = [(np.sum((T[:, 0] <= e) & (T[:, 5] > e))) for e in edges]
print("resident waves at t=0,1,2,..us:", act)
# by tile index (tile = idx // W)
tile = np.repeat(np.arange(ntiles), W)[ok]
for tt in range(ntiles):
    m = tile == tt
    print("tile %2d: start p50 %.1f  life p50 %.2f  phaseB p50 %.2f" % (tt, np.median(T[m, 0]), np.median(d[m]), np.median((T[m, 3] - T[m, 2]))))
