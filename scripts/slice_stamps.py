"""GPU box, diagnostic build (-DVP_STAMPS): where the slice sampler's round kernel spends its time (thread 0's clock at its phases).
    hipcc <flags of __graft_entry__> -DVP_STAMPS -o rbvfit_amd/lib/exp/lib_stamps.so rbvfit_amd/csrc/capi.hip
    RBVFIT_AMD_LIB=$PWD/rbvfit_amd/lib/exp/lib_stamps.so python scripts/slice_stamps.py [walkers]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rbvfit_amd import _lib
from rbvfit_amd.workloads import make_workload
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wl = make_workload("C1", walkers=W)
r0 = wl.engine.slice_run(wl.thetas, 20, seed=1, store_chain=False)
r1 = wl.engine.slice_run(r0["pos"], 20, lnprob=r0["lnprob"], seed=1, step0=20, mu=r0["mu"], tune=r0["tune_state"], store_chain=False)
lib = _lib.load()
lib.vp_debug_read_slice_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int]
buf = np.zeros(256 * 16, dtype=np.int64)
assert lib.vp_debug_read_slice_stamps(buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size) == 0
st = buf.reshape(256, 16)
names = ["entry", "state loaded", "results loaded", "consumed+scanned", "advance", "init", "ranks in LDS", "draws", "bookkeeping+store", "barrier", "rows", "end"]
print("round: cumulative ticks at each stage (100 MHz? shader clock?) ; stages:", names)
for r in range(1, 40):
    row = st[r, :12]
    if row[0] == 0: continue
    print(r, " ".join(f"{int(x - row[0]):6d}" for x in row))
d = st[1:200, :12]; d = d[d[:, 0] != 0]
print("mean", " ".join(f"{x:8.1f}" for x in (d - d[:, :1]).mean(axis=0)))
