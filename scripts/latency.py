"""Per-call latency of the host-buffer entry (vp_lnprob_batch: H2D + kernels + D2H + sync) vs batch size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rbvfit_amd.workloads import make_workload
wl = make_workload(sys.argv[1] if len(sys.argv) > 1 else "C1", walkers=4096)
for W in (1, 8, 64, 256, 512, 1024, 4096):
    th = wl.thetas[:W]
    for _ in range(20): wl.engine.lnprob(th)
    t0 = time.perf_counter(); n = 300
    for _ in range(n): wl.engine.lnprob(th)
    dt = (time.perf_counter() - t0) / n
    print(f"W={W:5d}: {dt*1e6:8.1f} us/call  {W/dt/1e6:7.3f} M evals/s")
