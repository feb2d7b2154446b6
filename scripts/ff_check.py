"""GPU box: far-field expansions on / off: largest difference in lnprob and in timing, per config."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rbvfit_amd.workloads import make_workload

for cfg, W in (("C2", 1024), ("C3", 2048), ("C4", 512)):
    res = {}
    for ff in (0, 1):
        os.environ["RBVFIT_AMD_FARFIELD"] = str(ff)
        wl = make_workload(cfg, walkers=W)
        e = wl.engine
        lp = e.lnprob(wl.thetas)
        for _ in range(5): e.lnprob(wl.thetas)
        t0 = time.perf_counter(); n = 20
        for _ in range(n): e.lnprob(wl.thetas)
        dt = (time.perf_counter() - t0) / n
        res[ff] = (lp, dt)
        e.close()
    a, b = res[0][0], res[1][0]
    fin = np.isfinite(a)
    print(cfg, W, "max |dlnprob|", np.max(np.abs(a[fin] - b[fin])), "max rel", np.max(np.abs(a[fin] - b[fin]) / np.abs(a[fin])),
          "us/call off/on", round(res[0][1] * 1e6, 1), round(res[1][1] * 1e6, 1), flush=True)
