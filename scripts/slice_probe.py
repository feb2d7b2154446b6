"""GPU box: ensemble steps per second of the device-resident slice sampler (vp_slice_run) on C1, after a burn-in that tunes mu."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rbvfit_amd.workloads import make_workload
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wl = make_workload("C1", walkers=W)
r0 = wl.engine.slice_run(wl.thetas, 20, seed=1, store_chain=False)
n = 100
t0 = time.perf_counter()
r1 = wl.engine.slice_run(r0["pos"], n, lnprob=r0["lnprob"], seed=1, step0=20, mu=r0["mu"], tune=r0["tune_state"], store_chain=False)
dt = time.perf_counter() - t0
print(f"{os.environ.get('RBVFIT_AMD_LIB', 'default')[-12:]:>12s} slice_rows={os.environ.get('RBVFIT_AMD_SLICE_ROWS', '2')} W={W}: {n / dt:8.1f} steps/s, "
      f"{r1['n_evals'] / (n * W):.2f} evals per walker-step, {r1['n_evals'] / dt / 1e6:.2f} M evals/s")
