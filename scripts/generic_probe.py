"""GPU box: batches in which EVERY walker has a line outside the fast domain (a fit with damped lines): the generic tile launch
does all the work.  us per lnprob batch and per model_flux batch (device-resident), C1 geometry, 512 rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rbvfit_amd.workloads import make_workload

wl = make_workload("C1", walkers=512)
eng = wl.engine
lb, ub = np.array(wl.lb, copy=True), np.array(wl.ub, copy=True)
C = lb.size // 3
lb[C:2 * C] = 0.0                       # b may reach 0: the instrument needs the generic path
eng.set_bounds(lb, ub)
th = np.array(wl.thetas, copy=True)
th[:, C] = 0.03                         # a ~ 0.25 for the first component: every walker is flagged
d_th = torch.tensor(th, device="cuda")
W, P = th.shape[0], eng.n_pixels[0]
d_out = torch.empty(W, dtype=torch.float64, device="cuda")
d_flux = torch.empty((W, P), dtype=torch.float64, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for name, fn in (("lnprob", lambda: eng.lnprob_device(d_th.data_ptr(), d_out.data_ptr(), W, st.cuda_stream)),
                     ("model_flux", lambda: eng.model_flux_device(0, d_th.data_ptr(), d_flux.data_ptr(), W, True, st.cuda_stream))):
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 200)
        print(f"all rows flagged, C1 512: {name} {1e6 * float(np.median(ts)):.1f} us per batch ({eng.last_launch_kind})", flush=True)
print("finite lnprob rows:", int(torch.isfinite(d_out).sum()))
