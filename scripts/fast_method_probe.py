"""Throughput of voigt_method='fast' (Tepper-Garcia, bug-compatible) vs 'wofz' on the C1 geometry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rbvfit_amd
from rbvfit_amd.model import FitConfiguration, VoigtModel
from rbvfit_amd.workloads import make_workload
wl = make_workload("C1", walkers=512)
wave, flux, err = wl.spectra[0]
for method in ("wofz", "fast"):
    cfg = FitConfiguration(); cfg.add_system(0.348, "MgII", [2796.35, 2803.53], 2)
    data = VoigtModel(cfg, FWHM="6.5", voigt_method=method).compile().data
    with rbvfit_amd.Engine(0) as e:
        e.set_bounds(wl.lb, wl.ub)
        e.add_instrument(wave, flux, 1 / err ** 2, np.log(1 / err ** 2), **data.engine_kwargs())
        for _ in range(300): e.lnprob(wl.thetas)
        t0 = time.perf_counter()
        for _ in range(300): e.lnprob(wl.thetas)
        dt = (time.perf_counter() - t0) / 300
        print(f"{method}: {1e6*dt:.1f} us per 512-walker host call ({512/dt/1e6:.2f} M evals/s)", flush=True)
