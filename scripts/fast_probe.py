"""GPU box: us per lnprob pass of C1's shape with voigt_method='fast' (the reference's Tepper-Garcia option), device-resident."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from rbvfit_amd.engine import Engine
from rbvfit_amd.model import FitConfiguration, VoigtModel
from rbvfit_amd.workloads import make_workload

for W in (256, 512, 2048):
    wl = make_workload("C1", walkers=W)
    cfg = FitConfiguration(); cfg.add_system(0.348, "MgII", [2796.35, 2803.53], 2)
    data = VoigtModel(cfg, FWHM="6.5", voigt_method="fast").compile().data
    wave, flux, err = wl.spectra[0]
    eng = Engine(0); eng.set_bounds(wl.lb, wl.ub)
    w = 1.0 / err ** 2
    eng.add_instrument(wave, flux, w, np.log(w), **data.engine_kwargs())
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    th = torch.from_numpy(wl.thetas).cuda(); out = torch.empty(W, dtype=torch.float64, device="cuda")
    for _ in range(200): eng.lnprob_device(th.data_ptr(), out.data_ptr(), W, s.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(400): eng.lnprob_device(th.data_ptr(), out.data_ptr(), W, s.cuda_stream)
    e1.record(s); torch.cuda.synchronize()
    print(f"{os.environ.get('RBVFIT_AMD_LIB', 'default')[-14:]:>14s} fast W={W}: {1e3 * e0.elapsed_time(e1) / 400:.2f} us per pass ({eng.last_launch_kind})")
    eng.close(); wl.engine.close()
