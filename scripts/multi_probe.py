"""GPU box: cost of the sharded device-resident stretch sampler (vp_multi_stretch_run) against vp_stretch_run on ONE GPU.
The device is listed 1, 2 and 3 times; half-steps ordered by events between the contexts' streams (multi_sync = 0) or by
flags polled inside the kernels (1).  Usage: python scripts/multi_probe.py [config] [walkers] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import rbvfit_amd
from rbvfit_amd.workloads import make_workload

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 300
wl = make_workload(cfg, walkers=W)
wl.engine.stretch_run(wl.thetas, 20, seed=1, store_chain=False)
t0 = time.perf_counter(); ref = wl.engine.stretch_run(wl.thetas, nst, seed=1, store_chain=False); dt = time.perf_counter() - t0
print(f"{cfg} W={W}: vp_stretch_run           {nst / dt:9.1f} steps/s  {0.5e6 * dt / nst:7.1f} us per half-step")
for ids in ([0], [0, 0], [0, 0, 0]):
    for sync in (0, 1):
        with rbvfit_amd.MultiEngine(ids) as m:
            m.set_bounds(wl.lb, wl.ub)
            for data, (wave, flux, err) in zip(wl.tables, wl.spectra):
                w = 1.0 / err ** 2
                m.add_instrument(wave, flux, w, np.log(w), **data.engine_kwargs())
            m.set_option("multi_sync", sync)
            m.stretch_run(wl.thetas, 20, seed=1, store_chain=False)
            t0 = time.perf_counter(); got = m.stretch_run(wl.thetas, nst, seed=1, store_chain=False); dt = time.perf_counter() - t0
            same = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
            print(f"  {len(ids)} context(s), {'flags ' if sync else 'events'}: {nst / dt:9.1f} steps/s  {0.5e6 * dt / nst:7.1f} us per half-step   "
                  f"identical to vp_stretch_run: {same}")
