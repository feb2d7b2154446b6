"""Stress: BASELINE 'Stress' config C4 at its full 4096 walkers on one GPU, and a very large C1 batch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rbvfit_amd.workloads import make_workload
from oracle import voigt_oracle as vo
def oracle_rows(wl, rows):
    insts = []
    for data, (wave, flux, err) in zip(wl.tables, wl.spectra):
        od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors, data.N_indices,
                                data.b_indices, data.v_indices, data.taps if data.taps is not None else np.zeros(0), data.lsf_mode, data.voigt_method)
        insts.append(vo.OracleInstrument.from_error(od, wave, flux, err))
    return vo.lnprob_batch(wl.thetas[rows], wl.lb, wl.ub, insts)
for name, W in (("C4", 4096), ("C1", 60000), ("C2", 8192)):
    wl = make_workload(name, walkers=W)
    t0 = time.perf_counter(); got = wl.engine.lnprob(wl.thetas); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); got2 = wl.engine.lnprob(wl.thetas); dt2 = time.perf_counter() - t0
    rows = np.array([0, W // 2, W - 1])
    ref = oracle_rows(wl, rows)
    print(f"{name} W={W}: first call {dt*1e3:.1f} ms, second {dt2*1e3:.1f} ms ({W/dt2/1e6:.2f} M evals/s), finite={np.isfinite(got).all()}, "
          f"repeatable={np.array_equal(got, got2)}, max rel vs oracle {np.max(np.abs(got[rows]/ref-1)):.2e}", flush=True)
    wl.engine.close()
