"""Soak: long runs of the device-resident slice sampler (the loop on the GPU, several segments, chain kept and not kept);
every run is compared with the same run cut into small segments, and the final state with a fresh lnprob evaluation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rbvfit_amd.workloads import make_workload
for cfg, W, px, nsteps, keep in (("C1", 64, 4096, 4000, True), ("C1", 512, 4096, 1500, False), ("C3", 64, 2048, 600, True), ("C2", 128, 4096, 400, False),
                                  ("C1", 2600, 4096, 150, False)):
    wl = make_workload(cfg, walkers=W, pixels=px)
    eng = wl.engine
    t0 = time.perf_counter()
    a = eng.slice_run(wl.thetas, nsteps, seed=3, store_chain=keep)
    dt = time.perf_counter() - t0
    eng.set_option("slice_seg", 97)
    b = eng.slice_run(wl.thetas, nsteps, seed=3, store_chain=keep)
    eng.set_option("slice_seg", 0)
    same = all(np.array_equal(a[k], b[k]) for k in ("pos", "lnprob", "mu_history")) and a["n_evals"] == b["n_evals"]
    if keep:
        same = same and np.array_equal(a["chain"], b["chain"]) and np.array_equal(a["chain_lnprob"], b["chain_lnprob"])
    half = W // 2
    ref = eng.lnprob(a["pos"])
    ok = np.all(np.isfinite(a["lnprob"])) and np.all(a["pos"] >= wl.lb) and np.all(a["pos"] <= wl.ub)
    rel = np.max(np.abs(a["lnprob"] / ref - 1))
    print(f"{cfg} W={W} P={px}: {nsteps} iterations in {dt:.2f} s ({nsteps/dt:.0f} steps/s), {a['n_evals'] / (nsteps * W):.2f} evals per walker-step, "
          f"mu {a['mu']:.3f}; segments of 97 give the same run: {same}; state in bounds and finite: {ok}; lnprob vs a fresh batch: {rel:.1e}", flush=True)
    eng.close()
