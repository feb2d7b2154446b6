"""Soak: long device-sampler runs at small ensemble sizes (last-tile final reduction, fused accept/propose)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rbvfit_amd.workloads import make_workload
for cfg, W, px, nsteps in (("C1", 64, 4096, 20000), ("C1", 512, 4096, 5000), ("C3", 64, 2048, 3000), ("C2", 128, 4096, 2000)):
    wl = make_workload(cfg, walkers=W, pixels=px)
    eng = wl.engine
    t0 = time.perf_counter()
    pos, lp, chain, clp, nacc = eng.stretch_run(wl.thetas, nsteps, seed=9, store_chain=False)
    dt = time.perf_counter() - t0
    half = W // 2            # the sampler evaluates half-ensembles: compare batch for batch (the tile geometry, hence the
    ref = np.concatenate([eng.lnprob(pos[:half]), eng.lnprob(pos[half:])])   # grouping of the chi^2 partial sums, depends on the batch size)
    ok = np.array_equal(lp, ref) and np.all(np.isfinite(lp)) and np.all(pos >= wl.lb) and np.all(pos <= wl.ub)
    rel = np.max(np.abs(lp / eng.lnprob(pos) - 1))
    print(f"{cfg} W={W} P={px}: {nsteps} steps in {dt:.2f} s ({nsteps/dt:.0f} steps/s), acceptance {nacc.mean()/nsteps:.3f}, state consistent {ok}, max rel vs one 2x-size batch {rel:.1e}", flush=True)
# overlapped half-steps with mailbox lines against the one-stream form: long chains, bit for bit
for W, nsteps in ((64, 30000), (256, 20000), (512, 20000)):
    wl = make_workload("C1", walkers=W)
    res = {}
    for name, ovl, mail in (("one stream", 0, 1), ("overlapped, mailbox lines", -1, 1), ("overlapped, separate arrays", -1, 0)):
        wl.engine.set_option("stretch_overlap", ovl)
        wl.engine.set_option("stretch_mailbox", mail)
        t0 = time.perf_counter()
        res[name] = wl.engine.stretch_run(wl.thetas, nsteps, seed=21, store_chain=False) + (time.perf_counter() - t0,)
    a = res["one stream"]
    same = all(np.array_equal(a[k], r[k]) for r in res.values() for k in (0, 1, 4))
    print(f"C1 W={W}: {nsteps} steps, " + ", ".join(f"{n}: {nsteps / r[-1]:.0f} steps/s" for n, r in res.items()) +
          f"; final positions, lnprob, acceptance counts identical: {same}", flush=True)
    assert same
    wl.engine.close()
