"""GPU box: device-resident stretch sampler, ensemble steps/s by walker count, overlapped half-steps on / off; chains compared."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rbvfit_amd.workloads import make_workload

for W in [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024]:
    wl = make_workload("C1", walkers=W)
    out = {}
    for ovl in (0, -1, 2):              # 2: overlapped, separate row / lnprob / version arrays instead of mailbox lines
        wl.engine.set_option("stretch_overlap", -1 if ovl == 2 else ovl)
        wl.engine.set_option("stretch_mailbox", 0 if ovl == 2 else 1)
        wl.engine.stretch_run(wl.thetas, 50, seed=1, store_chain=False)
        nst = 600
        t0 = time.perf_counter()
        r = wl.engine.stretch_run(wl.thetas, nst, seed=1, store_chain=False)
        dt = time.perf_counter() - t0
        rc = wl.engine.stretch_run(wl.thetas, 40, seed=3, store_chain=True)
        out[ovl] = (nst / dt, r[0], r[1], rc[2], rc[3], r[4])
    same = all(np.array_equal(out[0][k], out[-1][k]) and np.array_equal(out[0][k], out[2][k]) for k in (1, 2, 3, 4, 5))
    print(f"C1 W={W}: {out[0][0]:.0f} steps/s one stream, {out[2][0]:.0f} overlapped with separate arrays, {out[-1][0]:.0f} overlapped with mailbox lines "
          f"({1e6 / out[0][0] / 2:.2f} -> {1e6 / out[2][0] / 2:.2f} -> {1e6 / out[-1][0] / 2:.2f} us per half-step); "
          f"chains, positions, lnprob, acceptance counts identical: {same}", flush=True)
    wl.engine.close()
