"""GPU box: every launch structure over thousands of random permutations of the same walkers -- each row must come out with
the same bits every time (how the exp-table race of the 4-wave tile workgroups was pinned down: profiles/r02_notes.md)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rbvfit_amd.workloads import make_workload

rng = np.random.default_rng(7)
for cfgname, W, px, n in (("C1", 512, None, 3000), ("C1", 256, None, 3000), ("C1", 1024, None, 2000), ("C2", 1024, None, 2000),
                          ("C3", 2048, None, 3000), ("C3", 256, None, 2000), ("C3", 64, 512, 3000), ("C4", 512, None, 300),
                          ("C2", 128, 4096, 3000)):
    wl = make_workload(cfgname, walkers=W, pixels=px) if px else make_workload(cfgname, walkers=W)
    th = wl.thetas.copy()
    th[1, 0] = wl.lb[0] - 0.25
    e = wl.engine
    got = e.lnprob(th)
    bad = 0
    for it in range(n):
        perm = rng.permutation(W)
        if not np.array_equal(e.lnprob(th[perm]), got[perm]):
            bad += 1
    print(cfgname, W, px, e.last_launch_kind, "mismatching launches:", bad, "of", n, flush=True)
    e.close()
