"""GPU box: soak of the pre-armed launches of vp_lnprob_batch.

    python scripts/soak_prearm.py [seconds]

Random batches (three shapes, rows outside the box, NaN rows), random pauses between calls that straddle the waiting time
(`prearm_us` = 150 here), other entry points and a second context of the same GPU in between, a third context created and destroyed
now and then -- every result compared bit for bit with what a context with "prearm" = 0 returned for the same batch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rbvfit_amd.workloads import make_workload       # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(11)
    wl = make_workload("C1", walkers=512)
    ref, eng = wl.engine, make_workload("C1", walkers=512).engine
    other = make_workload("C1", walkers=64).engine
    ref.set_option("prearm", 0)
    eng.set_option("prearm_us", 150)
    other.set_option("prearm_us", 150)
    # (round 5: by default a launch waits only as long as the caller's rhythm suggests and a caller whose launches expire is left
    #  alone for a while -- with this script's random pauses almost nothing would be pre-armed: "prearm" = 1 arms after every
    #  eligible call and waits the whole prearm_us, which is what a soak of the PROTOCOL wants)
    eng.set_option("prearm", 1)
    lb, ub = np.asarray(wl.lb), np.asarray(wl.ub)
    base = np.ascontiguousarray(wl.thetas)
    batches = []
    for k in range(24):
        W = (512, 256, 64)[k % 3]
        th = np.clip(base[:W] + 2e-3 * rng.standard_normal((W, base.shape[1])) * (ub - lb), lb, ub)
        if k % 5 == 1:
            th[rng.integers(W), 0] = lb[0] - 1.0
        if k % 7 == 2:
            th[rng.integers(W), 3] = np.nan
        batches.append(np.ascontiguousarray(th))
    want = [ref.lnprob(th) for th in batches]
    flux_want = ref.model_flux(0, batches[0][:2])
    t_end = time.perf_counter() + seconds
    n = bad = 0
    stats = dict(flux=0, other=0, created=0)
    k = 0
    while time.perf_counter() < t_end:
        r = rng.random()
        # mostly runs of one shape (so that launches are armed and used), now and then a jump
        k = (k + 3) % 24 if r < 0.85 else int(rng.integers(24))
        pause = (0.0, 20e-6, 60e-6, 120e-6, 150e-6, 170e-6, 400e-6)[int(rng.integers(7))]
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < pause:
            pass
        got = eng.lnprob(batches[k])
        n += 1
        if not np.array_equal(got, want[k], equal_nan=True):
            bad += 1
            print("MISMATCH at call", n, "batch", k, flush=True)
        if r > 0.97:
            assert np.array_equal(eng.model_flux(0, batches[0][:2]), flux_want)
            stats["flux"] += 1
        elif r > 0.93:
            o = other.lnprob(batches[2])
            assert np.array_equal(o, want[2], equal_nan=True)
            stats["other"] += 1
        elif r > 0.925:
            tmp = make_workload("C1", walkers=64).engine
            assert np.array_equal(tmp.lnprob(batches[2]), want[2], equal_nan=True)
            assert np.array_equal(tmp.lnprob(batches[2]), want[2], equal_nan=True)
            tmp.close()
            stats["created"] += 1
        if n % 20000 == 0:
            print(f"{n} calls, {bad} mismatches, {eng.prearm_counts}", flush=True)
    print(f"soak: {n} calls in {seconds:.0f} s, {bad} mismatches; counts {eng.prearm_counts}; other context {other.prearm_counts}; {stats}")
    assert bad == 0
    for e in (ref, eng, other):
        e.close()


if __name__ == "__main__":
    main()
