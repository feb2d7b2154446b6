"""GPU box: the host-buffer entry (vp_lnprob_batch) with and without pre-armed launches (option "prearm").

    python scripts/prearm_probe.py [walkers ...]

us per call by walker count (median and mean over back-to-back calls), the counters, and what a caller that pauses between its
calls gets (busy wait of the given length between return and the next call: the launch waits `prearm_us` = 1000 us, and is left behind when calls come within half of that)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rbvfit_amd.workloads import make_workload       # noqa: E402


def run(eng, th, n, gap_us=0.0):
    wall = np.zeros(n)
    for i in range(n):
        if gap_us:
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < gap_us * 1e-6:
                pass
        t0 = time.perf_counter()
        eng.lnprob(th)
        wall[i] = time.perf_counter() - t0
    return 1e6 * np.median(wall), 1e6 * wall.mean()


def main():
    cfg = os.environ.get("PROBE_CONFIG", "C1")
    for W in [int(a) for a in sys.argv[1:]] or [64, 256, 512]:
        ref = None
        for mode in (0, 1, -1):
            wl = make_workload(cfg, walkers=W)
            eng = wl.engine
            eng.set_option("prearm", mode)
            th = np.ascontiguousarray(wl.thetas)
            for _ in range(300):
                out = eng.lnprob(th)
            if ref is None:
                ref = out
            same = bool(np.array_equal(out, ref, equal_nan=True))
            med, mean = run(eng, th, 3000)
            line = f"{cfg} W={W} prearm={mode:2d}: {med:.2f} us/call median, {mean:.2f} mean ({W / mean:.2f} M evals/s) same_bits={same} {eng.prearm_counts}"
            if mode != 0:
                gaps = []
                for gap in (20, 100, 400, 800):
                    m2, _ = run(eng, th, 300, gap)
                    gaps.append(f"gap {gap} us: {m2:.2f}")
                line += " | " + ", ".join(gaps) + f" {eng.prearm_counts}"
            else:
                m2, _ = run(eng, th, 300, 100)
                line += f" | gap 100 us: {m2:.2f}"
            print(line, flush=True)
            eng.close()


if __name__ == "__main__":
    main()
