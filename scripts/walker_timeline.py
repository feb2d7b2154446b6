"""GPU box, diagnostic build only: where the waves of walker_kernel spend their time.

    hipcc ... -DVP_STAMPS -o rbvfit_amd/lib/exp/lib_stamps.so rbvfit_amd/csrc/capi.hip
    RBVFIT_AMD_LIB=$PWD/rbvfit_amd/lib/exp/lib_stamps.so python scripts/walker_timeline.py 256 512

Every wave leaves the shader clock at: 0 kernel entry, 1 records ready (first barrier), 2 end of phase A (wings),
3 end of phase B (line cores), 4 end of LSF + chi^2, 5 behind the last barrier, 6 Dawson table staged (tiles with line cores).  Printed per walker count: the span of
each phase by tile (mean over walkers, us at a 2.4 GHz shader clock), when the phases end relative to the workgroup's first
stamp, and the whole launch (last stamp - first stamp over all workgroups)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rbvfit_amd import _lib as L                     # noqa: E402
from rbvfit_amd.workloads import make_workload       # noqa: E402

NW, NWAVES, NST = 1024, 16, 8


def main():
    lib = L.load()
    lib.vp_debug_read_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    lib.vp_debug_read_stamps.restype = C.c_int
    for W in [int(a) for a in sys.argv[1:]] or [256, 512]:
        wl = make_workload("C1", walkers=W)
        eng = wl.engine
        eng.set_option("walker", 1)
        for _ in range(20):
            eng.lnprob(wl.thetas)
        assert eng.last_launch_kind == "walker"
        buf = np.zeros(NW * NWAVES * NST, dtype=np.int64)
        assert lib.vp_debug_read_stamps(buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size) == 0
        st = buf.reshape(NW, NWAVES, NST)[:W, :12, :].astype(np.float64)
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            eng.lnprob(wl.thetas)
        host_us = (time.perf_counter() - t0) / n * 1e6
        span_ticks = st[:, :, 5].max() - st[:, :, 0].min()
        print(f"\n=== C1, {W} walkers: first entry -> last exit {span_ticks:.0f} ticks; host round trip {host_us:.1f} us per call")
        tick_us = 1.0 / 2400.0                    # the shader clock (2.4 GHz when the launch runs at full clock)
        print(f"    (2.4 GHz shader clock assumed; stamps of different XCDs do not share a base: launch span {span_ticks * tick_us:.1f} us)")
        wg0 = st[:, :, 0].min(axis=1, keepdims=True)         # first stamp of each workgroup
        rel = (st - wg0[:, :, None]) * tick_us
        rel[rel < 0] = np.nan                      # stages a wave did not pass in this launch keep an older launch's stamp
        names = ["entry", "records", "phaseA", "phaseB", "LSF", "final", "dawson", "-"]
        print("    end of stage relative to the workgroup's first wave entry, us (mean over walkers), by tile:")
        print("    tile " + " ".join(f"{n:>8s}" for n in names))
        for t in range(12):
            print(f"    {t:4d} " + " ".join(f"{np.nanmean(rel[:, t, k]):8.2f}" for k in range(8)))
        print("    all  " + " ".join(f"{np.nanmean(rel[:, :, k]):8.2f}" for k in range(8)))
        print("    p90  " + " ".join(f"{np.nanpercentile(rel[:, :, k], 90):8.2f}" for k in range(8)))
        print("    max  " + " ".join(f"{np.nanmax(rel[:, :, k]):8.2f}" for k in range(8)))
        start = (st[:, :, 0].min(axis=1) - st[:, :, 0].min()) * tick_us
        end = (st[:, :, 5].max(axis=1) - st[:, :, 0].min()) * tick_us
        print(f"    workgroup start after launch begin: mean {start.mean():.2f} p90 {np.percentile(start, 90):.2f} max {start.max():.2f} us;"
              f" workgroup end: mean {end.mean():.2f} p90 {np.percentile(end, 90):.2f} max {end.max():.2f} us")
        life = (st[:, :, 5].max(axis=1) - st[:, :, 0].min(axis=1)) * tick_us
        print(f"    workgroup life: mean {life.mean():.2f} p10 {np.percentile(life, 10):.2f} p90 {np.percentile(life, 90):.2f} max {life.max():.2f} us")
        wl.engine.close()


if __name__ == "__main__":
    main()
