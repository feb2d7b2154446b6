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

NW, NWAVES, NST = 1024, 16, 16


def main():
    lib = L.load()
    lib.vp_debug_read_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    lib.vp_debug_read_stamps.restype = C.c_int
    for W in [int(a) for a in sys.argv[1:]] or [256, 512]:
        PIX = int(os.environ.get("TIMELINE_PIXELS", "0")) or None      # (with RBVFIT_AMD_SPAN: other tile geometries)
        wl = make_workload("C1", walkers=W, pixels=PIX)
        NT = min(16, int(wl.engine._lib.vp_instrument_pixels(wl.engine._ctx, 0) and (os.environ.get("TIMELINE_TILES") or 12)))
        eng = wl.engine
        eng.set_option("walker", 1)
        eng.set_option("prearm", 0)          # (the ordinary launch: a pre-armed one spends its entry waiting for the host)
        for _ in range(20):
            eng.lnprob(wl.thetas)
        assert eng.last_launch_kind == "walker"
        cnt = np.zeros(8, dtype=np.uint32)
        if hasattr(lib, "vp_debug_read_counters"):
            lib.vp_debug_read_counters.argtypes = [C.c_void_p]
            lib.vp_debug_read_counters(cnt.ctypes.data)            # reset
            eng.lnprob(wl.thetas)
            lib.vp_debug_read_counters(cnt.ctypes.data)
            print(f"\n=== C1, {W} walkers, one launch: (chunk, line) pairs from helper slots {cnt[0]}, evaluated by the owner {cnt[1]}, "
                  f"polls that found a slot not ready {cnt[2]}, helper pair items {cnt[3]}  (per walker: {cnt[0] / W:.1f}, {cnt[1] / W:.1f}, {cnt[2] / W:.1f}, {cnt[3] / W:.1f})")
        buf = np.zeros(NW * NWAVES * NST, dtype=np.int64)
        assert lib.vp_debug_read_stamps(buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size) == 0
        G = int(os.environ.get("TIMELINE_SPLIT", "0"))      # split form: rows are workgroups (walker * G + group); print group by group
        if G:
            assert eng.last_walker_split == G, eng.last_walker_split
            allraw = buf.reshape(NW, NWAVES, NST)[:W * G, :NT, :]
            for sg in range(G):
                r = allraw[sg::G].astype(np.float64)
                rel = (r - r[:, :, 0].min(axis=1)[:, None, None]) / 2400.0
                rel[rel < 0] = np.nan
                print(f"\n=== C1, {W} walkers x {G} groups, group {sg}: end of stage after the workgroup's first entry, us (mean over walkers), by wave")
                print("    wave    entry  records   phaseA   phaseB      LSF    final  B-start        -  kernarg    theta  prepped  drained")
                for t in range(NT):
                    print(f"    {t:4d} " + " ".join(f"{np.nanmean(rel[:, t, k]):8.2f}" for k in range(12)))
                life = (r[:, :, 5].max(axis=1) - r[:, :, 0].min(axis=1)) / 2400.0
                print(f"    workgroup life: mean {np.nanmean(life):.2f} p90 {np.nanpercentile(life, 90):.2f} max {np.nanmax(life):.2f} us")
            wl.engine.close()
            continue
        raw = buf.reshape(NW, NWAVES, NST)[:W, :NT, :]
        hw = raw[:, :, 7]
        simd = (hw >> 4) & 3
        cu = ((hw >> 32) & 0xF) * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 0xF)     # (xcc, se, cu) as one key
        print(f"\n=== C1, {W} walkers: SIMD of wave t (share of walkers on SIMD 0..3), by tile:")
        for t in range(NT):
            print(f"    {t:4d} " + " ".join(f"{np.mean(simd[:, t] == k):6.2f}" for k in range(4)))
        rel_simd = (simd - simd[:, :1]) & 3
        print("    SIMD of wave t relative to wave 0 (mode, share): " + " ".join(
            f"{np.bincount(rel_simd[:, t], minlength=4).argmax()}:{np.bincount(rel_simd[:, t], minlength=4).max() / W:.2f}" for t in range(NT)))
        print(f"    workgroups whose waves all sit on one CU: {np.mean([len(set(cu[w])) == 1 for w in range(W)]):.2f};"
              f" distinct CUs used: {len(set(cu[:, 0]))}; max workgroups on one CU: {np.bincount(np.unique(cu[:, 0], return_inverse=True)[1]).max()}")
        st = raw.astype(np.float64)
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
        names = ["entry", "records", "phaseA", "phaseB", "LSF", "final", "B-start", "-", "kernarg", "theta", "prepped", "drained"]
        print("    end of stage relative to the workgroup's first wave entry, us (mean over walkers), by tile:")
        print("    tile " + " ".join(f"{n:>8s}" for n in names))
        for t in range(NT):
            print(f"    {t:4d} " + " ".join(f"{np.nanmean(rel[:, t, k]):8.2f}" for k in range(len(names))))
        print("    all  " + " ".join(f"{np.nanmean(rel[:, :, k]):8.2f}" for k in range(len(names))))
        print("    p90  " + " ".join(f"{np.nanpercentile(rel[:, :, k], 90):8.2f}" for k in range(len(names))))
        print("    max  " + " ".join(f"{np.nanmax(rel[:, :, k]):8.2f}" for k in range(len(names))))
        start = (st[:, :, 0].min(axis=1) - st[:, :, 0].min()) * tick_us
        end = (st[:, :, 5].max(axis=1) - st[:, :, 0].min()) * tick_us
        print(f"    workgroup start after launch begin: mean {start.mean():.2f} p90 {np.percentile(start, 90):.2f} max {start.max():.2f} us;"
              f" workgroup end: mean {end.mean():.2f} p90 {np.percentile(end, 90):.2f} max {end.max():.2f} us")
        life = (st[:, :, 5].max(axis=1) - st[:, :, 0].min(axis=1)) * tick_us
        print(f"    workgroup life: mean {life.mean():.2f} p10 {np.percentile(life, 10):.2f} p90 {np.percentile(life, 90):.2f} max {life.max():.2f} us")
        xcc = (hw[:, 0] >> 32) & 0xF
        print("    workgroup life by XCD: " + " ".join(f"{k}:{life[xcc == k].mean():.2f}" for k in sorted(set(xcc))))
        # the two workgroups of a CU: do their waves 0 sit on the same SIMD?  life of the later one by that
        keys, inv = np.unique(cu[:, 0], return_inverse=True)
        same, diff = [], []
        for k in range(len(keys)):
            idx = np.where(inv == k)[0]
            if len(idx) == 2:
                (same if simd[idx[0], 0] == simd[idx[1], 0] else diff).append(max(life[idx[0]], life[idx[1]]))
        slot = hw & 0xF
        pairs = [np.where(inv == k)[0] for k in range(len(keys)) if (inv == k).sum() == 2]
        if pairs:
            d = np.array([abs(int(a) - int(b)) for a, b in pairs])
            print(f"    the two workgroups of a CU: |index difference| values {sorted(set(d.tolist()))[:8]}; wave slots (HW_ID[3:0]) of the lower-index one "
                  f"{sorted(set(slot[[min(a, b) for a, b in pairs]].ravel().tolist()))}, of the higher-index one {sorted(set(slot[[max(a, b) for a, b in pairs]].ravel().tolist()))}")
            lo = np.array([life[min(a, b)] for a, b in pairs]); hi = np.array([life[max(a, b)] for a, b in pairs])
            print(f"    life of the lower-index workgroup of a CU {lo.mean():.2f} us, of the higher-index one {hi.mean():.2f} us")
        if same and diff:
            print(f"    CUs with two workgroups: first waves on the SAME SIMD {len(same)} CUs, slower workgroup's life {np.mean(same):.2f} us;"
                  f" on different SIMDs {len(diff)} CUs, {np.mean(diff):.2f} us")
        entry_skew = (st[:, :, 0].max(axis=1) - st[:, :, 0].min(axis=1)) * tick_us
        print(f"    entry skew inside a workgroup (last wave's entry - first): mean {entry_skew.mean():.2f} max {entry_skew.max():.2f} us")
        wl.engine.close()


if __name__ == "__main__":
    main()
