"""GPU box, diagnostic build (-DVP_STAMPS): a pre-armed walker_kernel launch on the 100 MHz clock all XCDs share.

    RBVFIT_AMD_LIB=$PWD/rbvfit_amd/lib/exp/lib_stamps.so python scripts/prearm_timeline.py [walkers ...]

Relative to the moment the polling wave (workgroup 0, wave 0) has the go word: when the record waves of the other workgroups have
it, when the workgroups pass their barrier (records formed), when they finish; next to the host's own stamps of the same call
(ns since entry: theta staged, go word / launch, first row seen, all rows seen, copied out)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rbvfit_amd import _lib as L                     # noqa: E402
from rbvfit_amd.workloads import make_workload       # noqa: E402

NW, NWAVES, NST = 1024, 16, 16


def main():
    lib = L.load()
    lib.vp_debug_read_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    lib.vp_debug_host_stamps.argtypes = [C.POINTER(C.c_double)]
    for W in [int(a) for a in sys.argv[1:]] or [256, 512]:
        for mode in (1, 0):
            wl = make_workload("C1", walkers=W)
            eng = wl.engine
            eng.set_option("walker", 1)
            eng.set_option("prearm", mode)
            th = np.ascontiguousarray(wl.thetas)
            for _ in range(200):
                eng.lnprob(th)
            rows = []
            hst = []
            for rep in range(50):
                eng.lnprob(th)
                h = np.zeros(8)
                lib.vp_debug_host_stamps(h.ctypes.data_as(C.POINTER(C.c_double)))
                hst.append(h / 1e3)
                buf = np.zeros(NW * NWAVES * NST, dtype=np.int64)
                assert lib.vp_debug_read_stamps(buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size) == 0     # (cancels a waiting launch)
                st = buf.reshape(NW, NWAVES, NST)[:W, :12, :].astype(np.float64) / 100.0      # us
                t_entry = st[:, :, 14]
                t_bar = st[:, :, 13]
                t_end = st[:, :, 15]
                if mode:
                    t_go = st[:, :2, 12]
                    T0 = t_go[0, 0]
                else:
                    T0 = t_entry.min()
                    t_go = t_entry[:, :2]
                rows.append([np.median(t_go[:, 0] - T0), (t_go[:, :2] - T0).max(), np.median(t_bar[:, 0] - T0), (t_bar - T0).max(),
                             np.median(t_end.max(axis=1) - T0), (t_end - T0).max(), T0 - t_entry.min(), t_entry.max() - t_entry.min()])
                for _ in range(3):
                    eng.lnprob(th)          # (the chain of pre-armed launches starts again)
            r = np.median(np.array(rows), axis=0)
            h = np.median(np.array(hst), axis=0)
            what = "go word seen by the poller" if mode else "first wave's entry"
            print(f"C1 W={W} prearm={mode}: us since {what}: record waves have it median {r[0]:.2f} / last {r[1]:.2f}; barrier median {r[2]:.2f} / last {r[3]:.2f}; "
                  f"workgroup done median {r[4]:.2f} / last {r[5]:.2f}   (first entry -> that moment {r[6]:.2f}; entries spread over {r[7]:.2f})")
            print(f"    host, us since the C call's entry: theta staged {h[0]:.2f}, go / launches enqueued {h[1]:.2f}, first row {h[2]:.2f}, all rows {h[3]:.2f}, copied out {h[4]:.2f}")
            eng.close()


if __name__ == "__main__":
    main()
