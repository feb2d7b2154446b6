"""GPU box, diagnostic build (-DVP_STAMPS): where a vp_lnprob_batch call with host buffers spends its time.

    RBVFIT_AMD_LIB=$PWD/rbvfit_amd/lib/exp/lib_stamps.so python scripts/seam_stamps.py [walkers ...]

Per call (median over 2000 calls, ns since the call's entry): theta staged in pinned memory -> launches enqueued -> first
output row seen by the polling host -> all rows seen -> copied out; next to the wall time per call measured around the ctypes
call and the device-resident pass time."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rbvfit_amd import _lib as L                     # noqa: E402
from rbvfit_amd.workloads import make_workload       # noqa: E402


def main():
    lib = L.load()
    lib.vp_debug_host_stamps.argtypes = [C.POINTER(C.c_double)]
    for W in [int(a) for a in sys.argv[1:]] or [1, 256, 512]:
        for spin in (1, 2):
            wl = make_workload("C1", walkers=W)
            eng = wl.engine
            eng.set_option("host_spin", spin)
            th = np.ascontiguousarray(wl.thetas)
            for _ in range(200):
                eng.lnprob(th)
            n = 2000
            st = np.zeros((n, 8))
            wall = np.zeros(n)
            for i in range(n):
                t0 = time.perf_counter()
                eng.lnprob(th)
                wall[i] = time.perf_counter() - t0
                lib.vp_debug_host_stamps(st[i].ctypes.data_as(C.POINTER(C.c_double)))
            med = np.median(st, axis=0) / 1e3
            print(f"C1 W={W} host_spin={spin}: wall {1e6 * np.median(wall):.2f} us/call | staged {med[0]:.2f} | enqueued {med[1]:.2f} | "
                  f"first row {med[2]:.2f} | all rows / done {med[3]:.2f} | copied out {med[4]:.2f}  (us since entry of the C call)")
            eng.close()


if __name__ == "__main__":
    main()
