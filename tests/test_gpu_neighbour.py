"""Pre-armed launches as a neighbour on the GPU: a launch that waits for the caller's next batch holds every
compute unit, and only this library's entry points can send it away -- a kernel that anything else launches meanwhile (here: torch
ops of the same process) waits until it leaves.  The wait a launch is given follows the caller's own rhythm (1.5 x its recent gap
between calls + 10 us, capped by "prearm_us"), and a caller whose launches expire unused is left alone for 8, 16, 32 ... calls."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch_op_latency(torch, x, n=1):
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        y = x * 2.0
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    return np.array(out)


def test_torch_kernels_next_to_a_loop_of_lnprob_calls():
    torch = pytest.importorskip("torch")
    from rbvfit_amd.workloads import make_workload
    wl = make_workload("C1", walkers=512)
    eng, th = wl.engine, wl.thetas
    try:
        ref = eng.lnprob(th)
        x = torch.ones(1 << 16, device="cuda")
        _torch_op_latency(torch, x, 50)
        base = np.median(_torch_op_latency(torch, x, 200))                 # the op alone: launch + synchronise
        # (a) behind a tight loop of calls: the launch left waiting gives up after ~20 us
        lat_a = []
        for rep in range(20):
            for _ in range(200):
                eng.lnprob(th)
            lat_a.append(_torch_op_latency(torch, x)[0])
        used = eng.prearm_counts["used"]
        assert used > 2000, eng.prearm_counts                               # the loop itself ran on pushed-to launches
        assert np.percentile(lat_a, 90) < base + 100e-6, (base, lat_a)
        # (b) a torch op between every two calls: the first launches expire unused (the op waits them out once), then the caller
        # is left alone for longer and longer stretches
        c0 = eng.prearm_counts
        lat_b = []
        for k in range(400):
            got = eng.lnprob(th)
            lat_b.append(_torch_op_latency(torch, x)[0])
        assert np.array_equal(got, ref, equal_nan=True)
        c1 = eng.prearm_counts
        assert c1["expired"] - c0["expired"] <= 12, (c0, c1)                 # 8, 16, 32, ... calls between two tries
        assert np.percentile(lat_b, 90) < base + 100e-6, (base, np.percentile(lat_b, [50, 90, 99]))
        # (c) off: nothing ever waits on the GPU
        eng.set_option("prearm", 0)
        c2 = eng.prearm_counts
        for _ in range(50):
            eng.lnprob(th)
        assert eng.prearm_counts == c2
        assert _torch_op_latency(torch, x)[0] < base + 100e-6
    finally:
        eng.close()
