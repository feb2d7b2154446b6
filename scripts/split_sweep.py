"""GPU box, round 5: walker_kernel's split form (several workgroups of one-pass tiles per walker) against the ordinary form by
batch size -- device-resident lnprob passes, the host entry (vp_lnprob_batch, pre-armed launches in play), and the device
stretch sampler.  Usage: split_sweep.py [lnprob|host|stretch ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rbvfit_amd.workloads import make_workload

what = sys.argv[1:] or ["lnprob", "host", "stretch"]


def resident(eng, th_np, n=3000):
    W = len(th_np)
    th = torch.from_numpy(th_np).cuda()
    out = torch.empty(W, dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(1500):
            eng.lnprob_device(th.data_ptr(), out.data_ptr(), W, s.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n):
            eng.lnprob_device(th.data_ptr(), out.data_ptr(), W, s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n, out.cpu().numpy()


if "lnprob" in what:
    wl = make_workload("C1", walkers=512)
    eng = wl.engine
    print("device-resident lnprob, C1, us per pass by walkers x workgroups per walker (0 = ordinary form; * = the automatic choice)", flush=True)
    for W in (16, 32, 50, 64, 100, 128, 192, 256, 384, 512):
        row, ref = [], None
        eng.set_option("walker_split", -1)
        eng.lnprob(wl.thetas[:W]); auto = eng.last_walker_split
        for G in (0, 2, 4, 8):
            eng.set_option("walker_split", G)
            us, got = resident(eng, wl.thetas[:W])
            used = eng.last_walker_split
            if G and used:
                ref = got if ref is None else ref
                assert np.array_equal(got, ref)
            row.append(f"{G}: {us:6.2f}{'*' if used == auto else ' '}" + ("" if used == G else f"(ran {used})"))
        print(f"  W={W:4d}  " + "   ".join(row), flush=True)
    eng.set_option("walker_split", -1)
    eng.close()

if "host" in what:
    print("host entry vp_lnprob_batch (Engine.lnprob, back-to-back calls), us per call: ordinary / split(auto)", flush=True)
    for W in (16, 50, 64, 128, 256):
        wl = make_workload("C1", walkers=W)
        eng, th = wl.engine, wl.thetas
        res = []
        for G in (0, -1):
            eng.set_option("walker_split", G)
            for _ in range(3000):
                eng.lnprob(th)
            t0 = time.perf_counter(); n = 20000
            for _ in range(n):
                eng.lnprob(th)
            res.append(1e6 * (time.perf_counter() - t0) / n)
        print(f"  W={W:4d}  {res[0]:6.2f} / {res[1]:6.2f}   prearm counts {eng.prearm_counts}", flush=True)
        eng.close()

if "stretch" in what:
    print("device stretch sampler (vp_stretch_run), ensemble steps/s: ordinary / split(auto); chains compared with a replay-free check", flush=True)
    for W in (32, 64, 128, 256, 512):
        wl = make_workload("C1", walkers=W)
        eng = wl.engine
        res = []
        for G in (0, -1):
            eng.set_option("walker_split", G)
            eng.stretch_run(wl.thetas, 50, seed=1, store_chain=False)
            nst = 800
            t0 = time.perf_counter()
            r = eng.stretch_run(wl.thetas, nst, seed=1, store_chain=False)
            dt = time.perf_counter() - t0
            ok = np.array_equal(r[1], eng.lnprob(r[0]))          # stored lnprob = lnprob of the final positions, same geometry
            res.append((nst / dt, ok))
        print(f"  W={W:4d}  {res[0][0]:8.0f} / {res[1][0]:8.0f} steps/s   lnprob of final state consistent: {res[0][1]} {res[1][1]}", flush=True)
        eng.close()
