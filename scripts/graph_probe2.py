"""Probe: lnprob pass (prep + tile [+ finalize]) replayed from a HIP graph vs launched eagerly, steady clocks."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rbvfit_amd.workloads import make_workload
for W in (256, 512, 2048):
    wl = make_workload("C1", walkers=W); eng = wl.engine
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    d_theta = torch.from_numpy(wl.thetas).cuda(); d_out = torch.empty(W, dtype=torch.float64, device="cuda")
    def step():
        eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, s.cuda_stream)
    def timeit(fn, n=2000):
        for _ in range(1500): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    te = timeit(step)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize()
    tg = timeit(g.replay)
    ok = np.array_equal(d_out.cpu().numpy(), eng.lnprob(wl.thetas))
    print(f"W={W:5d}: eager {te:6.2f} us/pass, graph replay {tg:6.2f} us/pass, result ok {ok}", flush=True)
