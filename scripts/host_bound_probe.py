"""Is the back-to-back pass rate sensitive to extra host time per pass (while the host is nominally far
ahead of the GPU)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rbvfit_amd.workloads import make_workload
wl = make_workload("C1", walkers=512); eng = wl.engine; W = 512
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
d_theta = torch.from_numpy(wl.thetas).cuda(); d_out = torch.empty(W, dtype=torch.float64, device="cuda")
tp, op, sp = d_theta.data_ptr(), d_out.data_ptr(), s.cuda_stream
def spin(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t: pass
for _ in range(2000): eng.lnprob_device(tp, op, W, sp)
torch.cuda.synchronize()
for extra in (0, 1, 2, 5, 10, 0):
    t0 = time.perf_counter()
    for _ in range(3000):
        eng.lnprob_device(tp, op, W, sp)
        if extra: spin(extra)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"extra host {extra:2d} us: issue {1e6*(t1-t0)/3000:6.2f} us/pass, total {1e6*(t2-t0)/3000:6.2f} us/pass", flush=True)
