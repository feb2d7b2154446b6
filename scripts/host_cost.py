"""Host-side issue cost of one lnprob_device call (GPU kept nearly idle with a tiny batch) and of
the Python around it; run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rbvfit_amd.workloads import make_workload

for W in (2, 512):
    wl = make_workload("C1", walkers=W)
    eng = wl.engine
    d_theta = torch.from_numpy(wl.thetas).cuda()
    d_out = torch.empty(W, dtype=torch.float64, device="cuda")
    rows = torch.empty(8, W, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    tp, op = d_theta.data_ptr(), d_out.data_ptr()
    for name, fn in (("raw ptrs", lambda: eng.lnprob_device(tp, op, W, s)),
                     ("data_ptr()", lambda: eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, s)),
                     ("row view", lambda: eng.lnprob_device(tp, rows[3].data_ptr(), W, s))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"W={W:4d} {name:11s}: issue {1e6*(t1-t0)/n:6.2f} us/call, with drain {1e6*(t2-t0)/n:6.2f} us/call", flush=True)
