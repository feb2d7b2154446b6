"""Does the step time depend on how long the GPU has been busy?  (clock ramp)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rbvfit_amd.workloads import make_workload
wl = make_workload("C1", walkers=512)
eng, W = wl.engine, 512
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
d_theta = torch.from_numpy(wl.thetas).cuda()
d_out = torch.empty(W, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
time.sleep(0.5)
for rep in range(8):
    t0 = time.perf_counter()
    for _ in range(200):
        eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, s.cuda_stream)
    torch.cuda.synchronize()
    print(f"block {rep}: {1e6*(time.perf_counter()-t0)/200:6.2f} us/step", flush=True)
time.sleep(0.5)
t0 = time.perf_counter()
for _ in range(200):
    eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, s.cuda_stream)
torch.cuda.synchronize()
print(f"after 0.5 s idle: {1e6*(time.perf_counter()-t0)/200:6.2f} us/step")

# --- does a moving output pointer change the GPU time per step? (PipelinedGather writes row k % every)
rows = torch.empty(128, W, dtype=torch.float64, device="cuda")
ptrs = [rows[j].data_ptr() for j in range(128)]
for name, get in (("fixed out", lambda k: d_out.data_ptr()), ("moving out (precomputed ptrs)", lambda k: ptrs[k % 128]),
                  ("moving out (tensor row view per step)", lambda k: rows[k % 128].data_ptr())):
    for _ in range(1500):
        eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(2000):
        eng.lnprob_device(d_theta.data_ptr(), get(k), W, s.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:40s}: {1e6*(time.perf_counter()-t0)/2000:6.2f} us/step (host issue {1e6*(t1-t0)/2000:5.2f})", flush=True)
