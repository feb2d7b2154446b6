"""Probe: capture (lnprob kernels + RCCL all_gather) into a HIP graph via torch and replay it."""
import os, sys, time
import numpy as np
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rbvfit_amd.workloads import make_workload
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
wl = make_workload("C1"); eng = wl.engine; W = 512
d_theta = torch.from_numpy(wl.thetas).cuda(); d_out = torch.empty(W, dtype=torch.float64, device="cuda")
gathered = torch.empty(W * dist.get_world_size(), dtype=torch.float64, device="cuda")
def step():
    eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, torch.cuda.current_stream().cuda_stream)
    dist.all_gather_into_tensor(gathered, d_out)
for _ in range(20): step()
torch.cuda.synchronize()
def timeit(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("eager step us:", timeit(step))
def only():
    eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, torch.cuda.current_stream().cuda_stream)
print("lnprob only us:", timeit(only))
def ag():
    dist.all_gather_into_tensor(gathered, d_out)
print("all_gather only us:", timeit(ag))
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize()
    print("graph replay us:", timeit(g.replay))
    ref = eng.lnprob(wl.thetas)
    print("graph result ok:", np.array_equal(gathered.cpu().numpy()[:W], ref))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
dist.destroy_process_group()
