"""Where does the pipelined all-gather cost go with one rank?  (GPU box; BENCH-like loop)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29519")):
    os.environ.setdefault(k, v)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from rbvfit_amd.workloads import make_workload
from rbvfit_amd.dist import PipelinedGather
wl = make_workload("C1", walkers=512)
eng, W = wl.engine, 512
d_theta = torch.from_numpy(wl.thetas).cuda()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
launch = lambda out: eng.lnprob_device(d_theta.data_ptr(), out.data_ptr(), W, s.cuda_stream)
for every in (8, 32, 1000):
    pg = PipelinedGather(launch, W, device="cuda", every=every)
    ship_t = []
    orig = pg._ship
    def timed_ship(i, rows, orig=orig):
        t = time.perf_counter(); orig(i, rows); ship_t.append(time.perf_counter() - t)
    pg._ship = timed_ship
    for _ in range(40): pg.step()
    pg.flush(); torch.cuda.synchronize(); ship_t.clear()
    N = 400
    t0 = time.perf_counter()
    for _ in range(N): pg.step()
    t1 = time.perf_counter()
    pg.flush()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"every={every:5d}: issue {1e6*(t1-t0)/N:6.2f} us/step, flush {1e6*(t2-t1):7.1f} us, drain {1e6*(t3-t2):8.1f} us, "
          f"total {1e6*(t3-t0)/N:6.2f} us/step; ships {len(ship_t)} mean {1e6*sum(ship_t)/max(len(ship_t),1):7.1f} us max {1e6*max(ship_t or [0]):7.1f}", flush=True)
dist.destroy_process_group()
