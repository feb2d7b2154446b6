"""Bandwidth of a ONE-rank all_gather_into_tensor (RCCL's degenerate path), for reading the pipelined-gather numbers."""
import os, time
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
    os.environ.setdefault(k, v)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
for nbytes in (4096, 1 << 19, 1 << 23):
    a = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): dist.all_gather_into_tensor(b, a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dist.all_gather_into_tensor(b, a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{nbytes:9d} B: {1e6*dt:8.1f} us per all_gather ({nbytes/dt/1e9:6.2f} GB/s)", flush=True)
dist.destroy_process_group()
