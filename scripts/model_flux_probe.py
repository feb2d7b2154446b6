import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from rbvfit_amd.workloads import make_workload
for cfg, W in (("C1", 512), ("C2", 1024)):
    wl = make_workload(cfg, walkers=W)
    eng = wl.engine
    th = torch.tensor(wl.thetas, device="cuda")
    P = eng.n_pixels[0]
    out = torch.empty((W, P), dtype=torch.float64, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(300):
            eng.model_flux_device(0, th.data_ptr(), out.data_ptr(), W, True, st.cuda_stream)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(400):
                eng.model_flux_device(0, th.data_ptr(), out.data_ptr(), W, True, st.cuda_stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 400)
    print(cfg, W, "model_flux device-resident us/batch", round(1e6 * float(np.median(ts)), 2), flush=True)
    eng.close()
