"""GPU box: what halving the tile size buys the walker kernel where it is latency-bound (<= 256 walkers on C1).

Emulation with what exists: C1's spectrum sampled on 2048 pixels, 512 walkers, walker kernel forced -- with the default tiles of
384 evaluated pixels (6 waves per workgroup, 3 per SIMD: the load of C1 at 256 walkers) and with RBVFIT_AMD_SPAN=192 (13 waves
per workgroup: each workgroup is what HALF a walker would be if a walker's tiles were one-pass tiles dealt to two workgroups)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(span, walkers, pixels):
    if span:
        os.environ["RBVFIT_AMD_SPAN"] = str(span)
    else:
        os.environ.pop("RBVFIT_AMD_SPAN", None)
    from rbvfit_amd.workloads import make_workload
    wl = make_workload("C1", walkers=walkers, pixels=pixels)
    eng = wl.engine
    eng.set_option("walker", 1)
    th = torch.from_numpy(wl.thetas).cuda()
    out = torch.empty(walkers, dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2000):
            eng.lnprob_device(th.data_ptr(), out.data_ptr(), walkers, s.cuda_stream)
        torch.cuda.synchronize()
        kind = eng.last_launch_kind
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        n = 3000
        for _ in range(n):
            eng.lnprob_device(th.data_ptr(), out.data_ptr(), walkers, s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    print(f"pixels {pixels} walkers {walkers} span {span or 384}: {kind}, {us:.2f} us per pass", flush=True)
    eng.close()


if __name__ == "__main__":
    for walkers, pixels in ((512, 2048), (256, 2048), (128, 2048), (256, 4096), (128, 4096), (64, 4096)):
        for span in (0, 192):
            if pixels == 4096 and span:
                continue
            run(span, walkers, pixels)
