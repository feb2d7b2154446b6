"""GPU box: would ONE pass cut into two half-batches on two streams (joined by events every pass) be shorter than the pass as it is?
Two contexts with W/2 walkers each stand in for the two halves (the launch structure of a half is that of the whole for these
configurations); streams and events straight from the HIP runtime (two streams of torch's pool do not run side by side here).
    python3 scripts/split_pass_probe.py C2 1024"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    from rbvfit_amd.workloads import make_workload
    cfg, W = sys.argv[1], int(sys.argv[2])
    hip = C.CDLL("libamdhip64.so")
    def stream():
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        return s
    def event():
        e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0
        return e
    whole = make_workload(cfg, walkers=W)
    a, b = make_workload(cfg, walkers=W // 2, walker_seed=1), make_workload(cfg, walkers=W // 2, walker_seed=2)
    tw, ta, tb = (torch.from_numpy(x.thetas).cuda() for x in (whole, a, b))
    ow = torch.empty(W, dtype=torch.float64, device="cuda")
    oa, ob = torch.empty(W // 2, dtype=torch.float64, device="cuda"), torch.empty(W // 2, dtype=torch.float64, device="cuda")
    sA, sB, e0, e1 = stream(), stream(), event(), event()
    n = 300 if cfg in ("C0", "C1") else (60 if cfg != "C4" else 30)

    def one():
        whole.engine.lnprob_device(tw.data_ptr(), ow.data_ptr(), W, sA.value)

    def halves():
        hip.hipEventRecord(e0, sA); hip.hipStreamWaitEvent(sB, e0, 0)
        a.engine.lnprob_device(ta.data_ptr(), oa.data_ptr(), W // 2, sA.value)
        b.engine.lnprob_device(tb.data_ptr(), ob.data_ptr(), W // 2, sB.value)
        hip.hipEventRecord(e1, sB); hip.hipStreamWaitEvent(sA, e1, 0)

    def free():
        a.engine.lnprob_device(ta.data_ptr(), oa.data_ptr(), W // 2, sA.value)
        b.engine.lnprob_device(tb.data_ptr(), ob.data_ptr(), W // 2, sB.value)

    for name, fn in (("whole batch, one stream", one), ("two halves, joined every pass", halves), ("two halves, free-running", free),
                     ("whole batch, one stream", one)):
        for _ in range(max(20, n // 3)):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / n)
        print(f"{cfg} {W} walkers: {name}: {1e6 * sorted(ts)[2]:.1f} us per pass", flush=True)


if __name__ == "__main__":
    main()
