"""GPU box: two contexts on two streams, passes enqueued alternately (bench.py's `two_ensembles_side_by_side`), for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/side/trace -- python3 scripts/side_by_side.py 256
    python3 scripts/side_by_side.py --read gpurun_out/side/trace
prints how much of the time two walker kernels are on the GPU at once and which queues they came through."""
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(W, mode):
    import torch
    from rbvfit_amd.workloads import make_workload
    a, b = make_workload("C1", walkers=W, walker_seed=1), make_workload("C1", walkers=W, walker_seed=2)
    ta, tb = torch.from_numpy(a.thetas).cuda(), torch.from_numpy(b.thetas).cuda()
    oa, ob = torch.empty(W, dtype=torch.float64, device="cuda"), torch.empty(W, dtype=torch.float64, device="cuda")
    if mode == "torch":
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        ha, hb = sa.cuda_stream, sb.cuda_stream
    else:                                  # the contexts' own streams (handle 0)
        ha = hb = 0
    for n in (200, 400):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            a.engine.lnprob_device(ta.data_ptr(), oa.data_ptr(), W, ha)
            b.engine.lnprob_device(tb.data_ptr(), ob.data_ptr(), W, hb)
        if mode != "torch":
            a.engine.synchronize() if hasattr(a.engine, "synchronize") else None
            b.engine.synchronize() if hasattr(b.engine, "synchronize") else None
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"W={W} streams={mode}: {1e6 * dt:.2f} us per pair of passes")
    t0 = time.perf_counter()
    for _ in range(400):
        a.engine.lnprob_device(ta.data_ptr(), oa.data_ptr(), W, ha)
    torch.cuda.synchronize()
    print(f"W={W} one context alone: {1e6 * (time.perf_counter() - t0) / 400:.2f} us per pass")


def read(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "walker_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r.get("Stream_Id")))
    rows.sort()
    if not rows:
        print("no walker kernels in the trace under", d)
        return
    rows = rows[len(rows) // 2:]
    queues = sorted({(q, s) for _, _, q, s in rows})
    ov = sum(max(0, min(rows[i][1], rows[i + 1][1]) - rows[i + 1][0]) for i in range(len(rows) - 1))
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _, _ in rows)
    print(f"{len(rows)} walker kernels, (queue, stream) ids {queues}; mean duration {busy / len(rows) / 1e3:.2f} us; "
          f"two on the GPU at once for {100.0 * ov / span:.1f} % of the span; span per kernel {span / len(rows) / 1e3:.2f} us")


if __name__ == "__main__":
    if sys.argv[1] == "--read":
        read(sys.argv[2])
    else:
        run(int(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else "torch")
