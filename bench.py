#!/usr/bin/env python
"""bench.py -- walker-lnprob evals/sec on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N=1 directly; N>1 under torchrun)

A "step" is one pass of the hot path over one batch: lnprob of the rank's W_local walkers
(theta already resident in HBM, lnprob left in HBM) followed, for N > 1, by the RCCL all-gather
of the per-walker lnprob vector.  Workload at every N: BASELINE.json configs[1] per GPU ("C1":
MgII 2796/2803, z=0.348, 2 components, 4096 px, 23-tap Gaussian LSF, 512 walkers per GPU => weak
scaling).  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(wl, budget_s=12.0):
    """Oracle (NumPy/SciPy restatement of the reference path) timed serially on this box's host
    cores: single-theta lnprob over the same walker rows, what emcee does with pool=None."""
    from oracle import voigt_oracle as vo           # checker / baseline only, never the product
    insts = []
    for data, (wave, flux, err) in zip(wl.tables, wl.spectra):
        od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors,
                                data.N_indices, data.b_indices, data.v_indices,
                                data.taps if data.taps is not None else np.zeros(0), data.lsf_mode,
                                data.voigt_method)
        insts.append(vo.OracleInstrument.from_error(od, wave, flux, err))
    vo.lnprob(wl.thetas[0], wl.lb, wl.ub, insts)            # warm-up
    n, t0 = 0, time.perf_counter()
    vals = []
    while True:
        vals.append(vo.lnprob(wl.thetas[n % len(wl.thetas)], wl.lb, wl.ub, insts))
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 20 * len(wl.thetas):
            break
    base = dict(value=n / dt, unit="walker-lnprob evals/s", cores=1, kind="port",
                sample=f"{n} serial single-theta lnprob calls over the {wl.name} walker rows "
                       f"({dt:.1f} s, numpy/scipy oracle)")
    # second, stronger CPU number: the plain-C restatement with OpenMP over walkers on all cores
    try:
        from oracle import c_oracle
        co = c_oracle.COracle(insts, wl.lb, wl.ub)
        cores = min(16, len(os.sched_getaffinity(0)))     # the GPU box's CPU share per GPU
        co.lnprob_batch(wl.thetas[:cores], nthreads=cores)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 4.0:
            cvals = co.lnprob_batch(wl.thetas, nthreads=cores)
            reps += 1
        dtc = time.perf_counter() - t0
        base["c_openmp"] = dict(value=reps * len(wl.thetas) / dtc, cores=cores, kind="port",
                                sample=f"{reps} x {len(wl.thetas)} walkers, oracle/voigt_oracle.c, OpenMP",
                                max_rel_vs_numpy_oracle=float(np.max(np.abs(cvals[:len(vals)] / np.array(vals[:len(cvals)]) - 1))))
    except Exception as e:                                   # the C oracle is optional for the baseline
        base["c_openmp"] = {"error": str(e)}
    return base, np.array(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C1")
    ap.add_argument("--walkers", type=int, default=None, help="walkers per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torchrun with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1 exercises the RCCL path (init + all_gather) even with a single rank
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if use_dist:
        dist.barrier()
    from rbvfit_amd.workloads import make_workload

    # every rank owns a different block of walkers of the same ensemble (walker_seed = rank)
    wl = make_workload(args.config, walkers=args.walkers, device_id=local_rank, walker_seed=1 + rank)
    eng = wl.engine
    W, D = wl.thetas.shape
    d_theta = torch.from_numpy(wl.thetas).cuda()
    d_out = torch.empty(W, dtype=torch.float64, device="cuda")
    gathered = torch.empty(W * world, dtype=torch.float64, device="cuda") if use_dist else None
    stream = torch.cuda.current_stream()

    def step():
        eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, stream.cuda_stream)
        if use_dist:
            dist.all_gather_into_tensor(gathered, d_out)

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline leg: HIP events around the tile kernel on its launch stream (rank 0) -------
    roof = None
    if rank == 0:
        eng.profile_enable(True)
        nprof = min(args.steps, 50)
        for _ in range(nprof):
            eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, stream.cuda_stream)
        torch.cuda.synchronize()
        pr = eng.profile_read()
        eng.profile_enable(False)
        tile_ms = pr["tile_ms"] / max(pr["n_tile_launches"], 1)
        bytes_per_launch = wl.algorithmic_bytes_per_eval * W / len(wl.pixels)
        achieved = bytes_per_launch / (tile_ms * 1e-3) / 1e9
        # HBM traffic per launch from the committed PMC passes (FETCH_SIZE x2 per the gfx950 guide
        # + WRITE_SIZE, separate rocprofv3 --pmc runs of this same command): profiles/<round>_pmc.json
        traffic, traffic_src = None, None
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
            if cands:
                pm = json.load(open(cands[-1]))
                if pm.get("config") == args.config and pm.get("walkers_per_gpu") == W:
                    traffic, traffic_src = pm["tile_kernel_hbm_bytes_per_launch"], os.path.basename(cands[-1])
        except Exception:
            pass
        roof = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, traffic_source=traffic_src, kernel="vp::tile_kernel<0,0>", avg_kernel_ms=tile_ms,
                    algorithmic_bytes_per_launch=bytes_per_launch, prep_ms=pr["prep_ms"] / nprof,
                    note="kernel is fp64-VALU / latency bound; spectra are shared by all walkers through L2/MALL, "
                         "so measured HBM traffic is far below the algorithmic bytes (DESIGN.md section 4)")
        # PCIe-inclusive rate through the host-buffer entry (never `value`)
        nh = 20
        wl.engine.lnprob(wl.thetas)
        th0 = time.perf_counter()
        for _ in range(nh):
            wl.engine.lnprob(wl.thetas)
        host_rate = nh * W / (time.perf_counter() - th0)

    result = d_out.cpu().numpy()
    if rank == 0:
        evals = W * world * args.steps
        line = {
            "metric": "walker-lnprob evals/sec", "value": evals / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: MgII 2796/2803 z=0.348, 2 components, 4096 px, 23-tap Gaussian LSF"
                                   if args.config in ("C0", "C1") else args.config,
                       "walkers_per_gpu": W, "walkers_total": W * world, "ndim": D, "n_lines": wl.n_lines,
                       "pixels": wl.pixels, "parallelism": f"walker-shard x{world}" + (" + RCCL all_gather" if world > 1 else "")},
            "mcmc_steps_per_sec": (evals / elapsed) / (W * world),
            "host_entry_evals_per_sec_pcie_inclusive": host_rate,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_vals = cpu_baseline(wl)
            n = min(len(cpu_vals), W)
            line["cpu_baseline"] = cb
            line["parity_vs_cpu_baseline_max_rel"] = float(np.max(np.abs(result[:n] / cpu_vals[:n] - 1)))
        print(json.dumps(line), flush=True)
    if use_dist:
        if rank == 0 and gathered is not None:
            assert torch.equal(gathered[:W], d_out), "all-gathered lnprob differs from the local block"
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
