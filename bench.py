#!/usr/bin/env python
"""bench.py -- walker-lnprob evals/sec on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N=1 directly; N>1 under torchrun)

A "step" is one pass of the hot path over one batch: lnprob of the rank's W_local walkers
(theta already resident in HBM, lnprob left in HBM) plus, for N > 1, the RCCL all-gather of the
per-walker lnprob vector -- issued asynchronously for chunks of 128 steps and double-buffered
(island ensembles, rbvfit_amd/dist.py), every gather completing inside the timed region; the blocking-gather and
gather-free step times are reported beside it.  Workload at every N: BASELINE.json configs[1] per GPU ("C1":
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
FP64_VALU_PEAK_TFLOPS = 78.6   # fp64 vector = half the 157.3 TF fp32 vector rate of the same guide


# ---- CPU fan-out leg: the reference's Pool.map over walker rows --------------------
_POOL_CACHE = {}


def _pool_chunk(task):
    """Worker: serial oracle lnprob over a chunk of walker rows (what each forked worker of
    `OptimizedPool` does, vfit_mcmc.py:35-49)."""
    token, descr, lb, ub, rows = task
    from oracle import voigt_oracle as vo           # baseline only, never the product
    insts = _POOL_CACHE.get(token)
    if insts is None:
        insts = _POOL_CACHE[token] = _oracle_instruments(vo, descr)
    return [vo.lnprob(r, lb, ub, insts) for r in rows]


def _oracle_instruments(vo, descr):
    insts = []
    for (lam, gam, f, zf, ni, bi, vi, taps, mode, method), (wave, flux, err) in descr:
        od = vo.OracleModelData(lam, gam, f, zf, ni, bi, vi, taps, mode, method)
        insts.append(vo.OracleInstrument.from_error(od, wave, flux, err))
    return insts


def _describe(wl):
    return [((d.atomic_lambda0, d.atomic_gamma, d.atomic_f, d.z_factors, d.N_indices, d.b_indices, d.v_indices,
              d.taps if d.taps is not None else np.zeros(0), d.lsf_mode, d.voigt_method), sp)
            for d, sp in zip(wl.tables, wl.spectra)]


def cpu_pool_baseline(pool, cores, wl, budget_s=5.0):
    descr, rows = _describe(wl), wl.thetas
    chunk = max(1, len(rows) // (4 * cores))         # Pool.map's default chunking
    tasks = [(wl.name, descr, wl.lb, wl.ub, rows[i:i + chunk]) for i in range(0, len(rows), chunk)]
    pool.map(_pool_chunk, tasks[:cores], chunksize=1)           # imports + first-touch, untimed
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        out = pool.map(_pool_chunk, tasks, chunksize=1)
        n += sum(len(o) for o in out)
    dt = time.perf_counter() - t0
    return dict(value=n / dt, cores=cores, kind="port",
                sample=f"{n} lnprob calls through multiprocessing fork Pool({cores}).map over the {wl.name} "
                       f"walker rows ({dt:.1f} s, numpy/scipy oracle)")


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
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="C1")
    ap.add_argument("--walkers", type=int, default=None, help="walkers per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs that are not part of `value` (copy bandwidth, host-entry latency, device sampler): "
                         "used under rocprofv3 so that every tile-kernel launch in the trace is the benchmarked one")
    args = ap.parse_args()

    # The CPU fan-out leg forks its workers BEFORE anything touches the GPU (a context is not
    # fork-safe and forked children must not hold the device); they idle until the bench is done.
    pool, pool_cores = None, 0
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline
            and not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)):
        import multiprocessing as mp
        pool_cores = min(16, len(os.sched_getaffinity(0)))
        pool = mp.get_context("fork").Pool(pool_cores)

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
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
            os.environ.setdefault(k, v)            # only matters for a BENCH_FORCE_DIST=1 run without torchrun
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
    # One explicit (non-default) stream carries everything: the engine's kernels are enqueued on its
    # handle and torch / c10d order their work against it as the current stream.  (The default
    # stream's handle is 0, which the C ABI reads as "the context's own stream" -- kernels there would
    # not be ordered with the collectives.)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    d_theta = torch.from_numpy(wl.thetas).cuda()
    d_out = torch.empty(W, dtype=torch.float64, device="cuda")
    gathered = torch.empty(W * world, dtype=torch.float64, device="cuda") if use_dist else None
    torch.cuda.synchronize()

    def launch(out):
        eng.lnprob_device(d_theta.data_ptr(), out.data_ptr(), W, stream.cuda_stream)

    pg = None
    if use_dist:
        # island form (rbvfit_amd.dist): a rank's accept/reject needs its own lnprob only, so the
        # all-gather of a chunk of steps runs on RCCL's stream, double-buffered, under the next chunk's kernels
        from rbvfit_amd.dist import PipelinedGather
        gather_every = int(os.environ.get("BENCH_GATHER_EVERY", "128"))
        pg = PipelinedGather(launch, W, device="cuda", every=gather_every)

    def step():
        if pg is not None:
            pg.step()
        else:
            launch(d_out)

    def timed(fn, n, drain=None):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        if drain is not None:
            drain()                                   # every collective of the timed steps completes inside
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # The GPU needs ~25 ms of sustained load to reach its steady clocks (scripts/warm_probe.py: 37.9 us
    # per step in the first 8 ms after an idle period, 32.9 us from ~25 ms on); a sampler runs for
    # minutes, so the untimed part is made long enough whatever --warmup says.
    for _ in range(max(args.warmup, 1500)):
        step()
    if pg is not None:
        pg.flush()
    elapsed = timed(step, args.steps, drain=pg.flush if pg is not None else None)

    # strict single-ensemble form (every rank needs the full vector before the next half-step):
    # blocking all-gather after each pass; and the gather-free form -- neither is `value`
    sync_ms = nogather_ms = None
    if use_dist:
        def sync_step():
            launch(d_out)
            dist.all_gather_into_tensor(gathered, d_out)
        for _ in range(min(args.warmup, 5)):
            sync_step()
        sync_ms = 1e3 * timed(sync_step, args.steps) / args.steps
        nogather_ms = 1e3 * timed(lambda: launch(d_out), args.steps) / args.steps
        last = pg.chunk(0)
        assert torch.equal(last[rank, last.shape[1] - 1], d_out), "pipelined gather differs from the local block"

    # ---- roofline leg: HIP events around the tile kernel on its launch stream (rank 0) -------
    roof = None
    if rank == 0:
        eng.profile_enable(True)
        nprof = min(args.steps, 200)
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
                    finalize_ms=pr.get("finalize_ms", 0.0) / nprof,
                    note="kernel is fp64-VALU / latency bound; spectra are shared by all walkers through L2/MALL, "
                         "so measured HBM traffic is far below the algorithmic bytes (DESIGN.md section 4)")
        # secondary roof: fp64 VALU, with the survey's flop model
        flops = wl.algorithmic_flops_per_eval * W
        step_s = elapsed / args.steps
        roof["valu_fp64"] = dict(bound="valu_fp64", unit="TFLOP/s", peak=FP64_VALU_PEAK_TFLOPS,
                                 achieved=flops / len(wl.pixels) / (tile_ms * 1e-3) / 1e12,
                                 frac=flops / len(wl.pixels) / (tile_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                 algorithmic_flops_per_eval=wl.algorithmic_flops_per_eval,
                                 model="per (line,pixel) 8 + (150 if |x|<8 else 12) + 1; per pixel 30 + 2K + 4")
        host_rate = host_lat = sampler_steps = None
        if not args.no_extras:
            # measured device copy bandwidth next to the vendor peak (read + write of a 1 GiB buffer)
            src = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
            dst = torch.empty_like(src)
            dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            roof["measured_copy_GBps"] = 5 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
            # PCIe-inclusive rate through the host-buffer entry (never `value`): wall time around
            # vp_lnprob_batch including H2D theta + D2H lnprob, 100 calls after 5 warm-ups
            for _ in range(5):
                wl.engine.lnprob(wl.thetas)
            lat = []
            for _ in range(100):
                th0 = time.perf_counter()
                wl.engine.lnprob(wl.thetas)
                lat.append(time.perf_counter() - th0)
            lat = np.sort(np.array(lat))
            host_rate = W / float(np.median(lat))
            host_lat = dict(median_us=1e6 * float(np.median(lat)), p10_us=1e6 * float(lat[10]), p90_us=1e6 * float(lat[90]),
                            calls=100, walkers_per_call=W)

            # the walker loop itself on the GPU (vp_stretch_run): real ensemble steps per second, two
            # half-ensemble passes per step, proposals and accept/reject in HBM -- not part of `value`
            nst = 300
            wl.engine.stretch_run(wl.thetas, 20, seed=1, store_chain=False)
            ts0 = time.perf_counter()
            wl.engine.stretch_run(wl.thetas, nst, seed=1, store_chain=False)
            sampler_steps = nst / (time.perf_counter() - ts0)

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
            "host_entry_latency": host_lat,
            "device_sampler_steps_per_sec": sampler_steps,
            "roofline": roof,
        }
        if use_dist:
            line["gather"] = {"mode": f"async all_gather_into_tensor of {gather_every}-step chunks, double-buffered, "
                                      "overlapped with the following passes",
                              "ms_per_step_blocking_gather": sync_ms, "ms_per_step_without_gather": nogather_ms,
                              "blocking_gather_cost_us": 1e3 * (sync_ms - nogather_ms),
                              "overlapped_gather_cost_us": 1e3 * (line["ms_per_step"] - nogather_ms)}
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_vals = cpu_baseline(wl)
            n = min(len(cpu_vals), W)
            if pool is not None:
                try:
                    cb["fork_pool"] = cpu_pool_baseline(pool, pool_cores, wl)
                except Exception as e:
                    cb["fork_pool"] = {"error": str(e)}
            line["cpu_baseline"] = cb
            line["parity_vs_cpu_baseline_max_rel"] = float(np.max(np.abs(result[:n] / cpu_vals[:n] - 1)))
        print(json.dumps(line), flush=True)
    if pool is not None:
        pool.terminate()
        pool.join()
    if use_dist:
        if rank == 0 and gathered is not None:
            assert torch.equal(gathered[:W], d_out), "all-gathered lnprob differs from the local block"
        dist.barrier()                       # rank 0's extra legs are done: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
