#!/usr/bin/env python
"""bench.py -- walker-lnprob evals/sec on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config C1|C2|C3|C4] [--strong] [--walkers n]

Works as typed for any N: with N > 1 and no torchrun environment the script starts N ranks itself
(``python -m torch.distributed.run`` as a CHILD process, before anything in this process touches
the GPU) and relays rank 0's JSON line; under torchrun (how the driver starts it) it is a rank.

A "step" is one pass of the hot path over one batch: lnprob of the rank's block of walkers (theta
already resident in HBM, lnprob left in HBM) and, for N > 1, the blocking RCCL all-gather of the
per-walker lnprob vector that a single ensemble needs before its next half-step (north_star's
"RCCL gather of per-walker lnprob"; the *strict* form).  The island form (every rank samples its
own walkers, gathers shipped asynchronously in chunks) and the gather-free time are reported
beside it in ``gather``.

Workload: BASELINE.json configs[1] per GPU by default ("C1": MgII 2796/2803, z=0.348, 2
components, 4096 px, 23-tap Gaussian LSF, 512 walkers per GPU => weak scaling).  ``--config
C2|C3|C4`` selects the other configs (per-GPU share of their walker count: C2 1024, C3 2048/8,
C4 4096/8); ``--strong`` fixes the config's TOTAL walker count and splits it W/N.

Timing: W untimed warm-up steps (raised to what the GPU needs to reach steady clocks; the number
actually run is ``warmup_effective``), then ``repeats`` blocks of EXACTLY K steps, each bracketed
by barrier + synchronize on both sides and max-reduced over ranks; ``ms_per_step`` is the MEDIAN
block (>= ``--min-seconds`` = 2 s of timed passes in total, so a short ``--steps`` is not a one-shot
sample and the driver's GPU-busy sampler sees the run).  ``value`` is the device-resident rate (theta
in HBM when the timed region starts: the bench contract); ``value_host_entry`` is the same batch
through ``vp_lnprob_batch`` with host buffers (the seam north_star names), timed as long.
One JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK_TFLOPS = 78.6   # fp64 vector = half the 157.3 TF fp32 vector rate of the same guide

WORKLOAD_LABEL = {
    "C0": "C0: MgII 2796/2803 z=0.348, 2 components, 4096 px, 23-tap Gaussian LSF",
    "C1": "C1: MgII 2796/2803 z=0.348, 2 components, 4096 px, 23-tap Gaussian LSF",
    "C2": "C2: MgII+FeII+CIV, 8 components (19 lines), 16384 px, 23-tap Gaussian LSF",
    "C3": "C3: joint 2-instrument fit (101-tap tabulated COS-like LSF + 9-tap Gaussian), 8 components (19 lines), 2 x 8192 px",
    "C4": "C4: stress, 4 MgII systems x 8 components (64 lines), 65536 px, 23-tap Gaussian LSF",
}
# total walkers of the config (BASELINE.json) and the GPU count it is quoted on
CONFIG_WALKERS = {"C0": (50, 1), "C1": (512, 1), "C2": (1024, 1), "C3": (2048, 8), "C4": (4096, 8)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="C1", choices=sorted(WORKLOAD_LABEL))
    ap.add_argument("--walkers", type=int, default=None, help="walkers per GPU (default: the config's per-GPU share)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the config's TOTAL walker count (or --walkers x 1) split over the GPUs")
    ap.add_argument("--repeats", type=int, default=0, help="timed blocks of --steps passes (0 = enough for 0.25 s)")
    ap.add_argument("--spread", default="ball", choices=["ball", "posterior"],
                    help="walker positions of the timed passes: 'ball' = SURVEY 8(d)'s 1e-3 ball around theta_true (the default, what "
                         "`value` is defined on), 'posterior' = the ensemble after a burn-in of the device-resident stretch sampler")
    ap.add_argument("--entry", default="lnprob", choices=["lnprob", "model_flux"],
                    help="which C-ABI entry a step is: 'lnprob' (vp_lnprob_batch_device: the headline) or 'model_flux' "
                         "(vp_model_flux_batch_device, the (W, P) convolved model flux: seam 3 of SURVEY 8b, HBM-write-bound)")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="lower bound on the TOTAL timed region (sum of the timed blocks), so that an outside observer sees the GPU busy")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs that are not part of `value` (copy bandwidth, host-entry latency, device sampler): "
                         "used under rocprofv3 so that every tile-kernel launch in the trace is the benchmarked one")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU-only check of the N-rank launcher: gloo rendezvous + all-gather of a synthetic vector, no GPU")
    return ap.parse_args(argv)


# ---- N > 1 typed directly: start the ranks as child processes -----------------------------------
def launch_ranks(args, argv):
    """Parent of an N-rank run.  Nothing here initialises the GPU (device_count() does not), and the
    ranks are fresh child interpreters -- a process that has touched the GPU is never re-exec'ed."""
    if not args.selftest_launcher:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs on this node, found {have}\n")
            return 2
    port = 20000 + (os.getpid() * 7919) % 20000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    last = None
    for ln in proc.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and ln.endswith("}"):
            last = ln
        else:
            sys.stderr.write(ln + "\n")
    rc = proc.wait()
    if last is not None:
        print(last, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited without printing a result line\n")
        rc = 1
    return rc


def selftest_rank():
    """Rank body of --selftest-launcher: the rendezvous, barrier, max-reduce and all-gather the real
    ranks use, over gloo, on a synthetic per-rank vector."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    W = 5
    local = torch.arange(W, dtype=torch.float64) + 1000.0 * rank
    full = torch.empty(W * world, dtype=torch.float64)
    dist.barrier()
    dist.all_gather_into_tensor(full, local)
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = bool(torch.equal(full, torch.cat([torch.arange(W, dtype=torch.float64) + 1000.0 * r for r in range(world)]))
              and t.item() == world - 1)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_ranks": world, "backend": "gloo", "ok": ok}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


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


def cpu_quota():
    """CPUs the container may actually use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            return max(1, int(round(int(q) / int(per))))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        if q > 0:
            return max(1, int(round(q / per)))
    except Exception:
        pass
    return None


def host_cores():
    """Workers of the CPU-baseline legs: every core this process may use -- its affinity mask, cut to the container's CPU quota
    where one is set (a GPU box reports the whole host's cores, 128 on the round-5 boxes, while the job's share is 16: with 128
    workers the fork Pool measured 9.4 k evals/s and the OpenMP leg 12.8 k against 17.8 k / 23 k with 16) -- or BENCH_CPU_CORES;
    at most 128 (the GPU boxes limit the number of processes a job may hold)."""
    n = int(os.environ.get("BENCH_CPU_CORES", "0"))
    if n <= 0:
        n = len(os.sched_getaffinity(0))
        q = cpu_quota()
        if q:
            n = min(n, q)
        else:
            # no readable quota: a GPU box shares its host's cores among its GPUs' jobs, 16 per GPU (counting the devices does not
            # initialise the GPU, so the fork Pool can still be made afterwards)
            try:
                import torch
                n = min(n, 16 * max(1, torch.cuda.device_count()))
            except Exception:
                n = min(n, 16)
    return max(1, min(128, n))


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
    return dict(value=n / dt, cores=cores, cores_reported=os.cpu_count(), cpu_quota=cpu_quota(), kind="port",
                sample=f"{n} lnprob calls through multiprocessing fork Pool({cores}).map over the {wl.name} "
                       f"walker rows ({dt:.1f} s, numpy/scipy oracle)")


def cpu_baseline(wl, budget_s=12.0):
    """Oracle (NumPy/SciPy restatement of the reference path) timed serially on this box's host
    cores: single-theta lnprob over the same walker rows, what emcee does with pool=None."""
    from oracle import voigt_oracle as vo           # checker / baseline only, never the product
    insts = _oracle_instruments(vo, _describe(wl))
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
        cores = host_cores()                              # every core the box gives this process
        sub = wl.thetas[:max(cores, min(len(wl.thetas), 64))]
        if len(sub) < 2 * cores:                          # enough rows for every thread to have work
            sub = np.concatenate([wl.thetas] * (2 * cores // max(1, len(wl.thetas)) + 1))[:2 * cores]
        co.lnprob_batch(sub[:cores], nthreads=cores)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 4.0:
            cvals = co.lnprob_batch(sub, nthreads=cores)
            reps += 1
        dtc = time.perf_counter() - t0
        m = min(len(cvals), len(vals))
        base["c_openmp"] = dict(value=reps * len(sub) / dtc, cores=cores, cores_reported=os.cpu_count(), cpu_quota=cpu_quota(), kind="port",
                                sample=f"{reps} x {len(sub)} walkers, oracle/voigt_oracle.c, OpenMP",
                                max_rel_vs_numpy_oracle=float(np.max(np.abs(cvals[:m] / np.array(vals[:m]) - 1))))
    except Exception as e:                                   # the C oracle is optional for the baseline
        base["c_openmp"] = {"error": str(e)}
    return base, np.array(vals)


def model_flux_main(args, wl, d_theta, stream, rank, world, use_dist, spread_note):
    """--entry model_flux: a step = the convolved model flux of the rank's W walkers on every instrument
    (vp_model_flux_batch_device: CompiledVoigtModel.model_flux, voigt_model.py:295-315, batched; consumers
    unified_results.py:305-369, results_plot.py:383,571), theta resident in HBM, the (W, P) rows left in HBM.  Algorithmic
    bytes per step = W x sum(P) x 8 written (+ the spectral grids read once per walker as in the lnprob roofline)."""
    import torch
    import torch.distributed as dist
    eng = wl.engine
    W, D = wl.thetas.shape
    bufs = [torch.empty((W, P), dtype=torch.float64, device="cuda") for P in wl.pixels]

    def step():
        for k, b in enumerate(bufs):
            eng.model_flux_device(k, d_theta.data_ptr(), b.data_ptr(), W, True, stream.cuda_stream)

    def timed(n):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    t_probe = timed(10) / 10
    warm_eff = max(args.warmup, int(min(1500, max(20, 0.05 / max(t_probe, 1e-6)))))
    for _ in range(warm_eff):
        step()
    repeats = args.repeats if args.repeats > 0 else int(min(50000, max(3, np.ceil(args.min_seconds / max(args.steps * t_probe, 1e-9)))))
    ts = [timed(args.steps) for _ in range(repeats)]
    elapsed = float(np.median(ts))
    if rank != 0:
        return 0
    # live kernel-side timing: HIP events on the launch stream around back-to-back steps, with and without the far-field
    # expansions serving the flux path
    def ev_ms(n=100):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            step()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    step_ms = ev_ms()
    ff_kind = eng.last_farfield_info["variant"]
    variants = {}
    for name, val in (("flux_farfield_off", 0), ("flux_farfield_on", 1)):
        eng.set_option("flux_farfield", val)
        for _ in range(20):
            step()
        variants[name] = ev_ms()
    eng.set_option("flux_farfield", -1)
    bytes_written = 8.0 * W * sum(wl.pixels)
    bytes_algo = bytes_written + W * sum(16 * P for P in wl.pixels) + 8 * D * W       # + wave and 1/wave... read once per walker-eval
    achieved = bytes_written / (step_ms * 1e-3) / 1e9
    roof = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=None,
                kernel=("vp::walker_kernel<0, false, false, false, 1> (the rows in ONE launch; + tile_generic_kernel's small grid behind it)"
                        if (ff_kind == "none" and W * max(1, len(wl.pixels)) <= 512 and len(wl.pixels) == 1) else
                        "vp::tile_kernel<0, 1, false, %s> (+ prep_lines_kernel%s, tile_generic_kernel's small grid)" % (
                            "true" if ff_kind != "none" else "false", ", farfield_kernel" if ff_kind != "none" else "")),
                avg_kernel_ms=step_ms, kernel_timing="HIP events on the launch stream around 100 back-to-back steps: the WHOLE step "
                "(all its launches), not the tile kernel alone -- profiles/ has the rocprofv3 per-kernel means",
                algorithmic_bytes_per_launch=bytes_written, bytes_written_per_step=bytes_written,
                bytes_with_grid_reads_per_step=bytes_algo, farfield=ff_kind,
                ms_per_step_by_flux_farfield=variants,
                note="algorithmic bytes = the (W, P) float64 rows written; the arithmetic in front of the store is the lnprob path's "
                     "(tau -> exp -> LSF), so this entry is bound by the same fp64 VALU work, not by the store")
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
    # host variant: vp_model_flux_batch (theta H2D, rows D2H -- W x P x 8 bytes over PCIe per call)
    nh = 5 if sum(wl.pixels) * W > (1 << 24) else 20
    wl.engine.model_flux(0, wl.thetas[: min(W, 64)])
    th0 = time.perf_counter()
    for _ in range(nh):
        for k in range(len(wl.pixels)):
            got = wl.engine.model_flux(k, wl.thetas)
    host_s = (time.perf_counter() - th0) / nh
    line = {
        "metric": "model_flux walker-evals/sec", "value": W * world * args.steps / elapsed, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_effective": warm_eff, "repeats": repeats,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": WORKLOAD_LABEL[args.config] + " -- entry model_flux (vp_model_flux_batch_device, convolved)",
                   "walkers_per_gpu": W, "walkers_total": W * world, "ndim": D, "n_lines": wl.n_lines, "pixels": wl.pixels,
                   "parallelism": f"walker-shard x{world}"},
        "walker_spread": spread_note,
        "value_host_entry": W / host_s, "host_entry_ms_per_call": 1e3 * host_s,
        "host_entry_note": "vp_model_flux_batch: theta H2D + kernels + (W, P) rows D2H over PCIe (%.1f MB per call)" % (bytes_written / 1e6),
        "roofline": roof,
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle import voigt_oracle as vo           # checker / baseline only, never the product
        insts = _oracle_instruments(vo, _describe(wl))
        n, t0 = 0, time.perf_counter()
        worst = 0.0
        while True:
            i = n % W
            for k, inst in enumerate(insts):
                ref = vo.model_flux(inst.data, wl.thetas[i], inst.wave)
                if n < 8:
                    worst = max(worst, float(np.max(np.abs(ref - (got[i] if k == len(insts) - 1 else wl.engine.model_flux(k, wl.thetas[i])[0])))))
            n += 1
            dt = time.perf_counter() - t0
            if dt > 10.0 or n >= 20 * W:
                break
        line["cpu_baseline"] = dict(value=n / dt, unit="model_flux walker-evals/s", cores=1, kind="port",
                                    sample=f"{n} serial model_flux calls over the {wl.name} walker rows ({dt:.1f} s, numpy/scipy oracle)")
        line["parity_vs_cpu_baseline_max_abs_flux"] = worst
    print(json.dumps(line), flush=True)
    return 0


def rank_main(args):
    # The CPU fan-out leg forks its workers BEFORE anything touches the GPU (a context is not
    # fork-safe and forked children must not hold the device); they idle until the bench is done.
    pool, pool_cores = None, 0
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline
            and not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)):
        import multiprocessing as mp
        pool_cores = host_cores()                      # all host cores the box reports for this process
        pool = mp.get_context("fork").Pool(pool_cores)

    # the library (and the oracle's C restatement) is built -- compilers are child processes -- BEFORE this process touches
    # the GPU: a process that has initialised it must not start other programs on these boxes
    if int(os.environ.get("RANK", "0")) == 0:
        import __graft_entry__ as ge
        ge.build()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}; {torch.cuda.device_count()} visible "
                         "(rbvfit_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1 exercises the RCCL path (init + all_gather) even with a single rank
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
            os.environ.setdefault(k, v)            # only matters for a BENCH_FORCE_DIST=1 run without torchrun
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if use_dist:
        dist.barrier()                             # (rank 0 built the library before it touched the GPU, see above)
    from rbvfit_amd.workloads import make_workload

    # walkers per rank: weak = the config's per-GPU share whatever N; strong = its total split W/N
    tot, quoted_on = CONFIG_WALKERS[args.config]
    if args.strong:
        total = args.walkers if args.walkers is not None else tot
        if total % world:
            raise SystemExit(f"bench.py: --strong needs the walker count ({total}) divisible by --gpus ({world})")
        w_local = total // world
    else:
        w_local = args.walkers if args.walkers is not None else tot // quoted_on
    # every rank owns a different block of walkers of the same ensemble (walker_seed = rank)
    wl = make_workload(args.config, walkers=w_local, device_id=local_rank, walker_seed=1 + rank)
    eng = wl.engine
    W, D = wl.thetas.shape
    # One explicit (non-default) stream carries everything: the engine's kernels are enqueued on its
    # handle and torch / c10d order their work against it as the current stream.  (The default
    # stream's handle is 0, which the C ABI reads as "the context's own stream" -- kernels there would
    # not be ordered with the collectives.)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0

    def burned_in(nsteps=None):
        """The rank's walkers after a burn-in of the device-resident stretch sampler: a posterior-width ensemble."""
        if nsteps is None:
            nsteps = 1500 if args.config in ("C0", "C1") else (200 if args.config in ("C2", "C3") else 60)
        return np.ascontiguousarray(wl.engine.stretch_run(wl.thetas, nsteps, seed=7 + rank, store_chain=False)[0])

    spread_note = "1e-3 ball around theta_true (SURVEY 8d)"
    if args.spread == "posterior":
        wl.thetas = burned_in()
        spread_note = "ensemble after a stretch-move burn-in (posterior width)"
    d_theta = torch.from_numpy(wl.thetas).cuda()
    if args.entry == "model_flux":
        rc = model_flux_main(args, wl, d_theta, stream, rank, world, use_dist, spread_note)
        if pool is not None:
            pool.terminate(); pool.join()
        if use_dist:
            dist.barrier(); dist.destroy_process_group()
        return rc
    d_out = torch.empty(W, dtype=torch.float64, device="cuda")
    gathered = torch.empty(W * world, dtype=torch.float64, device="cuda") if use_dist else None
    torch.cuda.synchronize()

    def launch(out):
        eng.lnprob_device(d_theta.data_ptr(), out.data_ptr(), W, stream.cuda_stream)

    def rccl_step():
        """One pass of a single W_total-walker ensemble: local block, then every rank gets the whole vector."""
        launch(d_out)
        if use_dist:
            dist.all_gather_into_tensor(gathered, d_out)

    # The exchange of the strict form: the lnprob launch itself writes the rank's block into every rank's gathered vector
    # (rbvfit_amd.dist.DirectGather, vp_gather_*: peer-mapped stores + one flag per rank, the next pass waits on the device),
    # where the batch runs as one launch and the runtime shares device memory between the ranks; the blocking RCCL
    # all_gather otherwise (BENCH_GATHER=rccl forces it).  DirectGather.probe checks two passes against the collective.
    direct = None
    direct_reason = "not a multi-rank run"
    if use_dist:
        if os.environ.get("BENCH_GATHER", "direct") == "rccl":
            direct_reason = "BENCH_GATHER=rccl"
        else:
            from rbvfit_amd.dist import DirectGather
            direct = DirectGather.probe(eng, d_theta)
            direct_reason = DirectGather.last_reason

    def strict_step():
        if direct is not None:
            direct.step(stream.cuda_stream)
        else:
            rccl_step()

    def strict_drain():
        if direct is not None:
            direct.wait(stream.cuda_stream)         # (the last pass's blocks have landed everywhere inside the timed region)

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

    def blocks(fn, n, repeats, drain=None):
        ts = [timed(fn, n, drain) for _ in range(repeats)]
        return float(np.median(ts)), ts

    # The GPU needs ~25 ms of sustained load to reach its steady clocks (scripts/warm_probe.py: 37.9 us
    # per step in the first 8 ms after an idle period, 32.9 us from ~25 ms on); a sampler runs for
    # minutes, so the untimed part is made long enough whatever --warmup says (>= 50 ms of passes).
    t_probe = timed(strict_step, 10, strict_drain) / 10
    warm_eff = max(args.warmup, int(min(1500, max(20, 0.05 / max(t_probe, 1e-6)))))
    for _ in range(warm_eff):
        strict_step()
    # timed region: blocks of exactly --steps passes, as many as --min-seconds of passes need (>= 2 s by default: the driver's
    # GPU-busy sampler and its own clock can then corroborate the line)
    repeats = args.repeats if args.repeats > 0 else int(min(50000, max(3, np.ceil(args.min_seconds / max(args.steps * t_probe, 1e-9)))))
    elapsed, block_times = blocks(strict_step, args.steps, repeats, drain=strict_drain)
    rccl_ms = None
    if direct is not None:
        torch.cuda.synchronize()
        if direct.timed_out():
            raise SystemExit("bench.py: a device-side wait of the direct gather timed out")
        launch(d_out)
        torch.cuda.synchronize()
        me = direct.gathered[rank * W:(rank + 1) * W]
        assert torch.equal(torch.nan_to_num(me), torch.nan_to_num(d_out)), "direct gather differs from the local block"
        rccl_ms = 1e3 * blocks(rccl_step, args.steps, min(repeats, 20))[0] / args.steps       # the collective beside it

    # island form (rbvfit_amd.dist): a rank's accept/reject needs its own lnprob only, so the all-gather of a
    # chunk of steps runs on RCCL's stream, double-buffered, under the next chunk's kernels; and the gather-free
    # form -- neither is `value`
    island_ms = nogather_ms = None
    if use_dist:
        from rbvfit_amd.dist import PipelinedGather
        gather_every = int(os.environ.get("BENCH_GATHER_EVERY", "128"))
        pg = PipelinedGather(launch, W, device="cuda", every=gather_every)
        for _ in range(min(warm_eff, 2 * gather_every)):
            pg.step()
        pg.flush()
        island_ms = 1e3 * blocks(pg.step, args.steps, min(repeats, 20), drain=pg.flush)[0] / args.steps
        last = pg.chunk(0)
        assert torch.equal(last[rank, last.shape[1] - 1], d_out), "pipelined gather differs from the local block"
        nogather_ms = 1e3 * blocks(lambda: launch(d_out), args.steps, min(repeats, 20))[0] / args.steps

    # ---- roofline leg: HIP events around the tile kernel on its launch stream (rank 0) -------
    roof = None
    if rank == 0:
        eng.profile_enable(True)
        nprof = min(max(args.steps, 20), 200)
        for _ in range(nprof):
            eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, stream.cuda_stream)
        torch.cuda.synchronize()
        pr = eng.profile_read()
        eng.profile_enable(False)
        kind = eng.last_launch_kind
        kernel_name = {"walker": "vp::walker_kernel<0, false, false, false>", "tiles+farfield": "vp::tile_kernel1<0, true>"}.get(
            kind, "vp::tile_kernel<0, 0, false, false>")
        # Per-launch event pairs put record gaps (and, between dependent kernels, extra serialisation) into every
        # interval -- 8-10 % on a 200-700 us kernel, more on a 30 us one -- so they are only used for the SHARE of the
        # step each kernel kind takes; the step itself is timed by ONE pair of events around nprof back-to-back passes
        # (no per-launch records), and the dominant kernel's duration is that step time x its share / its launches.
        # (seven such blocks, the median: one block of a 20 us pass is 4 ms, short enough for one clock ramp or one
        # neighbour's burst on the box to move it by 20 %)
        step_samples = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(nprof):
                eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            step_samples.append(e0.elapsed_time(e1) / nprof)
        step_ev_ms = float(np.median(step_samples))
        ev_total = pr["prep_ms"] + pr["tile_ms"] + pr.get("finalize_ms", 0.0)
        launches_per_step = max(pr["n_tile_launches"], 1) / nprof
        share = pr["tile_ms"] / ev_total if ev_total > 0 else 1.0
        tile_ms = step_ev_ms * share / launches_per_step
        timing_note = (f"HIP events on the launch stream: one pair around {nprof} back-to-back passes (median of 7 such blocks) gives the step time "
                       f"({1e3 * step_ev_ms:.2f} us); per-launch event pairs give the dominant kernel's share of it ({share:.3f}, "
                       f"{launches_per_step:.0f} launch(es) per step)")
        bytes_per_launch = wl.algorithmic_bytes_per_eval * W / len(wl.pixels)
        achieved = bytes_per_launch / (tile_ms * 1e-3) / 1e9
        # HBM traffic per launch from the committed PMC passes (FETCH_SIZE x2 per the gfx950 guide
        # + WRITE_SIZE, separate rocprofv3 --pmc runs of this same command): profiles/<round>[_<config>]_pmc.json
        traffic, traffic_src = None, None
        try:
            import glob
            for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
                pm = json.load(open(cand))
                if pm.get("config") == args.config and pm.get("walkers_per_gpu") == W:
                    traffic, traffic_src = pm["tile_kernel_hbm_bytes_per_launch"], os.path.basename(cand)
                    kernel_name = pm.get("kernel", kernel_name)       # (the name rocprofv3 gave the dominant kernel of this command)
                    break
        except Exception:
            pass
        valu_issue = None
        try:
            if traffic_src is not None and "sq" in pm and pm["sq"].get("SQ_INSTS_VALU"):
                # measured issue utilisation of the dominant kernel in the committed counter run: wave-level VALU instructions
                # x 4 cycles (what one costs a SIMD at full rate) / (1024 SIMDs x the kernel's duration in that run x 2.4 GHz)
                kus = float(pm["rocprofv3_mean_kernel_us"])
                nv = float(pm["sq"]["SQ_INSTS_VALU"])
                valu_issue = dict(frac=nv * 4.0 / (1024.0 * kus * 1e-6 * 2.4e9), valu_insts_per_launch=nv,
                                  valu_insts_per_eval=nv / W, kernel_us_in_counter_run=kus, cycles_per_inst=4.0, simds=1024,
                                  clock_GHz=2.4, source=traffic_src,
                                  note="the roof that binds this kernel: fp64 VALU issue.  frac = SQ_INSTS_VALU x 4 cycles / "
                                       "(SIMDs x kernel time x clock) from the committed rocprofv3 --pmc run of this command; "
                                       "`achieved` of this object restates the live kernel time as instructions issued per second")
                valu_issue["achieved_Ginst_per_s"] = nv / (tile_ms * 1e-3) / 1e9
                valu_issue["peak_Ginst_per_s"] = 1024 * 2.4 / 4.0
                valu_issue["frac_live"] = valu_issue["achieved_Ginst_per_s"] / valu_issue["peak_Ginst_per_s"]
        except Exception:
            valu_issue = None
        roof = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, traffic_source=traffic_src, kernel=kernel_name, avg_kernel_ms=tile_ms, kernel_timing=timing_note,
                    algorithmic_bytes_per_launch=bytes_per_launch, launches_per_step=pr["n_tile_launches"] / nprof,
                    prep_ms=pr["prep_ms"] / nprof, finalize_ms=pr.get("finalize_ms", 0.0) / nprof,
                    note="kernel is fp64-VALU / latency bound; spectra are shared by all walkers through L2/MALL, "
                         "so measured HBM traffic is far below the algorithmic bytes (DESIGN.md section 4)")
        roof["valu_issue"] = valu_issue
        # secondary roof: fp64 VALU, with the survey's flop model
        flops = wl.algorithmic_flops_per_eval * W
        vt = flops / len(wl.pixels) / (tile_ms * 1e-3) / 1e12
        roof["valu_fp64"] = dict(bound="valu_fp64", unit="TFLOP/s", peak=FP64_VALU_PEAK_TFLOPS, achieved=vt,
                                 frac=vt / FP64_VALU_PEAK_TFLOPS, algorithmic_flops_per_eval=wl.algorithmic_flops_per_eval,
                                 model="per (line,pixel) 8 + (150 if |x|<8 else 12) + 1; per pixel 30 + 2K + 4",
                                 note="ALGORITHMIC flops: every (line, pixel) pair counted as the reference evaluates it; the "
                                      "multipole and far-field expansions execute far fewer (profiles/*_pmc.json has the "
                                      "instruction counts), so this is a throughput statement, not VALU utilisation")
        host_rate = host_lat = sampler_steps = slice_info = None
        if not args.no_extras or os.environ.get("BENCH_HOST_ENTRY") == "1":
            # ---- the same batch through the host-buffer entry (north_star's seam: the sampler ships the (W, D) theta batch through
            # the C ABI and gets (W,) lnprob back): wall time around vp_lnprob_batch calls made back to back, as an ensemble
            # sampler makes them, for >= --min-seconds; theta H2D (zero-copy reads over PCIe), kernels, lnprob D2H and the
            # completion wait are all inside.  Through the Python wrapper (what emcee's vectorize=True sees) and through the
            # bare ctypes function with prebuilt arguments (what a C caller sees).
            import ctypes as C
            th_host = np.ascontiguousarray(wl.thetas)
            for _ in range(20):
                ref_host = wl.engine.lnprob(th_host)
            t_call = 0.0
            for _ in range(20):
                th0 = time.perf_counter(); wl.engine.lnprob(th_host); t_call += (time.perf_counter() - th0) / 20
            ncall = int(max(50, min(200000, np.ceil(args.min_seconds / max(t_call, 1e-7)))))
            th0 = time.perf_counter()
            for _ in range(ncall):
                wl.engine.lnprob(th_host)
            wall = time.perf_counter() - th0
            host_rate = W * ncall / wall
            nlat = min(ncall, 2000)
            lat = np.empty(nlat)
            for i in range(nlat):
                th0 = time.perf_counter(); wl.engine.lnprob(th_host); lat[i] = time.perf_counter() - th0
            lat.sort()
            out_raw = np.empty(W)
            fn, ctx = wl.engine._lib.vp_lnprob_batch, wl.engine._ctx
            pa, pb = C.c_void_p(th_host.ctypes.data), C.c_void_p(out_raw.ctypes.data)
            nraw = min(ncall, 5000)
            th0 = time.perf_counter()
            for _ in range(nraw):
                fn(ctx, W, D, pa, pb)
            raw_us = 1e6 * (time.perf_counter() - th0) / nraw
            assert np.array_equal(np.nan_to_num(out_raw), np.nan_to_num(ref_host)), "raw C-ABI call differs from the wrapper"
            # ... and what the pre-armed launch of the next call contributes (vp_lnprob_batch, option "prearm"): the same loop with
            # it switched off, and both forms as a caller with 100 us of its own work between calls sees them
            def paced(n, gap_s):
                tot = 0.0
                for _ in range(n):
                    t1 = time.perf_counter()
                    while time.perf_counter() - t1 < gap_s:
                        pass
                    t1 = time.perf_counter(); wl.engine.lnprob(th_host); tot += time.perf_counter() - t1
                return 1e6 * tot / n
            prearm_counts = dict(wl.engine.prearm_counts)
            paced_on = paced(2000, 100e-6)
            wl.engine.set_option("prearm", 0)
            for _ in range(20):
                wl.engine.lnprob(th_host)
            noff = min(ncall, 20000)
            th0 = time.perf_counter()
            for _ in range(noff):
                wl.engine.lnprob(th_host)
            off_us = 1e6 * (time.perf_counter() - th0) / noff
            paced_off = paced(2000, 100e-6)
            wl.engine.set_option("prearm", -1)
            host_lat = dict(us_per_call=1e6 * wall / ncall, calls=ncall, seconds=wall, walkers_per_call=W,
                            prearm=dict(counts_after_timed_loop=prearm_counts, us_per_call_without=off_us,
                                        us_per_call_caller_with_100us_between_calls=paced_on,
                                        us_per_call_caller_with_100us_between_calls_without=paced_off,
                                        note="vp_lnprob_batch leaves the NEXT call's launch waiting on the GPU for its theta when calls "
                                             "follow each other within prearm_us / 2 (include/rbvfit_amd.h); 'without' = option prearm 0"),
                            median_us=1e6 * float(np.median(lat)), p10_us=1e6 * float(lat[nlat // 10]),
                            p90_us=1e6 * float(lat[(9 * nlat) // 10]), latency_sample=nlat,
                            us_per_call_bare_cabi=raw_us, python_wrapper_us=1e6 * wall / ncall - raw_us,
                            us_per_pass_device_resident=1e3 * step_ev_ms,
                            note="back-to-back vp_lnprob_batch calls with host theta / host lnprob (Engine.lnprob); bare_cabi = the ctypes "
                                 "function with prebuilt arguments; device_resident = the same batch with theta and lnprob left in HBM")
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

            # the walker loop itself on the GPU (vp_stretch_run): real ensemble steps per second, two
            # half-ensemble passes per step, proposals and accept/reject in HBM -- not part of `value`
            if W % 2 == 0:
                nst = 3000 if args.config in ("C0", "C1") else (200 if args.config in ("C2", "C3") else 60)
                wl.engine.stretch_run(wl.thetas, max(2, nst // 15), seed=1, store_chain=False)
                ts0 = time.perf_counter()
                wl.engine.stretch_run(wl.thetas, nst, seed=1, store_chain=False)
                sampler_steps = nst / (time.perf_counter() - ts0)
                # and the reference's second sampler (zeus' ensemble slice sampling) with the walker loop on the GPU
                if W <= 4096:
                    nsl = 60 if args.config in ("C0", "C1") else 6
                    r0 = wl.engine.slice_run(wl.thetas, max(2, nsl // 6), seed=1, store_chain=False)
                    ts0 = time.perf_counter()
                    r1 = wl.engine.slice_run(r0["pos"], nsl, lnprob=r0["lnprob"], seed=1, step0=max(2, nsl // 6), mu=r0["mu"],
                                             tune=r0["tune"], store_chain=False)
                    dts = time.perf_counter() - ts0
                    slice_info = dict(steps_per_sec=nsl / dts, lnprob_evals_per_walker_step=r1["n_evals"] / (nsl * W),
                                      evals_per_sec=r1["n_evals"] / dts, mu=r1["mu"])
                    if args.config in ("C0", "C1"):
                        # BASELINE config 4 names zeus walkers on the joint two-instrument fit: the same sampler on C3 at its
                        # whole ensemble of 2048 walkers (one GPU), a few iterations
                        try:
                            from rbvfit_amd.workloads import make_workload as _mk
                            w3 = _mk("C3", walkers=2048)
                            q0 = w3.engine.slice_run(w3.thetas, 2, seed=1, store_chain=False)
                            ts0 = time.perf_counter()
                            q1 = w3.engine.slice_run(q0["pos"], 6, lnprob=q0["lnprob"], seed=1, step0=2, mu=q0["mu"], tune=q0["tune"],
                                                     store_chain=False)
                            dt3 = time.perf_counter() - ts0
                            slice_info["c3_2048_walkers"] = dict(steps_per_sec=6 / dt3, lnprob_evals_per_walker_step=q1["n_evals"] / (6 * 2048),
                                                                 evals_per_sec=q1["n_evals"] / dt3)
                            w3.engine.close()
                        except Exception as exc:                   # (an extra: never fails the line)
                            slice_info["c3_2048_walkers"] = {"error": str(exc)}

        spread_info = multi_info = None
        if not args.no_extras and W % 2 == 0 and args.spread == "ball":
            # the same passes on a posterior-width ensemble (walkers' line cores no longer in the same tiles)
            try:
                th_post = burned_in()
                d_post = torch.from_numpy(th_post).cuda()
                torch.cuda.synchronize()
                for _ in range(50):
                    eng.lnprob_device(d_post.data_ptr(), d_out.data_ptr(), W, stream.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                npass = nprof
                e0.record(stream)
                for _ in range(npass):
                    eng.lnprob_device(d_post.data_ptr(), d_out.data_ptr(), W, stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize()
                ms_post = e0.elapsed_time(e1) / npass
                sd = th_post.std(axis=0)
                spread_info = dict(evals_per_sec=W / (ms_post * 1e-3), ms_per_step=ms_post, ms_per_step_ball=step_ev_ms,
                                   theta_std_min=float(sd.min()), theta_std_max=float(sd.max()),
                                   note="lnprob passes on the ensemble a stretch-move burn-in leaves (posterior width) next to the "
                                        "1e-3 ball `value` is defined on; HIP events around back-to-back passes")
                eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, stream.cuda_stream)     # d_out = the ball's result again
                torch.cuda.synchronize()
            except Exception as e:
                spread_info = {"error": str(e)}
            # one ensemble sharded over TWO device contexts on this one GPU (vp_multi_stretch_run, device listed twice): what the
            # exchange of moved rows + the event barrier per half-step cost next to the single-context sampler
            try:
                import rbvfit_amd
                nst = 300 if args.config in ("C0", "C1") else 20
                with rbvfit_amd.MultiEngine([local_rank, local_rank]) as me:
                    me.set_bounds(wl.lb, wl.ub)
                    for data, (wave, flux, err) in zip(wl.tables, wl.spectra):
                        wgt = 1.0 / err ** 2
                        me.add_instrument(wave, flux, wgt, np.log(wgt), **data.engine_kwargs())
                    me.stretch_run(wl.thetas, max(2, nst // 15), seed=1, store_chain=False)
                    ts0 = time.perf_counter()
                    me.stretch_run(wl.thetas, nst, seed=1, store_chain=False)
                    multi_steps = nst / (time.perf_counter() - ts0)
                multi_info = dict(steps_per_sec_two_contexts_one_gpu=multi_steps, steps_per_sec_one_context=sampler_steps,
                                  us_per_half_step_two_contexts=0.5e6 / multi_steps,
                                  us_per_half_step_one_context=(0.5e6 / sampler_steps) if sampler_steps else None,
                                  note="vp_multi_stretch_run with this GPU listed twice: two half-size kernels side by side, moved rows "
                                       "written into both replicas, 2 event records + 2 stream waits per half-step, against "
                                       "vp_stretch_run; both chains are identical bit for bit")
            except Exception as e:
                multi_info = {"error": str(e)}

    side_info = None
    if rank == 0 and not args.no_extras and args.config in ("C0", "C1"):
        # Two independent ensembles (two contexts, two streams) evaluated side by side: what one GPU sustains for a caller with more
        # than one chain in flight (tempered chains, several absorbers fitted at once).  A lone stream of dependent passes leaves
        # the SIMDs idle through every launch's entry; a second stream's passes move into those gaps.  Not `value`.
        try:
            from rbvfit_amd.workloads import make_workload as _mk
            wl2 = _mk(args.config, walkers=W, device_id=local_rank, walker_seed=101 + rank)
            th2 = torch.from_numpy(wl2.thetas).cuda()
            out2 = torch.empty(W, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()

            # (stream handle 0 = each context's OWN stream, made by the library: those run side by side.  Two streams of torch's
            #  pool did not on this stack -- 45.4 us per pair of 512-walker passes against 39.7, 31.5 against 20.2 at 256,
            #  scripts/side_by_side.py -- they seem to share a hardware queue)
            def both():
                eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, 0)
                wl2.engine.lnprob_device(th2.data_ptr(), out2.data_ptr(), W, 0)
            for _ in range(200):
                both()
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                t0 = time.perf_counter()
                for _ in range(400):
                    both()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 400)
            pair_s = float(np.median(ts))
            chk = wl2.engine.lnprob(wl2.thetas)
            assert np.array_equal(out2.cpu().numpy(), chk, equal_nan=True), "side-by-side pass differs from the context's own"
            side_info = dict(evals_per_sec=2 * W / pair_s, us_per_pair_of_passes=1e6 * pair_s, us_per_pass_one_stream=1e3 * step_ev_ms,
                             note="two contexts of W walkers each on their own streams, passes enqueued alternately; wall time over 400 pairs, "
                                  "median of 7; the second context's result checked against its own lone pass")
            wl2.engine.close()
            eng.lnprob_device(d_theta.data_ptr(), d_out.data_ptr(), W, stream.cuda_stream)
            torch.cuda.synchronize()
        except Exception as e:                                   # (an extra: never fails the line)
            side_info = {"error": str(e)}

    result = d_out.cpu().numpy()
    if rank == 0:
        evals = W * world * args.steps
        line = {
            "metric": "walker-lnprob evals/sec", "value": evals / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_effective": warm_eff,
            "repeats": repeats, "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_min_block": 1e3 * min(block_times) / args.steps,
            "ms_per_step_max_block": 1e3 * max(block_times) / args.steps,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": WORKLOAD_LABEL[args.config],
                       "walkers_per_gpu": W, "walkers_total": W * world, "ndim": D, "n_lines": wl.n_lines,
                       "pixels": wl.pixels,
                       "parallelism": f"walker-shard x{world}" + ((" + direct-write gather per pass" if direct is not None else
                                                                   " + blocking RCCL all_gather per pass") if world > 1 else "")},
            "passes_per_sec": args.steps / elapsed,
            "mcmc_steps_per_sec": sampler_steps,
            "mcmc_steps_per_sec_note": "device-resident stretch move (vp_stretch_run): one ensemble step = two half-ensemble "
                                       "launches (proposal, lnprob, accept / reject inside each walker's workgroup; consecutive half-steps "
                                       "overlap on two streams, every walker waiting for its partner's row only); rank 0's walkers; wall "
                                       "time around the call, upload of the start positions and download of the final state included",
            "slice_sampler": slice_info,
            "slice_sampler_note": "device-resident ensemble slice sampling (vp_slice_run, zeus' differential move): ensemble steps/s, "
                                  "lnprob evaluations per walker and step, and the evaluations/s they amount to",
            "value_host_entry": host_rate,
            "value_definition": "`value`: theta resident in HBM when the timed region starts and lnprob left in HBM (vp_lnprob_batch_device; the "
                                "bench contract's definition, PCIe-inclusive rates are never `value`).  `value_host_entry`: the same batch through "
                                "vp_lnprob_batch with host buffers -- what north_star's sampler loop and SURVEY 8(d)'s 'wall time around "
                                "vp_lnprob_batch including H2D/D2H' describe (emcee vectorize=True, the INTEGRATION.md stub) --, timed over as "
                                "many seconds, calls back to back as a sampler makes them (so the library's pre-armed launch of the next call "
                                "is in play: host_entry_latency.prearm has the same loop without it and a caller with 100 us between calls); "
                                "both are first-class numbers of this line",
            "seam_evals_per_sec": host_rate,
            "host_entry_evals_per_sec_pcie_inclusive": host_rate,
            "host_entry_latency": host_lat,
            "walker_spread": spread_note,
            "posterior_spread": spread_info if not args.no_extras else None,
            "sharded_sampler_one_gpu": multi_info if not args.no_extras else None,
            "two_ensembles_side_by_side": side_info,
            "roofline": roof,
        }
        if use_dist:
            line["gather"] = {"value_form": ("strict: every pass's lnprob block written by the launch itself into every rank's gathered "
                                             "vector (peer-mapped stores, one flag per rank; the next pass waits for it on the device): "
                                             "rbvfit_amd.dist.DirectGather / vp_gather_*") if direct is not None else
                                            "strict: blocking all_gather_into_tensor of the per-walker lnprob after every pass",
                              "direct_gather": direct is not None, "direct_gather_unavailable_because": direct_reason or None,
                              "direct_gather_probe": direct_reason or "clean: two passes matched all_gather_into_tensor on every rank",
                              "ranks_seen_by_rccl": dist.get_world_size(), "backend": dist.get_backend(),
                              "visible_gpus_on_rank0": torch.cuda.device_count(),
                              "ms_per_step_blocking_rccl_all_gather": rccl_ms if rccl_ms is not None else line["ms_per_step"],
                              "ms_per_step_island_form": island_ms,
                              "island_form": f"async all_gather_into_tensor of {gather_every}-step chunks, double-buffered, "
                                             "overlapped with the following passes (rbvfit_amd.dist.PipelinedGather)",
                              "island_form_evals_per_sec": W * world / (island_ms * 1e-3),
                              "ms_per_step_without_gather": nogather_ms,
                              "blocking_gather_cost_us": 1e3 * (line["ms_per_step"] - nogather_ms),
                              "overlapped_gather_cost_us": 1e3 * (island_ms - nogather_ms)}
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
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        return launch_ranks(args, argv)
    if args.selftest_launcher:
        if not under_launcher:
            return launch_ranks(args, argv)        # also exercises the launcher with one rank
        return selftest_rank()
    return rank_main(args)


if __name__ == "__main__":
    sys.exit(main())
