"""N > 1 path on CPU: two gloo ranks shard the walkers, evaluate their blocks (here with the CPU
oracle standing in for the per-rank GPU engine -- this test is about the sharding/gather logic) and
all-gather the per-walker lnprob."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


WS = (16, 10, 7, 3, 1)


def _rows(z, W):
    return np.concatenate([z["thetas"], z["thetas"][::-1] + 1e-4])[:W]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import voigt_oracle as vo
    from rbvfit_amd.dist import ShardedPosterior, shard_bounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(ROOT, "tests", "golden", "ragged_1000.npz"))
    insts = vo.instruments_from_fixture(z)
    calls = []

    def local_eval(block):
        calls.append(len(block))
        return vo.lnprob_batch(block, z["lb"], z["ub"], insts)

    post = ShardedPosterior(local_eval)
    res = {}
    for W in WS:                               # even, ragged, fewer walkers than ranks
        res[W] = post(_rows(z, W))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, res, calls))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_gloo_sharding_matches_serial(world):
    """SURVEY 8e on CPU for world = 2, 4 and 8 (the node the scaling bench runs on): contiguous blocks of ceil(W / G) rows,
    ranks past the end idle, every rank ends up with the whole vector -- also for batches of fewer rows than ranks."""
    import torch.multiprocessing as mp
    from oracle import voigt_oracle as vo
    from rbvfit_amd.dist import shard_bounds
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 131) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    z = np.load(os.path.join(ROOT, "tests", "golden", "ragged_1000.npz"))
    insts = vo.instruments_from_fixture(z)
    for W in WS:
        ref = vo.lnprob_batch(_rows(z, W), z["lb"], z["ub"], insts)
        for rank, res, calls in out:
            assert np.array_equal(res[W], ref, equal_nan=True)      # every rank holds the full vector
    calls = dict((r, c) for r, _, c in out)
    for r in range(world):
        want = []
        for W in WS:
            lo, hi = shard_bounds(W, world, r)
            if hi > lo:
                want.append(hi - lo)
        assert calls[r] == want, (r, calls[r], want)                # blocks of ceil(W / world) rows; idle ranks evaluate nothing
        for W in WS:
            lo, hi = shard_bounds(W, world, r)
            B = -(-W // world)
            assert (lo, hi) == (min(r * B, W), min((r + 1) * B, W))
    if world == 2:
        assert calls[0] == [8, 5, 4, 2, 1] and calls[1] == [8, 5, 3, 1]


def _island_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import voigt_oracle as vo
    from rbvfit_amd.dist import IslandEnsemble, PipelinedGather
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(ROOT, "tests", "golden", "tiny_7px.npz"))
    insts = vo.instruments_from_fixture(z)
    thetas = z["thetas"]
    W = 3
    step_no = [0]

    # --- pipelined all-gather: step k's block differs per rank and per step
    def launch(out):
        lo = (step_no[0] * world + rank) * W % (len(thetas) - W)
        out.copy_(torch.from_numpy(vo.lnprob_batch(thetas[lo:lo + W], z["lb"], z["ub"], insts)))
        step_no[0] += 1

    pg = PipelinedGather(launch, W, every=2)
    got = []
    for k in range(5):
        pg.step()
        if k % 2 == 1:
            got.append(pg.chunk(0).clone().numpy())              # (world, 2, W): steps k-1, k
    with pytest.raises(ValueError):
        pg.chunk(1)                                              # its buffer is being refilled by step 4
    pg.flush()                                                   # ships the partial chunk (step 4 only)
    got.append(pg.chunk(0).clone().numpy())
    assert got[-1].shape == (world, 1, W)
    got.append(pg.chunk(1).clone().numpy())                      # after flush the previous chunk is readable again
    with pytest.raises(ValueError):
        pg.chunk(2)

    # --- island ensemble: independent per-rank samplers, one bulk gather of the chains
    D = thetas.shape[1]
    nw = 2 * D + 2
    rng = np.random.default_rng(100 + rank)
    lb, ub = z["lb"], z["ub"]
    mid = thetas[0]
    p0 = np.clip(mid + 1e-4 * rng.standard_normal((nw, D)), lb + 1e-10, ub - 1e-10)
    isl = IslandEnsemble(lambda th: vo.lnprob_batch(th, lb, ub, insts), nw, D, seed=7)
    isl.run_mcmc(p0, 6)
    chain, lnp = isl.gather_chain()
    means = isl.island_means()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, got, chain, lnp, means, isl.sampler.chain, isl.sampler.lnprobability))


@pytest.mark.timeout(300)
def test_two_rank_gloo_pipelined_gather_and_islands():
    import torch.multiprocessing as mp
    from oracle import voigt_oracle as vo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_island_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    z = np.load(os.path.join(ROOT, "tests", "golden", "tiny_7px.npz"))
    insts = vo.instruments_from_fixture(z)
    thetas, W, world = z["thetas"], 3, 2
    def ref_step(k):
        return np.stack([vo.lnprob_batch(thetas[(k * world + r) * W % (len(thetas) - W):][:W], z["lb"], z["ub"], insts)
                         for r in range(world)])                 # (world, W)
    for rank, got, *_ in out:
        assert np.array_equal(got[0], np.stack([ref_step(0), ref_step(1)], axis=1), equal_nan=True)
        assert np.array_equal(got[1], np.stack([ref_step(2), ref_step(3)], axis=1), equal_nan=True)
        assert np.array_equal(got[2], ref_step(4)[:, None, :], equal_nan=True)
        assert np.array_equal(got[3], got[1], equal_nan=True)
    # every rank holds the same pooled chain = concatenation of the two islands' own chains
    (_, _, c0, l0, m0, own0, ownl0), (_, _, c1, l1, m1, own1, ownl1) = out
    assert np.array_equal(c0, c1) and np.array_equal(l0, l1) and np.array_equal(m0, m1)
    assert np.array_equal(c0, np.concatenate([own0, own1], axis=1))
    assert np.array_equal(l0, np.concatenate([ownl0, ownl1], axis=1))
    assert not np.array_equal(own0, own1)                       # different seeds per island
    assert m0.shape == (2, thetas.shape[1])


# ---- bench.py's own N-rank launcher (python bench.py --gpus N typed directly) ------------------
@pytest.mark.timeout(300)
def test_bench_launcher_starts_its_own_ranks_over_gloo():
    """`bench.py --gpus 2 --selftest-launcher`: the parent spawns 2 ranks through
    torch.distributed.run (fresh child interpreters), they rendezvous on 127.0.0.1, run the
    barrier / all-gather / max-reduce of the timed loop over gloo, and rank 0's JSON line is relayed."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"selftest": "launcher", "n_ranks": 2, "backend": "gloo", "ok": True}


@pytest.mark.timeout(120)
def test_bench_without_enough_gpus_fails_cleanly():
    """On a box with fewer GPUs than --gpus the launcher says so (exit 2), before starting anything."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has 2 GPUs")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5"],
                       capture_output=True, text=True, env=env, timeout=100)
    assert r.returncode == 2
    assert "needs 2 GPUs" in r.stderr and r.stdout.strip() == ""
