"""N > 1 path on CPU: two gloo ranks shard the walkers, evaluate their blocks (here with the CPU
oracle standing in for the per-rank GPU engine -- this test is about the sharding/gather logic) and
all-gather the per-walker lnprob."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    for W in (10, 7, 1):                       # even, ragged, fewer walkers than ranks
        res[W] = post(z["thetas"][:W])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, res, calls))


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding_matches_serial():
    import torch.multiprocessing as mp
    from oracle import voigt_oracle as vo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    z = np.load(os.path.join(ROOT, "tests", "golden", "ragged_1000.npz"))
    insts = vo.instruments_from_fixture(z)
    for W in (10, 7, 1):
        ref = vo.lnprob_batch(z["thetas"][:W], z["lb"], z["ub"], insts)
        for rank, res, calls in out:
            assert np.array_equal(res[W], ref, equal_nan=True)      # every rank holds the full vector
    calls = dict((r, c) for r, _, c in out)
    assert calls[0] == [5, 4, 1] and calls[1] == [5, 3]             # blocks: ceil(W/2) rows, rank 1 idle at W=1
