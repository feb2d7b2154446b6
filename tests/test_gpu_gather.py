"""Direct-write gather (vp_gather_*, rbvfit_amd.dist.DirectGather): the per-pass exchange of a walker-sharded ensemble
written by the lnprob launch itself into every rank's gathered vector.  One GPU is enough to exercise the whole mechanism:
two PROCESSES on device 0 export / map each other's buffers through hipIpc handles (carried by a gloo process group), write
into them from their kernels and wait for each other's flags on the device."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_gather_is_the_plain_pass():
    torch = pytest.importorskip("torch")
    from rbvfit_amd.dist import DirectGather
    from rbvfit_amd.workloads import make_workload
    wl = make_workload("C1", walkers=96)
    th = wl.thetas.copy()
    th[5, 2] = 1e9                                        # an out-of-bounds walker: -inf through the same path
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        d_theta = torch.from_numpy(th).cuda()
        ref = torch.empty(96, dtype=torch.float64, device="cuda")
        dg = DirectGather(wl.engine, d_theta)
        for _ in range(3):
            dg.step()
        dg.wait()
        wl.engine.lnprob_device(d_theta.data_ptr(), ref.data_ptr(), 96, stream.cuda_stream)
        torch.cuda.synchronize()
        assert not dg.timed_out()
        assert torch.equal(dg.gathered, ref) and torch.isinf(ref[5])
        assert wl.engine.last_launch_kind == "walker"
        dg.close()
    # a batch that takes several launches (records, far-field expansions, tiles, final reduction): the first one handshakes,
    # the last one writes into the gathered vectors
    wl2 = make_workload("C2", walkers=64)
    th2 = wl2.thetas.copy()
    th2[7, 0] = -1e9
    with torch.cuda.stream(stream):
        d2 = torch.from_numpy(th2).cuda()
        ref2 = torch.empty(64, dtype=torch.float64, device="cuda")
        dg2 = DirectGather.probe(wl2.engine, d2)
        assert dg2 is not None, DirectGather.last_reason
        for _ in range(3):
            dg2.step()
        dg2.wait()
        assert wl2.engine.last_launch_kind.startswith("tiles")
        wl2.engine.lnprob_device(d2.data_ptr(), ref2.data_ptr(), 64, stream.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(dg2.gathered, ref2) and torch.isinf(ref2[7]) and not dg2.timed_out()
        dg2.close()
    wl.engine.close(); wl2.engine.close()


def test_gather_argument_and_state_errors():
    torch = pytest.importorskip("torch")
    from rbvfit_amd.workloads import make_workload
    wl = make_workload("C1", walkers=32)
    eng = wl.engine
    d_theta = torch.from_numpy(wl.thetas).cuda()
    with pytest.raises(RuntimeError, match="no connected gather"):
        eng.lnprob_gather_device(d_theta.data_ptr(), 32, 0)
    with pytest.raises(RuntimeError, match="world"):
        eng.gather_create(32, 9, 0)                       # more ranks than a vector has flags for
    with pytest.raises(RuntimeError, match="rank"):
        eng.gather_create(32, 2, 2)
    eng.gather_create(32, 2, 0)                           # two ranks: usable only once the peer's handles are mapped
    with pytest.raises(RuntimeError, match="no connected gather"):
        eng.lnprob_gather_device(d_theta.data_ptr(), 32, 0)
    eng.gather_destroy()
    eng.gather_create(32, 1, 0)
    with pytest.raises(RuntimeError, match="block size"):
        eng.lnprob_gather_device(d_theta.data_ptr(), 16, 0)
    eng.lnprob_gather_device(d_theta.data_ptr(), 32, 0)
    ptr, timed_out = eng.gather_state()
    assert ptr != 0 and not timed_out
    eng.gather_destroy()
    eng.close()


def _rank(rank, world, port, q, config="C1", inkernel=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if inkernel:
        os.environ["RBVFIT_AMD_GATHER_INKERNEL"] = "1"     # (small batches: both ranks' launches fit on the GPU together)
    import torch
    import torch.distributed as dist
    from rbvfit_amd.dist import DirectGather
    from rbvfit_amd.workloads import make_workload
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    W = 64
    wl = make_workload(config, walkers=W, walker_seed=1 + rank)    # every rank its own block of walkers
    stream = torch.cuda.Stream()
    out = {}
    with torch.cuda.stream(stream):
        d_theta = torch.from_numpy(wl.thetas).cuda()
        local = torch.empty(W, dtype=torch.float64, device="cuda")
        dg = DirectGather.probe(wl.engine, d_theta)
        out["reason"] = DirectGather.last_reason
        out["shared"] = dg.shared_device if dg is not None else None
        if dg is not None:
            for _ in range(50):                                    # back-to-back passes: each waits for the peer's last block
                dg.step()
            dg.wait()
            wl.engine.lnprob_device(d_theta.data_ptr(), local.data_ptr(), W, stream.cuda_stream)
            torch.cuda.synchronize()
            out["timed_out"] = dg.timed_out()
            out["gathered"] = dg.gathered.cpu().numpy().copy()
            out["local"] = local.cpu().numpy().copy()
    dist.barrier()
    if dg is not None:
        dg.close()
    wl.engine.close()
    dist.destroy_process_group()
    q.put((rank, out))


@pytest.mark.parametrize("world,config", [(8, "C1"), (5, "C1"), (8, "C2")])
def test_eight_ranks_in_one_process_gather_into_each_other(world, config):
    """The exchange at the size of the node the scaling bench runs on: 8 ranks -- 8 flags per vector, 8
    blocks, the handshake raising 7 peers' flags -- as 8 contexts of THIS process on device 0 (the GPU boxes allow six processes
    on a card, and a process cannot open its own IPC handles: vp_gather_connect_local takes the peers' device pointers
    instead).  Every rank evaluates its own block of 64 walkers with launches on its own stream; after 30 back-to-back passes,
    each waiting on the device for all peers' blocks of the pass before, every rank's gathered vector holds all 8 blocks bit
    for bit.  C1: one launch per pass (walker_kernel writes into the vectors); C2: several (the finalize launch does)."""
    torch = pytest.importorskip("torch")
    from rbvfit_amd.workloads import make_workload
    W = 64 if config == "C1" else 24
    wls = [make_workload(config, walkers=W, walker_seed=1 + r) for r in range(world)]
    engs = [w.engine for w in wls]
    try:
        for r, e in enumerate(engs):
            e.gather_create(W, world, r)
        for e in engs:
            e.gather_connect_local(engs, shared_device=True)          # ranks share the GPU: the handshake is a launch of its own
        streams = [torch.cuda.Stream() for _ in engs]
        thetas = [torch.from_numpy(w.thetas).cuda() for w in wls]
        torch.cuda.synchronize()
        for k in range(30):
            for r, e in enumerate(engs):
                e.lnprob_gather_device(thetas[r].data_ptr(), W, streams[r].cuda_stream)
        for r, e in enumerate(engs):
            e.gather_wait(streams[r].cuda_stream)
        torch.cuda.synchronize()
        local = [e.lnprob(w.thetas) for e, w in zip(engs, wls)]
        want = np.concatenate(local)
        assert len({a.tobytes() for a in local}) == world              # every rank its own walkers
        for r, e in enumerate(engs):
            ptr, timed_out = e.gather_state()
            assert not timed_out
            from rbvfit_amd.dist import _DevicePointer
            got = torch.as_tensor(_DevicePointer(ptr, W * world), device="cuda").cpu().numpy()
            np.testing.assert_array_equal(got, want)
        with pytest.raises(RuntimeError, match="peers"):
            fresh = make_workload("C1", walkers=W)
            try:
                fresh.engine.gather_create(W, world, 0)
                fresh.engine.gather_connect_local([fresh.engine] + engs[1:-1] + [fresh.engine], shared_device=True)   # last entry: not rank world-1
            finally:
                fresh.engine.close()
    finally:
        for e in engs:
            e.gather_destroy()
        for e in engs:
            e.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("config,inkernel,world", [("C1", False, 2), ("C1", True, 2), ("C2", False, 2), ("C2", True, 2), ("C1", False, 4)])
def test_two_processes_on_one_gpu_gather_into_each_other(config, inkernel, world):
    """config: one launch per pass / several; inkernel: the handshake inside the pass's first launch (what ranks on GPUs of their
    own get) or -- ranks that share a GPU are told apart by their device identity -- a one-wave launch in front of it."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 777 + (13 if config == "C2" else 0) + (29 if inkernel else 0) + 41 * world) % 2000
    procs = [ctx.Process(target=_rank, args=(r, world, port, q, config, inkernel)) for r in range(world)]      # (with this process: <= 6 on the card)
    for p in procs:
        p.start()
    res = dict(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(res[r]["reason"] == "" for r in range(world)), [res[r]["reason"] for r in range(world)]
    want = np.concatenate([res[r]["local"] for r in range(world)])
    for r in range(world):
        assert res[r]["shared"] == (not inkernel)
        assert not res[r]["timed_out"]
        np.testing.assert_array_equal(res[r]["gathered"], want)    # every rank holds all blocks, bit for bit
    assert not np.array_equal(res[0]["local"], res[1]["local"])
