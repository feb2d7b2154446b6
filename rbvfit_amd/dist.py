"""Walker sharding across GPUs: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The path shards naturally: walkers are independent, so every rank evaluates a contiguous block of
theta rows on its own GPU with the same kernels and NO data-path collective; the only exchange
step is one all-gather of the per-walker lnprob vector (W/G doubles per rank -- latency-bound, a
few KB).  Static data (spectra, tables, bounds) is replicated: every rank builds the same engine.

SPMD usage (every rank runs the same sampler loop with the same RNG seed, so every rank holds the
full ensemble and proposes identical moves; only the likelihood evaluation is split):

    post = ShardedPosterior(local_eval)         # local_eval(theta_block (w,D)) -> (w,) lnprob
    lnp = post(theta_all)                        # (W,) on every rank

Island usage (``IslandEnsemble`` / ``PipelinedGather``): every rank owns its block of walkers
outright -- proposals, lnprob and accept/reject need local data only -- so the all-gather of step
k's lnprob (for chain storage and convergence monitoring) is issued asynchronously on RCCL's stream
and overlaps step k+1's kernels; nothing on a rank's critical path waits for a peer.
"""
from __future__ import annotations

import os

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(W: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank ``rank``: ceil(W/world) rows per rank, the last ranks may
    get fewer (or none when W < world -- ragged zeus batches)."""
    per = -(-W // world) if W > 0 else 0
    lo = min(rank * per, W)
    return lo, min(lo + per, W)


class ShardedPosterior:
    """Evaluate lnprob for the full (W, D) batch, each rank doing its block, then all-gather.

    ``local_eval`` maps a host array (w, D) -> (w,) float64 (e.g. ``Engine.lnprob``).  The gather
    runs on ``device`` tensors when given ("cuda" for RCCL) or on CPU tensors (gloo)."""

    def __init__(self, local_eval: Callable[[np.ndarray], np.ndarray], group=None, device: Optional[str] = None):
        import torch.distributed as dist
        self._dist = dist
        self.local_eval = local_eval
        self.group = group
        self.device = device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def __call__(self, theta) -> np.ndarray:
        import torch
        theta = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        W = theta.shape[0]
        lo, hi = shard_bounds(W, self.world, self.rank)
        local = self.local_eval(theta[lo:hi]) if hi > lo else np.empty(0, dtype=np.float64)
        if self.world == 1:
            return np.asarray(local, dtype=np.float64)
        per = -(-W // self.world)
        buf = torch.full((per,), float("nan"), dtype=torch.float64)
        buf[: hi - lo] = torch.from_numpy(np.asarray(local, dtype=np.float64))
        if self.device:
            buf = buf.to(self.device)
        out = torch.empty(per * self.world, dtype=torch.float64, device=buf.device)
        self._dist.all_gather_into_tensor(out, buf, group=self.group)
        out = out.cpu().numpy().reshape(self.world, per)
        parts = [out[r, : shard_bounds(W, self.world, r)[1] - shard_bounds(W, self.world, r)[0]] for r in range(self.world)]
        return np.concatenate(parts)


class DeviceShardedPosterior:
    """Device-resident variant: theta block and lnprob stay in HBM, the engine is driven through
    ``vp_lnprob_batch_device`` and the RCCL all-gather follows on the same stream.  The work runs
    on a dedicated ``torch.cuda.Stream`` (torch's default stream has handle 0, which the C ABI reads
    as "the context's own stream" -- the collective would not be ordered behind the kernels); the
    caller's current stream is made to wait before and after, so it composes with ordinary torch
    code."""

    def __init__(self, engine, theta_block_device, group=None):
        import torch
        import torch.distributed as dist
        self.engine = engine
        self.theta = theta_block_device
        self.W = theta_block_device.shape[0]
        dev = theta_block_device.device
        self.out = torch.empty(self.W, dtype=torch.float64, device=dev)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.gathered = (torch.empty(self.W * self.world, dtype=torch.float64, device=dev)
                         if self.world > 1 else self.out)
        self.stream = torch.cuda.Stream(device=dev)
        self._dist = dist
        self._torch = torch

    def step(self):
        torch = self._torch
        outer = torch.cuda.current_stream(self.stream.device)
        self.stream.wait_stream(outer)                      # theta written on the caller's stream
        with torch.cuda.stream(self.stream):
            self.engine.lnprob_device(self.theta.data_ptr(), self.out.data_ptr(), self.W, self.stream.cuda_stream)
            if self.world > 1:
                self._dist.all_gather_into_tensor(self.gathered, self.out, group=self.group)
        outer.wait_stream(self.stream)
        return self.gathered


class _DevicePointer:
    """A raw device allocation as torch can view it (``torch.as_tensor`` reads ``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None}


class DirectGather:
    """The per-pass exchange of a walker-sharded ensemble WITHOUT a collective launch (``vp_gather_*``): every rank owns a
    (world, W) vector that its peers map through IPC handles; ``step()`` is the pass's own launches, the last of which writes
    this rank's lnprob block into every rank's vector, and the next ``step()``'s first launch raises this rank's flag in every
    peer and waits on the device for its peers' -- the dependency of a blocking all-gather (an ensemble step needs the whole
    ensemble's lnprob) at the cost of 8 bytes per walker and rank of peer stores and one flag per rank.

    The IPC handles travel through ``torch.distributed``'s object all-gather (any backend).  ``DirectGather.probe`` builds
    one, runs two passes and checks them against ``all_gather_into_tensor``; callers fall back to the collective
    (``DeviceShardedPosterior``) when it returns None -- runtimes without IPC between the ranks' devices."""

    def __init__(self, engine, theta_block_device, group=None):
        import torch
        import torch.distributed as dist
        self.engine = engine
        self.theta = theta_block_device
        self.W = int(theta_block_device.shape[0])
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.shared_device = False
        # every rank takes part in both object all-gathers whatever happens on it, and all ranks fail together: a rank that
        # left early would leave its peers inside a collective
        mine, err = None, None
        try:
            mine = engine.gather_create(self.W, self.world, self.rank)
        except Exception as exc:
            err = exc
        if self.world > 1:
            handles = [None] * self.world
            dist.all_gather_object(handles, (mine, engine.device_identity), group=group)
            ids = [h[1] for h in handles]
            handles = [h[0] for h in handles]
            # ranks that share a GPU cannot wait for each other inside full-size launches (vp_gather_connect)
            self.shared_device = len(set(ids)) < len(ids) and os.environ.get("RBVFIT_AMD_GATHER_INKERNEL") != "1"
            if err is None and all(h is not None for h in handles):
                try:
                    engine.gather_connect(b"".join(handles), self.shared_device)
                except Exception as exc:
                    err = exc
            elif err is None:
                err = RuntimeError("a peer could not create its gather buffers")
            oks = [None] * self.world
            dist.all_gather_object(oks, err is None, group=group)      # (also: every rank has mapped its peers before anyone writes)
            if err is None and not all(oks):
                err = RuntimeError("a peer could not map the gather buffers")
        if err is not None:
            raise err
        ptr, _ = engine.gather_state()
        self.gathered = torch.as_tensor(_DevicePointer(ptr, self.W * self.world), device=theta_block_device.device)
        self._torch = torch

    def step(self, stream_ptr: Optional[int] = None, theta=None):
        """Enqueue one pass on ``stream_ptr`` (default: torch's current stream, which must not be the default stream);
        ``theta``: another (W, D) device block than the one the object was made with."""
        if stream_ptr is None:
            stream_ptr = self._torch.cuda.current_stream().cuda_stream
        th = self.theta if theta is None else theta
        self.engine.lnprob_gather_device(th.data_ptr(), self.W, stream_ptr)

    def wait(self, stream_ptr: Optional[int] = None):
        """Enqueue the device-side wait for every rank's block of the last pass (what the next ``step`` does by itself)."""
        if stream_ptr is None:
            stream_ptr = self._torch.cuda.current_stream().cuda_stream
        self.engine.gather_wait(stream_ptr)

    def timed_out(self) -> bool:
        return self.engine.gather_state()[1]

    def close(self):
        self.gathered = None
        self.engine.gather_destroy()

    @classmethod
    def probe(cls, engine, theta_block_device, group=None):
        """A working DirectGather, or None (with the reason in ``DirectGather.last_reason``): two passes are compared with
        the collective's result on every rank and the ranks agree on the outcome."""
        import torch
        import torch.distributed as dist
        ok, dg, reason = 1, None, ""
        try:
            dg = cls(engine, theta_block_device, group)
            ref_local = torch.empty(dg.W, dtype=torch.float64, device=theta_block_device.device)
            s = torch.cuda.current_stream().cuda_stream
            # a pass on other values first, read back here, so that a stale copy of the vector anywhere would show below
            other = theta_block_device.roll(1, dims=0).contiguous()
            dg.step(s, theta=other)
            dg.wait(s)
            stale_bait = float(torch.nan_to_num(dg.gathered).sum().item())
            dg.step(s)
            dg.wait(s)
            engine.lnprob_device(theta_block_device.data_ptr(), ref_local.data_ptr(), dg.W, s)
            if dg.world > 1:
                ref = torch.empty(dg.W * dg.world, dtype=torch.float64, device=ref_local.device)
                dist.all_gather_into_tensor(ref, ref_local, group=group)
            else:
                ref = ref_local
            torch.cuda.synchronize()
            if dg.timed_out():
                ok, reason = 0, "a device-side wait timed out"
            elif not torch.equal(torch.nan_to_num(dg.gathered), torch.nan_to_num(ref)):
                ok, reason = 0, f"the gathered vector differs from the collective's (sum of the pass before: {stale_bait:.6g})"
        except Exception as exc:                           # no IPC, not a one-launch batch, ...
            ok, reason = 0, f"{type(exc).__name__}: {exc}"
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=theta_block_device.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0 and ok:
                ok, reason = 0, "another rank could not set it up"
        cls.last_reason = reason
        if not ok:
            if dg is not None:
                try:
                    dg.close()
                except Exception:
                    pass
            return None
        return dg


class PipelinedGather:
    """Chunked, double-buffered, asynchronous all-gather of a rank-local device vector.

    ``launch(out)`` enqueues the local evaluation that fills ``out`` (a (W,) float64 tensor) on
    the CURRENT torch stream -- e.g. ``Engine.lnprob_device(..., torch.cuda.current_stream().cuda_stream)``
    with a non-default stream made current (handle 0 would select the engine's own stream and the
    collective would not be ordered behind the kernels).  ``step()`` runs it into row k % every of
    the current chunk buffer; when a chunk of ``every`` steps is full, ONE
    ``all_gather_into_tensor(..., async_op=True)`` ships the whole (every, W) block: with RCCL the
    collective runs on the process group's own stream behind an event, so it overlaps the next
    chunk's kernels, and both the per-collective host cost and the latency-bound wire time are
    paid once per ``every`` steps (a few larger collectives instead of many tiny ones).  A chunk
    buffer is reused only after its collective has been waited for (a stream-side wait, not a
    host block).  ``flush()`` ships a partial chunk and waits for everything in flight;
    ``chunk(age)`` is the (world, every, W) result of the ``age``-th most recent SHIPPED chunk."""

    def __init__(self, launch: Callable, W: int, device=None, group=None, every: int = 32, depth: int = 2):
        import torch
        import torch.distributed as dist
        if every < 1 or depth < 2:
            raise ValueError("every >= 1 and depth >= 2 required")
        self._dist, self.group, self.launch = dist, group, launch
        self.W, self.every, self.depth = int(W), int(every), int(depth)
        self.on = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.out = [torch.empty(self.every, self.W, dtype=torch.float64, device=device) for _ in range(depth)]
        self.full = ([torch.empty(self.world, self.every, self.W, dtype=torch.float64, device=device)
                      for _ in range(depth)] if self.on else [o.view(1, self.every, self.W) for o in self.out])
        # row views made once: indexing a tensor costs ~2 us of host time per pass otherwise
        self._row = [[o[j] for j in range(self.every)] for o in self.out] if self.every <= 4096 else None
        self.work = [None] * depth
        self.rows = [0] * depth          # valid rows in each shipped chunk
        self.k = 0                       # steps taken
        self.shipped = 0                 # chunks shipped

    def _wait(self, i):
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None

    def _ship(self, i, rows):
        self.rows[i] = rows
        if self.on:
            self.work[i] = self._dist.all_gather_into_tensor(self.full[i].view(-1), self.out[i].view(-1),
                                                             group=self.group, async_op=True)
        self.shipped += 1

    def step(self):
        c, j = divmod(self.k, self.every)
        i = c % self.depth
        if j == 0:
            self._wait(i)                # the collective that last read this buffer
        row = self._row[i][j] if self._row is not None else self.out[i][j]
        self.launch(row)
        self.k += 1
        if j == self.every - 1:
            self._ship(i, self.every)
        return row

    def flush(self):
        c, j = divmod(self.k, self.every)
        if j:                            # partial chunk: ship it (rows >= j are stale) and start a new one
            self._ship(c % self.depth, j)
            self.k = (c + 1) * self.every
        for i in range(self.depth):
            self._wait(i)

    def chunk(self, age: int = 0):
        """(world, rows, W) view of a shipped chunk; age 0 = the most recent one."""
        if not 0 <= age < min(self.depth, self.shipped):
            raise ValueError("no such chunk in flight (never shipped, or its buffer has been reused)")
        if age == self.depth - 1 and self.k % self.every:
            raise ValueError("that chunk's buffer is being refilled")
        i = (self.shipped - 1 - age) % self.depth
        self._wait(i)
        return self.full[i][:, : self.rows[i]]


class IslandEnsemble:
    """G independent stretch-move ensembles, one per rank, targeting the same posterior.

    Each rank runs ``StretchMoveSampler`` (or, with ``engine=``, the device-resident
    ``DeviceStretchSampler``) over its own ``nwalkers_local`` walkers (seed offset by the rank)
    against its own engine, so an MCMC step needs no exchange at all; ``gather_chain`` then
    collects the per-rank chains with ONE all-gather (one large collective instead of one per
    step).  Statistically this is the usual "many short independent ensembles" scheme: the pooled
    samples are draws from the same posterior, and between-island agreement is a convergence check
    (``island_means``)."""

    def __init__(self, local_lnprob: Optional[Callable], nwalkers_local: int, ndim: int, seed: int = 0, group=None,
                 device: Optional[str] = None, engine=None):
        import torch.distributed as dist
        from .sampler import DeviceStretchSampler, StretchMoveSampler
        self._dist, self.group, self.device = dist, group, device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if engine is not None:           # the rank's whole walker loop on its GPU (vp_stretch_run)
            self.sampler = DeviceStretchSampler(nwalkers_local, ndim, engine, seed=seed * 1000003 + self.rank)
        else:
            self.sampler = StretchMoveSampler(nwalkers_local, ndim, local_lnprob, seed=seed * 1000003 + self.rank)

    def run_mcmc(self, p0_local, nsteps: int):
        return self.sampler.run_mcmc(p0_local, nsteps)

    def gather_chain(self, discard: int = 0):
        """(nsteps-discard, nwalkers_local*world, D) chain and matching lnprob on every rank."""
        import torch
        c = np.ascontiguousarray(self.sampler.chain[discard:])
        lp = np.ascontiguousarray(self.sampler.lnprobability[discard:])
        if self.world == 1:
            return c, lp
        packed = torch.from_numpy(np.concatenate([c.reshape(-1), lp.reshape(-1)]))
        if self.device:
            packed = packed.to(self.device)
        full = torch.empty(packed.numel() * self.world, dtype=torch.float64, device=packed.device)
        self._dist.all_gather_into_tensor(full, packed, group=self.group)
        full = full.cpu().numpy().reshape(self.world, -1)
        cs = [full[r, : c.size].reshape(c.shape) for r in range(self.world)]
        lps = [full[r, c.size:].reshape(lp.shape) for r in range(self.world)]
        return np.concatenate(cs, axis=1), np.concatenate(lps, axis=1)

    def island_means(self, discard: int = 0):
        """(world, D) posterior means per island -- they must agree within the sampling error."""
        c, _ = self.gather_chain(discard)
        n = self.sampler.nwalkers
        return np.stack([c[:, r * n:(r + 1) * n].reshape(-1, c.shape[-1]).mean(axis=0) for r in range(self.world)])
