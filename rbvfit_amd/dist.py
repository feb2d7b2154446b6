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
"""
from __future__ import annotations

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
    """Device-resident variant used by bench.py: theta block and lnprob stay in HBM, the engine is
    driven through ``vp_lnprob_batch_device`` on torch's current stream and the RCCL all-gather
    follows on the same stream."""

    def __init__(self, engine, theta_block_device, group=None):
        import torch
        import torch.distributed as dist
        self.engine = engine
        self.theta = theta_block_device
        self.W = theta_block_device.shape[0]
        self.out = torch.empty(self.W, dtype=torch.float64, device=theta_block_device.device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.gathered = (torch.empty(self.W * self.world, dtype=torch.float64, device=theta_block_device.device)
                         if self.world > 1 else self.out)
        self._dist = dist
        self._torch = torch

    def step(self):
        stream = self._torch.cuda.current_stream().cuda_stream
        self.engine.lnprob_device(self.theta.data_ptr(), self.out.data_ptr(), self.W, stream)
        if self.world > 1:
            self._dist.all_gather_into_tensor(self.gathered, self.out, group=self.group)
        return self.gathered
