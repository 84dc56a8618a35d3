"""Batch sharding over the GPUs of a node: every controller is its own QP, so the batch
splits into contiguous index ranges with NO collective on the data path.

* one process per GPU (bench.py under torch.distributed.run): `shard_range` gives this rank's slice
  of the global batch -- the same rule as `mpcqp_multi_create` in the library -- and `gather` collects
  the per-rank results on every rank with ONE all_gather per array (RCCL over xGMI on the GPU box,
  gloo in the CPU tests);
  `scatter` is the way in when a period's inputs are born on ONE rank (a plant-wide observer, a supervisory layer): ONE
  scatter collective per array hands every rank its slice;
* one process driving several GPUs: `mpcqp.MultiHandle` (mpcqp_multi_* of the C-ABI; mpcqp_multi_scatter_device /
  mpcqp_multi_gather_device move the slices with peer copies over xGMI).
"""
from __future__ import annotations

import numpy as np


def shard_range(B: int, rank: int, world: int) -> tuple[int, int]:
    """(offset, count) of rank's contiguous shard: floor(B/world) each, +1 for the first B mod world."""
    if not 0 <= rank < world or B < world:
        raise ValueError("need 0 <= rank < world <= B")
    base, rem = divmod(B, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def gather(local, B: int, dist, device=None):
    """All ranks' shards of a problem-major array -> the whole-batch array on every rank.

    `local` is this rank's (count, ...) NumPy array or torch tensor (rows = its problems); shards may
    differ by one row, so they are padded to the largest for the single all_gather.  `dist` is
    torch.distributed (initialised); `device` the torch device collectives run on (None = CPU/gloo)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    t = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    cmax = shard_range(B, 0, world)[1]
    pad = torch.zeros((cmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.cat([parts[r][: shard_range(B, r, world)[1]] for r in range(world)])
    return out if isinstance(local, torch.Tensor) else out.cpu().numpy()


def scatter(whole, B: int, dist, src: int = 0, device=None, like=None):
    """Rank `src` holds the whole-batch problem-major array `whole` ((B, ...) NumPy array or torch tensor; ignored on the
    other ranks, which pass `like`: any array / tensor with the trailing shape and dtype); every rank receives its
    contiguous shard (shard_range) with ONE scatter collective (RCCL on the GPU box, gloo in the CPU tests).  Shards
    may differ by one row: they travel padded to the largest."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    ref = whole if rank == src else like
    as_numpy = not isinstance(ref, torch.Tensor)
    t = torch.from_numpy(np.ascontiguousarray(ref)) if as_numpy else ref
    if device is not None:
        t = t.to(device)
    cmax = shard_range(B, 0, world)[1]
    tail = tuple(t.shape[1:])
    recv = torch.empty((cmax,) + tail, dtype=t.dtype, device=t.device)
    parts = None
    if rank == src:
        if t.shape[0] != B:
            raise ValueError(f"whole-batch array has {t.shape[0]} rows, expected {B}")
        parts = []
        for r in range(world):
            o, n = shard_range(B, r, world)
            p = torch.zeros((cmax,) + tail, dtype=t.dtype, device=t.device)
            p[:n] = t[o:o + n]
            parts.append(p)
    dist.scatter(recv, parts, src=src)
    out = recv[: shard_range(B, rank, world)[1]].contiguous()
    return out.cpu().numpy() if as_numpy else out
