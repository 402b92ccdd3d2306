"""Sharding of independent MPC instances over the GPUs of one node (one process per GPU).

The path has NO exchange step: an instance depends only on its own inputs and on the shared
parameter block (REF:22-48), so a rank solves a contiguous slice of the batch and nothing else.
What this module adds around that is plumbing over `torch.distributed` (backend "nccl" = RCCL over
xGMI on MI355X, "gloo" in the CPU tests):
  * `broadcast_params`  -- rank `src`'s parameter block to every rank (~0.6 KB),
  * `shard_bounds`      -- the slice a rank owns,
  * `gather_controls`   -- optional collection of the per-rank results with ONE
                           all_gather of equal-sized shards (direct peer copies; at ~1 KB/instance
                           ring-vs-tree is irrelevant).
"""
from __future__ import annotations

import ctypes

import numpy as np


def shard_bounds(total, rank, world):
    """[lo, hi) of the contiguous shard of `rank`; shards differ by at most one instance."""
    total, rank, world = int(total), int(rank), int(world)
    if not 0 <= rank < world:
        raise ValueError("rank outside [0, world)")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_params(cparams, src=0, group=None, device=None):
    """In-place broadcast of a `bmpc_params` struct from `src`.  Returns it."""
    import torch
    import torch.distributed as dist
    n = ctypes.sizeof(cparams)
    buf = np.frombuffer(ctypes.string_at(ctypes.addressof(cparams), n), dtype=np.uint8).copy()
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src, group=group)
    out = t.cpu().numpy().tobytes()
    ctypes.memmove(ctypes.addressof(cparams), out, n)
    return cparams


def solve_sharded(solve_fn, inputs, total=None, rank=None, world=None):
    """Run `solve_fn(**shard)` on this rank's slice of every array in `inputs` (dict name -> array
    with leading batch dimension, or None).  Returns (lo, hi, result)."""
    import torch.distributed as dist
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    if total is None:
        total = next(len(v) for v in inputs.values() if v is not None)
    lo, hi = shard_bounds(total, rank, world)
    shard = {k: (None if v is None else v[lo:hi]) for k, v in inputs.items()}
    return lo, hi, solve_fn(**shard)


def gather_controls(local, total, group=None):
    """All ranks receive the full (total, ...) tensor from per-rank shards `local` (torch tensor
    (hi-lo, ...)).  Shards are padded to the largest shard so a single all_gather suffices."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = -(-int(total) // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        pieces.append(out[r * per:r * per + (hi - lo)])
    return torch.cat(pieces, dim=0)
