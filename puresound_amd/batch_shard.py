"""Utterance-batch data parallelism: one process per GPU, weights replicated, the batch split on dim 0,
no communication during compute, one all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) to
reassemble the output waveforms.  Replaces the reference's only multi-GPU mechanism,
torch.nn.DataParallel (puresound/task/base.py:226-229: per-call scatter + parameter broadcast + gather).
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `batch` utterances: the first batch % world ranks get one extra."""
    if world <= 0 or not (0 <= rank < world) or batch < 0:
        raise ValueError(f"bad shard request batch={batch} world={world} rank={rank}")
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_utterances(local: torch.Tensor, batch: int, group=None) -> torch.Tensor:
    """All-gather per-rank outputs [b_r, L] into the full [batch, L] on every rank.

    Equal shards use one all_gather_into_tensor (a single RCCL all-gather); ragged shards are padded
    to the largest shard first and trimmed afterwards."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(batch, world, r) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    width = max(counts)
    if min(counts) == width:
        out = torch.empty(batch, *local.shape[1:], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros(width, *local.shape[1:], dtype=local.dtype, device=local.device)
    padded[:counts[rank]] = local
    buf = torch.empty(world * width, *local.shape[1:], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * width:r * width + counts[r]] for r in range(world)], dim=0)


def sharded_inference(infer: Callable[[torch.Tensor], torch.Tensor], noisy: torch.Tensor, group=None) -> torch.Tensor:
    """Run `infer` on this rank's slice of `noisy` [B, L] and return the gathered [B, L_out]."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(noisy.shape[0], world, rank)
    return gather_utterances(infer(noisy[lo:hi]), noisy.shape[0], group)
