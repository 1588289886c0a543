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


class OverlappedGather:
    """gather_utterances for a STREAM of steps with equal shards: the all-gather of step i is issued asynchronously (RCCL
    runs it on its own stream, behind the kernels of step i) and is only waited for after step i+1's kernels have been
    queued, so the collective runs under the next step's compute instead of between two steps.  Two result buffers
    alternate; a buffer is reused only after the gather that filled it two steps ago has been waited for.

        og = OverlappedGather(batch)
        for x in stream:
            prev = og.submit(infer(x))   # full [batch, L] of the PREVIOUS step (None on the first call)
        last = og.flush()                # the final step's result; every collective has completed
    """

    def __init__(self, batch: int, group=None):
        self.batch, self.group = batch, group
        self.world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        counts = [hi - lo for lo, hi in (shard_bounds(batch, self.world, r) for r in range(self.world))]
        if min(counts) != max(counts):
            raise ValueError("OverlappedGather needs equal shards (use gather_utterances for ragged batches)")
        self.rows = counts[rank]
        self._buf = [None, None]
        self._work = [None, None]
        self._n = 0

    def submit(self, local: torch.Tensor):
        if local.shape[0] != self.rows:
            raise ValueError(f"expected a shard of {self.rows} utterances, got {local.shape[0]}")
        i = self._n & 1
        if self._work[i] is not None:      # the gather that filled this buffer two steps ago
            self._work[i].wait()
            self._work[i] = None
        shape = (self.batch,) + tuple(local.shape[1:])
        if self._buf[i] is None or self._buf[i].shape != shape or self._buf[i].dtype != local.dtype:
            self._buf[i] = torch.empty(shape, dtype=local.dtype, device=local.device)
        self._work[i] = dist.all_gather_into_tensor(self._buf[i], local.contiguous(), group=self.group, async_op=True)
        self._n += 1
        j = i ^ 1
        if self._n < 2:
            return None
        if self._work[j] is not None:      # the previous step's gather: it ran under this step's compute
            self._work[j].wait()
            self._work[j] = None
        return self._buf[j]

    def flush(self):
        """Wait for every outstanding gather; returns the last submitted step's result (None if nothing was submitted)."""
        for k in (0, 1):
            if self._work[k] is not None:
                self._work[k].wait()
                self._work[k] = None
        return None if self._n == 0 else self._buf[(self._n - 1) & 1]
