"""View-parallel data parallelism: one process per GPU, one camera view per rank per step, one
gradient all-reduce (RCCL over xGMI; `nccl` backend on ROCm) per step.

New functionality relative to the reference (single process, single GPU: utils/general_utils.py:213,
one camera per iteration: train.py:90-92).  Semantics: the all-reduced gradient equals the sum over
the ranks' single-view gradients divided by world size (mean-loss semantics), SURVEY.md 8(e).

The six parameter gradients (vertices, _alpha, f_dc, f_rest, opacity, scale: 53 floats/Gaussian,
63.6 MB at 300k) are reduced as a few large buckets -- xGMI is point-to-point, so large messages
that RCCL can split over all seven links beat many small ones; the 54 MB f_rest gradient goes
first because the backward produces it last-but-largest and it dominates the wire time.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def shard_views(num_views: int, step: int, rank: int, world: int, seed: int = 0) -> int:
    """Camera index rendered by `rank` at `step`: a shared seeded permutation dealt round-robin
    (replaces the single-camera pop at train.py:90-92; identical on every rank, no communication)."""
    epoch, pos = divmod(step * world + rank, num_views)
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(num_views, generator=g)
    return int(perm[pos])


def allreduce_gradients(params: Iterable[torch.Tensor], world: int, average: bool = True) -> None:
    """Sum (then average) `.grad` of every parameter over all ranks, largest tensors first, async."""
    if world <= 1 or not dist.is_initialized():
        return
    grads: List[torch.Tensor] = [p.grad for p in params if p.grad is not None]
    grads.sort(key=lambda g: -g.numel())
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in grads]
    for w in works:
        w.wait()
    if average:
        torch._foreach_div_(grads, float(world))
