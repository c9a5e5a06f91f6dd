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

import contextlib

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


class OverlappedGradAllReduce:
    """Gradient all-reduce driven by post-accumulate-grad hooks, shaped for RCCL over xGMI:

    * a LARGE gradient (>= `big_numel` elements; here the 54 MB SH-rest block, ready right after the rasterizer's
      preprocess-backward kernel) starts its own in-place collective the moment autograd has produced it, so it
      travels while the mesh->Gaussian backward still runs;
    * the SMALL ones (f_dc, opacity, _alpha, _scale, vertices: ~10 MB together) are packed into ONE flat bucket in
      `finish()` and reduced with a single collective -- collectives of one communicator run back to back, so five
      more launches would add their fixed cost to the wire time of the big one; the parameters' `.grad` become views
      of the reduced bucket (no copy back).

    `finish()` makes the compute stream wait for the collectives and applies the 1/world averaging.  Semantics are
    identical to `allreduce_gradients` (tested with gloo, world size 2, in tests/test_ddp_cpu.py).

    Gradient accumulation over several `backward()` calls: run all but the last under `with reducer.no_sync():` (as
    with torch's DistributedDataParallel) so that the collectives start once, on the accumulated gradients."""

    def __init__(self, params: Iterable[torch.Tensor], world: int, average: bool = True, big_numel: int = 1 << 22,
                 force: bool = False):
        self.params = [p for p in params]
        self.world, self.average, self.big_numel = world, average, int(big_numel)
        self._works, self._big, self._small, self._handles = [], [], [], []
        self._enabled = True
        # `force`: run the collectives even in a world of one (exercises the backend on a single GPU)
        if (world > 1 or force) and dist.is_initialized():
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside this context only accumulate `.grad`; no collective is started."""
        prev, self._enabled = self._enabled, False
        try:
            yield
        finally:
            self._enabled = prev

    def _hook(self, p: torch.Tensor) -> None:
        g = p.grad
        if g is None or not self._enabled:
            return
        if g.numel() >= self.big_numel and g.is_contiguous():
            self._big.append(g)
            self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self._small.append(p)

    def finish(self) -> None:
        flat = None
        if self._small:
            grads = [p.grad for p in self._small]
            flat = torch.cat([g.reshape(-1) for g in grads])
            self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for w in self._works:
            w.wait()
        if self.average:
            scaled = self._big + ([flat] if flat is not None else [])
            if scaled:
                torch._foreach_div_(scaled, float(self.world))
        if flat is not None:
            off = 0
            for p, g in zip(self._small, grads):
                n = g.numel()
                p.grad = flat[off:off + n].view(g.shape)
                off += n
        self._works, self._big, self._small = [], [], []

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
