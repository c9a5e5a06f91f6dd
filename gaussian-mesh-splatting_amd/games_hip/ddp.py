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


class DirectAllReduce:
    """Sum all-reduce of ONE large contiguous tensor as two all-to-all phases, shaped for a FULLY CONNECTED xGMI node.

    A ring all-reduce pushes 2 (W-1)/W of the message through one link per GPU, hop after hop (64 MB on 8 GPUs: 112 MB over a
    153 GB/s link, ~0.7 ms).  xGMI gives every GPU a dedicated link to each of its 7 peers, so the textbook direct algorithm
    uses all of them at once:
        phase 1  all-to-all: shard j of my buffer goes to rank j (W-1 transfers of n/W elements, one per link);
                 I then sum the W shards I hold -- my slice of the result -- in one local pass;
        phase 2  all-to-all again: my reduced slice goes to every peer (one per link), theirs come to me.
    Each link carries 2 n/W elements in total instead of 2 (W-1) n/W: 7x less per link on 8 GPUs.  Both phases are
    `all_to_all_single` (grouped point-to-point sends in RCCL).  `start()` launches phase 1 asynchronously (from the autograd
    hook, while the rest of the backward still runs); `finish()` does the local sum and phase 2 and leaves the result in the
    caller's tensor.  Tested against `all_reduce` with gloo, world size 2 and 3, sizes not divisible by the world size."""

    def __init__(self, world: int, group=None, gather: str = "all_to_all"):
        """`gather`: phase 2 as a second all-to-all of the replicated slice ("all_to_all": every link in parallel, one extra
        local pass to replicate the slice) or as `all_gather_into_tensor` ("all_gather": no replication, RCCL's own choice
        of all-gather algorithm)."""
        self.world, self.group, self.gather = world, group, gather
        self._pending = None

    def start(self, t: torch.Tensor):
        assert t.is_contiguous()
        flat = t.view(-1)
        n, W = flat.numel(), self.world
        shard = (n + W - 1) // W
        if shard * W != n:                       # pad to a multiple of the world size (rare: 45 P is divisible by 8 for even P)
            send = torch.zeros(shard * W, dtype=flat.dtype, device=flat.device)
            send[:n] = flat
        else:
            send = flat
        recv = torch.empty_like(send)
        work = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        self._pending = (flat, n, shard, recv, work)

    def finish(self):
        flat, n, shard, recv, work = self._pending
        self._pending = None
        W = self.world
        work.wait()
        mine = recv.view(W, shard).sum(dim=0)                    # my slice of the result
        out = torch.empty(W * shard, dtype=flat.dtype, device=flat.device) if shard * W != n else flat
        if self.gather == "all_gather":
            dist.all_gather_into_tensor(out, mine, group=self.group)
        else:
            dist.all_to_all_single(out, mine.repeat(W), group=self.group)    # my slice to everyone, everyone's to me
        if out is not flat:
            flat.copy_(out[:n])


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
                 force: bool = False, algorithm: str = "ring"):
        """`algorithm`: how a LARGE gradient is reduced -- "ring" = one `all_reduce` (RCCL picks its algorithm), "direct" =
        two all-to-all phases over all xGMI links at once (DirectAllReduce), "direct_ag" = all-to-all + all-gather.  The flat bucket of small gradients always uses
        `all_reduce` (latency bound)."""
        self.params = [p for p in params]
        self.world, self.average, self.big_numel = world, average, int(big_numel)
        self.algorithm = algorithm
        self._force = bool(force)
        self._direct = []
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
            if self.algorithm in ("direct", "direct_ag") and (self.world > 1 or self._force):
                d = DirectAllReduce(self.world, gather="all_gather" if self.algorithm == "direct_ag" else "all_to_all")
                d.start(g)
                self._direct.append(d)
            else:
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
        for d in self._direct:
            d.finish()
        self._direct = []
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


class ShFactorExchange:
    """Multi-view steps without the 54 MB SH-gradient all-reduce (include/gmsplat.h, gms_sh_grad_expand).

    Per view, dL/d(sh coefficients) is an outer product: Y(dir) (x) dL/dcolour, where dir depends only on the Gaussian's
    position and the view's camera centre -- replicated knowledge.  While this exchange is enabled the rasterizer's backward
    writes no SH gradient; it queues the [P+1,3] factor of each view (row P = camera centre).  `start()` all-gathers the
    factors of this rank's views (3.6 MB per view at 300 k Gaussians instead of 57.5 MB of dense SH gradient), `finish()`
    forms  sum over all views of all ranks  Y(dir_v) (x) g_v  on every rank, views in (rank, view) order -- bit-identical
    everywhere -- and stores it as `.grad` of the two feature tensors (accumulating into an existing `.grad`).

    The other gradients (alpha, scale, opacity, vertices: ~6 MB) stay with `OverlappedGradAllReduce`, whose hooks never see
    the feature tensors in this mode (autograd produces no gradient for them).

    `ops` = (set_mode, take, expand) defaults to the HIP path of `diff_gaussian_rasterization`; the CPU tests inject the
    oracle's restatements (tests/test_ddp_cpu.py)."""

    def __init__(self, features_dc: torch.Tensor, features_rest: torch.Tensor, world: int, group=None, force: bool = False, ops=None,
                 average: bool = True):
        """`average` (default True, as `OverlappedGradAllReduce` / `allreduce_gradients`): the SH gradient is the MEAN over the
        ranks (sum over all views of all ranks / world), so that pairing this exchange with `OverlappedGradAllReduce` under
        default arguments gives every parameter the same semantics.  Pass False when the 1/world already rides in the upstream
        gradient (bench.py does that: the collective is then a plain sum)."""
        if ops is None:
            import diff_gaussian_rasterization as dgr
            ops = (dgr.set_sh_factor_mode, dgr.take_sh_factors, dgr.sh_grad_expand)
        self._set_mode, self._take, self._expand = ops
        self.f_dc, self.f_rest = features_dc, features_rest
        self.world, self.group, self.average = int(world), group, bool(average)
        if self.world > 1 and not dist.is_initialized():
            # (round-4 advisor finding: the 1/world of `average` used to be applied although nothing was gathered)
            raise RuntimeError(f"ShFactorExchange(world={self.world}) needs an initialised torch.distributed process group")
        self._comm = (self.world > 1 or force) and dist.is_initialized()
        self._work, self._gathered, self._local = None, None, None
        self._views_checked = None

    def enable(self) -> "ShFactorExchange":
        self._set_mode(True)
        return self

    def watch(self, means3D: torch.Tensor) -> None:
        """Start the gather from inside the backward pass: `means3D` is the (non-leaf) position tensor the rasterizer was
        given; its gradient is complete exactly when the last view's rasterizer backward has returned -- the factors are
        queued by then -- so the all-gather travels while the mesh->Gaussian backward still runs.  Call once per step,
        after the forward of the step's views; without it `start()` must be called after `backward()`."""
        if means3D.requires_grad:
            means3D.register_hook(self._hook)

    def _hook(self, grad):
        if self._gathered is None:
            self.start()
        return None

    def disable(self) -> None:
        self._set_mode(False)

    def start(self) -> None:
        """Right after `backward()`: take the factors this rank's views left behind and start gathering everybody's."""
        facs = self._take()
        if not facs:
            raise RuntimeError("ShFactorExchange.start(): no factor was queued -- was backward() run on the SH path with the mode on?")
        local = facs[0].unsqueeze(0) if len(facs) == 1 else torch.stack(facs)          # [v, P+1, 3]
        self._local = local                       # (kept alive until the gather has completed)
        if self._comm:
            self._views_checked = _check_equal_views(local.shape[0], self._views_checked, self.group, local.device)
            self._gathered = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            self._work = dist.all_gather_into_tensor(self._gathered, local.contiguous(), group=self.group, async_op=True)
        else:
            self._gathered, self._work = local.contiguous(), None

    def finish(self, means3D: torch.Tensor, sh_degree: int, scale: float = None) -> None:
        """Wait for the gather and write the SH gradients.  `scale` multiplies the sum over all views of all ranks; None (the
        default since round 4; it was 1.0 = a plain sum before) = 1/world if `average` and the factors were gathered over the
        ranks, else 1.  Pass `scale=1.0` (or `average=False`) for the old plain-sum behaviour."""
        if scale is None:
            scale = 1.0 / self.world if (self.average and self._comm and self.world > 1) else 1.0
        if self._gathered is None:
            self.start()
        if self._work is not None:
            self._work.wait()
        P = means3D.shape[0]
        dc = torch.empty((P, 1, 3), dtype=torch.float32, device=means3D.device)
        rest = torch.empty((P, self.f_rest.shape[1], 3), dtype=torch.float32, device=means3D.device)
        self._expand(self._gathered, means3D.detach(), int(sh_degree), dc, rest, False)
        if scale != 1.0:
            dc.mul_(scale); rest.mul_(scale)
        for p, g in ((self.f_dc, dc), (self.f_rest, rest)):
            g = g.view(p.shape).to(p.dtype)
            p.grad = g if p.grad is None else p.grad.add_(g)
        self._gathered, self._work, self._local = None, None, None



def _check_equal_views(v: int, checked, group, device):
    """all_gather_into_tensor needs the same number of queued views on every rank; verify it once per view count (one tiny
    collective on the first step and whenever the local count changes) instead of failing inside the collective."""
    if checked == v:
        return checked
    t = torch.tensor([v, -v], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    lo, hi = -int(t[1]), int(t[0])
    if lo != hi:
        raise RuntimeError(f"view-parallel exchange: ranks queued different numbers of views this step (min {lo}, max {hi}); "
                           "every rank must render the same number of views per step")
    return v


class PackedGradExchange:
    """The whole gradient exchange of a view-parallel step as ONE collective.

    Every rank all-gathers one packed buffer  [ its small gradients | the [P+1,3] SH factors of its views ]  and then forms,
    locally and in rank order, (a) the sum of the W small-gradient slices and (b) the SH gradient from the W x views factors
    (`gms_sh_grad_expand`).  Compared with `OverlappedGradAllReduce` + `ShFactorExchange` (one all-reduce + one all-gather):
    one RCCL launch and one stream hand-over per step instead of two, results bit-identical on every rank (fixed summation
    order), at the price of receiving W x 6.6 MB of small gradients instead of all-reducing 6.6 MB (at 8 GPUs and 300 k
    Gaussians 82 MB in, against 36 MB).  Which of the three wins is measured, not assumed: `bench.py --sh-exchange auto`.

    `ops` = (set_mode, take, expand) as for `ShFactorExchange`."""

    def __init__(self, params: Iterable[torch.Tensor], features_dc: torch.Tensor, features_rest: torch.Tensor, world: int, group=None,
                 force: bool = False, ops=None, average: bool = True):
        """`average` (default True, the semantics of `allreduce_gradients` / `OverlappedGradAllReduce`): every gradient -- the
        small ones and the expanded SH gradient -- is the mean over the ranks.  False = plain sums (the caller folded 1/world
        into the upstream gradient, as bench.py does)."""
        if ops is None:
            import diff_gaussian_rasterization as dgr
            ops = (dgr.set_sh_factor_mode, dgr.take_sh_factors, dgr.sh_grad_expand)
        self._set_mode, self._take, self._expand = ops
        self.f_dc, self.f_rest = features_dc, features_rest
        self.small = [p for p in params if p is not features_dc and p is not features_rest]
        self.world, self.group, self.average = int(world), group, bool(average)
        if self.world > 1 and not dist.is_initialized():
            raise RuntimeError(f"PackedGradExchange(world={self.world}) needs an initialised torch.distributed process group")
        self._comm = (self.world > 1 or force) and dist.is_initialized()
        self._views_checked = None

    def enable(self) -> "PackedGradExchange":
        self._set_mode(True)
        return self

    def disable(self) -> None:
        self._set_mode(False)

    def finish(self, means3D: torch.Tensor, sh_degree: int) -> None:
        facs = self._take()
        if not facs:
            raise RuntimeError("PackedGradExchange.finish(): no SH factor was queued -- was backward() run on the SH path with the mode on?")
        local = facs[0].unsqueeze(0) if len(facs) == 1 else torch.stack(facs)          # [v, P+1, 3]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.small]
        send = torch.cat([g.reshape(-1) for g in grads] + [local.reshape(-1)])
        n, nf = send.numel(), local.numel()
        W = self.world if self._comm else 1
        if self._comm:
            self._views_checked = _check_equal_views(local.shape[0], self._views_checked, self.group, local.device)
            gathered = torch.empty(W * n, dtype=send.dtype, device=send.device)
            dist.all_gather_into_tensor(gathered, send, group=self.group)
        else:
            gathered = send
        G = gathered.view(W, n)
        flat = G[:, :n - nf].sum(dim=0) if W > 1 else G[0, :n - nf]
        inv = 1.0 / self.world if (self.average and self._comm and self.world > 1) else 1.0
        if inv != 1.0:
            flat = flat * inv
        off = 0
        for p, g in zip(self.small, grads):
            p.grad = flat[off:off + g.numel()].view(g.shape)
            off += g.numel()
        factors = G[:, n - nf:].reshape((W * local.shape[0],) + tuple(local.shape[1:])).contiguous()     # (rank, view) order
        P = means3D.shape[0]
        dc = torch.empty((P, 1, 3), dtype=torch.float32, device=means3D.device)
        rest = torch.empty((P, self.f_rest.shape[1], 3), dtype=torch.float32, device=means3D.device)
        self._expand(factors, means3D.detach(), int(sh_degree), dc, rest, False)
        if inv != 1.0:
            dc.mul_(inv); rest.mul_(inv)
        for p, g in ((self.f_dc, dc), (self.f_rest, rest)):
            g = g.view(p.shape).to(p.dtype)
            p.grad = g if p.grad is None else p.grad.add_(g)
