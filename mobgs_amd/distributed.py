"""Multi-GPU sharding of the K latent sub-frame renders of one blurry training view (one process per GPU,
torch.distributed over RCCL/xGMI; backend "nccl" IS RCCL on ROCm, "gloo" on CPU for tests).

The reference is single-GPU: train.py:502-541 renders the K = 9 latent sharp frames of a blurry view one after
the other and averages them, `pred = mean_k(render_k) + 1e-10`.  The sub-frames only depend on the (replicated)
Gaussians and on their own camera / exposure offset, so they shard with exactly two exchange steps
(SURVEY.md section 8e):

  forward   all_reduce(SUM) of the rank-local partial image sum  [3,H,W] fp32 (16.4 MB at 1352x1014)
            -> every rank holds the identical blurry prediction and computes the identical loss;
  backward  dL/dpred is already replicated, so the all-reduce back-propagates as the identity (no traffic);
            each rank back-propagates its own sub-frames, then ONE flat all_reduce(SUM) over all parameter
            gradients (<= 57 floats per Gaussian) gives every rank the full gradient.

Full replicas of the parameters live on every rank (300k x 57 floats = 68 MB, trivial next to 288 GB HBM).
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


class _SumAcrossRanks(torch.autograd.Function):
    """y = sum_r x_r (all-reduce).  The caller's loss is a function of y that is IDENTICAL on every rank, so the
    gradient of that single loss w.r.t. this rank's x_r is dL/dy itself: backward is the identity."""

    @staticmethod
    def forward(ctx, x, group):
        y = x.contiguous().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


class SubframeShard:
    def __init__(self, world_size: Optional[int] = None, rank: Optional[int] = None, group=None):
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
        self.world, self.rank, self.group = int(world_size), int(rank), group

    # ---- partition ----------------------------------------------------------------------------------
    def units(self, n_units: int) -> List[int]:
        """Sub-frame indices this rank renders: u mod world == rank (round-robin keeps 9 units over 8 ranks at
        ceil(9/8) = 2 on one rank, 1 elsewhere)."""
        return [u for u in range(n_units) if u % self.world == self.rank]

    def owner(self, unit: int) -> int:
        return unit % self.world

    # ---- forward exchange ---------------------------------------------------------------------------
    def mean_of_subframes(self, local_sum: torch.Tensor, n_units: int) -> torch.Tensor:
        """local_sum = sum of THIS rank's sub-frame renders -> mean over all n_units sub-frames (+1e-10, as
        train.py:541), identical on every rank.  With one process and one unit it is the render itself."""
        if self.world == 1:
            return local_sum if n_units == 1 else local_sum / n_units + 1e-10
        total = _SumAcrossRanks.apply(local_sum, self.group)
        return total / n_units + 1e-10

    def render_blurry_view(self, render_unit: Callable[[int], torch.Tensor], n_units: int,
                           like: Optional[torch.Tensor] = None) -> torch.Tensor:
        """render_unit(k) -> sharp latent image of sub-frame k.  Returns the blurry prediction (all ranks).
        A rank that owns no sub-frame (world > n_units) contributes zeros; its prediction then does not require
        grad, so it must skip loss.backward() but still call all_reduce_gradients()."""
        mine = self.units(n_units)
        local = None
        for k in mine:
            img = render_unit(k)
            local = img if local is None else local + img
        if local is None:  # more ranks than sub-frames
            if like is None:
                raise ValueError("rank without sub-frames needs `like` to know the image shape")
            local = torch.zeros_like(like)
        return self.mean_of_subframes(local, n_units)

    # ---- backward exchange --------------------------------------------------------------------------
    def all_reduce_gradients(self, params: Sequence[torch.Tensor]) -> None:
        """One flat all_reduce(SUM) over the .grad of every parameter (missing grads count as zero)."""
        if self.world == 1:
            return
        params = [p for p in params if p.requires_grad]
        if not params:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32)
                          for p in params])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        # the reduced gradients stay in the flat buffer: every .grad becomes a view of it (no copy back)
        off = 0
        for p in params:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            p.grad = g if p.dtype == torch.float32 else g.to(p.dtype)
            off += n


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s replica."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for p in params:
        dist.broadcast(p.data, src=src, group=group)
