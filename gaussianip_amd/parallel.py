"""Per-step exchange for view-sharded data parallelism (one process per GPU, RCCL over xGMI; gloo on CPU in tests).

The reference is single-GPU (launch.py:77,131-138).  The hot path shards by views and seeds: every
rank renders its own cameras of the replicated Gaussian state; per step the ranks exchange
  * the parameter gradients            — one flat bucket, all_reduce(sum): 14·P floats at SH degree 0 (5.6 MB @100k)
  * the densification statistics       — all_reduce(sum) of the per-Gaussian view-space gradient norms,
                                         all_reduce(max) of the radii (GaussianIP.py:452-457, gaussian_model.py:420-422)
  * the depth normaliser               — all_reduce(max) of one float (GaussianIP.py:225 uses the batch-global max)
xGMI is point-to-point (7 links x ~153 GB/s): at these sizes a ring all-reduce is latency-bound (~60-100 us), so the
whole exchange is TWO collectives (one SUM bucket, one MAX bucket) rather than nine small ones.
"""
from typing import Dict, Optional, Sequence

import torch
import torch.distributed as dist


def _on(group):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def allreduce_gradients(params: Sequence[torch.Tensor], group=None, average: bool = False) -> None:
    """In-place all-reduce of `p.grad` for all params through one flat bucket."""
    if not _on(group):
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def exchange_step(params: Sequence[torch.Tensor], viewspace_grad_norm: Optional[torch.Tensor] = None,
                  radii: Optional[torch.Tensor] = None, depth_max: Optional[torch.Tensor] = None, group=None,
                  average: bool = False) -> Dict[str, Optional[torch.Tensor]]:
    """Everything a step exchanges, in TWO collectives (each small all-reduce costs tens of microseconds of latency on
    xGMI, comparable to a raster kernel): one SUM bucket = [parameter gradients | view-space gradient norms], one MAX
    bucket = [radii | depth maximum].  `viewspace_grad_norm` [P] = sum over the local views of ||grad_xy||,
    `radii` [P] = max over the local views (int32: exact in float32 below 2^24), `depth_max` = local depth maximum
    (0-d tensor).  All arguments are updated in place, like separate all_reduce calls would."""
    if not _on(group):
        return {"viewspace_grad_norm": viewspace_grad_norm, "radii": radii, "depth_max": depth_max}
    grads = [p.grad for p in params if p.grad is not None]
    world = dist.get_world_size(group)
    sums = [g.reshape(-1) for g in grads]
    if viewspace_grad_norm is not None:
        sums.append(viewspace_grad_norm.reshape(-1).to(grads[0].dtype if grads else viewspace_grad_norm.dtype))
    if sums:
        flat = torch.cat(sums)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            if average:
                g /= world
            off += n
        if viewspace_grad_norm is not None:
            viewspace_grad_norm.copy_(flat[off:off + viewspace_grad_norm.numel()].view_as(viewspace_grad_norm))
    maxes = []
    if radii is not None:
        maxes.append(radii.reshape(-1).to(torch.float32))
    if depth_max is not None:
        maxes.append(depth_max.reshape(-1).to(torch.float32))
    if maxes:
        flat = torch.cat(maxes)
        dist.all_reduce(flat, op=dist.ReduceOp.MAX, group=group)
        off = 0
        if radii is not None:
            radii.copy_(flat[:radii.numel()].view_as(radii).to(radii.dtype))
            off = radii.numel()
        if depth_max is not None:
            depth_max.copy_(flat[off:off + depth_max.numel()].view_as(depth_max).to(depth_max.dtype))
    return {"viewspace_grad_norm": viewspace_grad_norm, "radii": radii, "depth_max": depth_max}


def broadcast_gaussians(model, src: int = 0, group=None) -> None:
    """Broadcast the six parameter tensors (and the densification statistics) from `src`; used after a densify /
    prune executed on one rank, or at start-up.  Shapes must already agree (broadcast the new P first if they may not)."""
    if not _on(group):
        return
    for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "max_radii2D",
                 "xyz_gradient_accum", "denom"):
        t = getattr(model, name)
        dist.broadcast(t.data if isinstance(t, torch.nn.Parameter) else t, src=src, group=group)


def shard_views(n_views: int, rank: int, world_size: int):
    """Indices of the views rank `rank` renders: views are dealt round-robin; with more ranks than views the extra
    ranks replicate views for a different seed (SURVEY.md §8e '4 views x 2 seeds')."""
    if world_size <= n_views:
        return list(range(rank, n_views, world_size))
    return [rank % n_views]
