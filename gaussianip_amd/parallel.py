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


def _copy_back(dsts, srcs):
    if hasattr(torch, "_foreach_copy_"):
        torch._foreach_copy_(list(dsts), list(srcs))          # one multi-tensor kernel instead of one per tensor
    else:
        for d, s_ in zip(dsts, srcs):
            d.copy_(s_)


def exchange_sum(params: Sequence[torch.Tensor], viewspace_grad_norm: Optional[torch.Tensor] = None, group=None,
                 average: bool = False) -> Optional[torch.Tensor]:
    """all_reduce(sum) of `p.grad` for all params and of `viewspace_grad_norm` [P] through ONE flat bucket, in place
    (pack = one cat kernel, unpack = one multi-tensor copy).  `average` divides the gradients (not the norms) by the
    world size."""
    if not _on(group):
        return viewspace_grad_norm
    grads = [p.grad for p in params if p.grad is not None]
    dsts = list(grads)
    if viewspace_grad_norm is not None:
        dsts.append(viewspace_grad_norm)
    if not dsts:
        return viewspace_grad_norm
    dtype = grads[0].dtype if grads else viewspace_grad_norm.dtype
    flat = torch.cat([d.reshape(-1).to(dtype) for d in dsts])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average and grads:
        n_grad = sum(g.numel() for g in grads)
        flat[:n_grad].mul_(1.0 / dist.get_world_size(group))
    _copy_back(dsts, [c.view_as(d) for d, c in zip(dsts, flat.split([d.numel() for d in dsts]))])
    return viewspace_grad_norm


def allreduce_gradients(params: Sequence[torch.Tensor], group=None, average: bool = False) -> None:
    """In-place all-reduce of `p.grad` for all params through one flat bucket."""
    exchange_sum(params, None, group, average)


class _MaxExchange:
    """Handle of exchange_max: wait() makes the current stream wait for the collective and writes the results back."""

    def __init__(self, work, flat, dsts):
        self.work, self.flat, self.dsts = work, flat, dsts

    def wait(self):
        if self.flat is None:
            return
        if self.work is not None:
            self.work.wait()
        _copy_back([d.reshape(-1).view(torch.int32) for d in self.dsts], self.flat.split([d.numel() for d in self.dsts]))
        self.flat = None


def exchange_max(radii: Optional[torch.Tensor] = None, depth_max: Optional[torch.Tensor] = None, group=None,
                 async_op: bool = False) -> _MaxExchange:
    """all_reduce(max) of `radii` [P] (int32) and `depth_max` (0-d float32) as ONE int32 bucket, in place.  Non-negative
    IEEE floats order like their bit patterns, so the depth maximum rides along as its int32 view — no conversion kernels
    and the result is exact.  Both are non-negative by construction (pixel radii; depths of Gaussians beyond the near
    plane).  With async_op the collective overlaps whatever is enqueued next (the raster backward): call .wait() on the
    returned handle before reading the tensors."""
    dsts = [t for t in (radii, depth_max) if t is not None]
    if not _on(group) or not dsts:
        return _MaxExchange(None, None, [])
    for t in dsts:
        if t.dtype not in (torch.int32, torch.float32) or not t.is_contiguous():
            raise TypeError("exchange_max takes contiguous int32 / float32 tensors")
    flat = torch.cat([t.reshape(-1).view(torch.int32) for t in dsts])
    work = dist.all_reduce(flat, op=dist.ReduceOp.MAX, group=group, async_op=async_op)
    h = _MaxExchange(work if async_op else None, flat, dsts)
    if not async_op:
        h.wait()
    return h


_side_streams = {}


class _ForwardStats:
    def __init__(self, radii, depth_max, side):
        self.radii, self.depth_max, self._side = radii, depth_max, side

    def wait(self):
        """Joins the side stream; returns (radii_max [P] int32, depth_max 0-d float32), global over all ranks."""
        if self._side is not None:
            cur = torch.cuda.current_stream(self.radii.device)
            cur.wait_stream(self._side)
            self.radii.record_stream(cur)          # allocated on the side stream, consumed on the caller's
            self.depth_max.record_stream(cur)
            self._side = None
        return self.radii, self.depth_max


def exchange_forward_stats(radii_per_view: torch.Tensor, depth: torch.Tensor, group=None) -> _ForwardStats:
    """The MAX bucket straight from the forward outputs — radii [V, P] int32 and the depth images — reduced over the
    local views / pixels AND exchanged on a side HIP stream, so that the small reduction / packing kernels and the
    collective's latency all hide under the raster backward that the caller enqueues next.  .wait() before use."""
    if radii_per_view.is_cuda:
        dev = radii_per_view.device
        side = _side_streams.get(dev.index)
        if side is None:
            side = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        radii_per_view.record_stream(side)
        depth.record_stream(side)
        with torch.cuda.stream(side):
            rmax, dmax = radii_per_view.amax(dim=0), depth.detach().amax()
            exchange_max(rmax, dmax, group)
        return _ForwardStats(rmax, dmax, side)
    rmax, dmax = radii_per_view.amax(dim=0), depth.detach().amax()
    exchange_max(rmax, dmax, group)
    return _ForwardStats(rmax, dmax, None)


def exchange_step(params: Sequence[torch.Tensor], viewspace_grad_norm: Optional[torch.Tensor] = None,
                  radii: Optional[torch.Tensor] = None, depth_max: Optional[torch.Tensor] = None, group=None,
                  average: bool = False) -> Dict[str, Optional[torch.Tensor]]:
    """Everything a step exchanges, in TWO collectives (each small all-reduce costs tens of microseconds of latency on
    xGMI, comparable to a raster kernel): one SUM bucket = [parameter gradients | view-space gradient norms]
    (exchange_sum), one MAX bucket = [radii | depth maximum] (exchange_max).  `viewspace_grad_norm` [P] = sum over the
    local views of ||grad_xy||, `radii` [P] = max over the local views, `depth_max` = local depth maximum (0-d tensor).
    All arguments are updated in place, like separate all_reduce calls would.  A caller that has the MAX inputs before
    its backward (they are forward outputs) can start exchange_max(..., async_op=True) there instead and overlap it."""
    if _on(group):
        exchange_max(radii, depth_max, group)
        exchange_sum(params, viewspace_grad_norm, group, average)
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
