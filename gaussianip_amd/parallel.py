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


def _bucket_fast_path(dsts, extra):
    ts = list(dsts) + ([extra] if extra is not None else [])
    return bool(ts) and len(dsts) <= 12 and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts)


def exchange_sum(params: Sequence[torch.Tensor], viewspace_grad_norm: Optional[torch.Tensor] = None, group=None,
                 average: bool = False, viewspace_grads: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """all_reduce(sum) of `p.grad` for all params and of the per-Gaussian view-space gradient norms through ONE flat
    bucket.  The norms come either ready-made (`viewspace_grad_norm` [P], updated in place) or as the raw means2D
    gradients of the local views (`viewspace_grads` [V, P, 3]): then sum_v |grad_xy| is computed by the packing kernel
    and the reduced [P] tensor is returned.  `average` divides the gradients (not the norms) by the world size.
    On the GPU the bucket is packed and unpacked by one launch each (gip_pack_bucket / gip_unpack_bucket)."""
    if not _on(group):
        if viewspace_grads is not None:
            return torch.linalg.vector_norm(viewspace_grads[..., :2], dim=-1).sum(0)
        return viewspace_grad_norm
    grads = [p.grad for p in params if p.grad is not None]
    world = dist.get_world_size(group)
    if _bucket_fast_path(grads + ([viewspace_grad_norm] if viewspace_grad_norm is not None else []), viewspace_grads):
        import ctypes
        from . import _lib
        lib = _lib.model_lib()
        dsts = grads + ([viewspace_grad_norm] if viewspace_grad_norm is not None else [])
        n_scaled = len(grads)
        counts = [d.numel() for d in dsts]
        P = int(viewspace_grads.shape[1]) if viewspace_grads is not None else 0
        V = int(viewspace_grads.shape[0]) if viewspace_grads is not None else 0
        dev = (dsts[0] if dsts else viewspace_grads).device
        flat = torch.empty(sum(counts) + P, device=dev, dtype=torch.float32)
        seg = (ctypes.c_void_p * max(len(dsts), 1))(*[d.data_ptr() for d in dsts])
        cnt = (ctypes.c_int64 * max(len(dsts), 1))(*counts)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        g2d = ctypes.c_void_p(viewspace_grads.data_ptr()) if viewspace_grads is not None else ctypes.c_void_p(None)
        rc = lib.gip_pack_bucket(seg, cnt, len(dsts), g2d, V, P, ctypes.c_void_p(flat.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError("gip_pack_bucket failed with status %d" % rc)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        # gradients (scaled when averaged) and a ready-made norm tensor (never scaled) go back in place
        if average and n_scaled < len(dsts):
            rc = lib.gip_unpack_bucket(seg, cnt, n_scaled, ctypes.c_void_p(dsts[-1].data_ptr()), counts[-1],
                                       ctypes.c_void_p(flat.data_ptr()), 1.0 / world, stream)
        else:
            rc = lib.gip_unpack_bucket(seg, cnt, len(dsts), ctypes.c_void_p(None), 0, ctypes.c_void_p(flat.data_ptr()),
                                       1.0 / world if average else 1.0, stream)
        if rc != 0:
            raise RuntimeError("gip_unpack_bucket failed with status %d" % rc)
        if viewspace_grads is not None:
            return flat[sum(counts):]
        return viewspace_grad_norm
    if viewspace_grads is not None:
        viewspace_grad_norm = torch.linalg.vector_norm(viewspace_grads[..., :2], dim=-1).sum(0)
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
        flat[:n_grad].mul_(1.0 / world)
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


class _ForwardStats:
    def __init__(self, radii, depth_max, work):
        self.radii, self.depth_max, self._work = radii, depth_max, work

    def wait(self):
        """Returns (radii_max [P] int32, depth_max 0-d float32), global over all ranks."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self.radii, self.depth_max


def exchange_forward_stats(radii_per_view: torch.Tensor, depth: torch.Tensor, group=None) -> _ForwardStats:
    """The MAX bucket straight from the forward outputs — radii [V, P] int32 and the depth images (>= 0).  On the GPU one
    kernel (gip_max_bucket) reduces both into an int32 bucket [P + 1] and the all-reduce is started asynchronously, so its
    latency hides under the raster backward that the caller enqueues next; the results are views of the bucket (no
    unpacking).  .wait() before use."""
    if radii_per_view.is_cuda and radii_per_view.dtype == torch.int32 and radii_per_view.is_contiguous() and \
            depth.dtype == torch.float32 and depth.is_contiguous():
        import ctypes
        from . import _lib
        V, P = int(radii_per_view.shape[0]), int(radii_per_view.shape[1])
        bucket = torch.empty(P + 1, device=radii_per_view.device, dtype=torch.int32)
        rc = _lib.model_lib().gip_max_bucket(ctypes.c_void_p(radii_per_view.data_ptr()), V, P, ctypes.c_void_p(depth.data_ptr()),
                                             depth.numel(), ctypes.c_void_p(bucket.data_ptr()),
                                             ctypes.c_void_p(torch.cuda.current_stream(bucket.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_max_bucket failed with status %d" % rc)
        work = dist.all_reduce(bucket, op=dist.ReduceOp.MAX, group=group, async_op=True) if _on(group) else None
        return _ForwardStats(bucket[:P], bucket[P:].view(torch.float32).reshape(()), work)
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


class _GroupMax(torch.autograd.Function):
    """max over the ranks of a group of a non-negative 0-d float32 tensor, differentiable like torch.max over the
    concatenated data: the gradient of everything that used the maximum (on EVERY rank) flows to ONE element — the one on
    the rank that holds the maximum.  TIES: the backward of the full reduction torch.max() / amax() spreads the gradient
    EVENLY among tied elements, so a single-process step splits it between the tied pixels; here, when several RANKS tie
    (identical views, a saturated depth), the whole gradient goes to the lowest rank of the group (which splits it among
    its own tied pixels as usual) — a documented deviation that matters only for exactly equal float maxima on different
    ranks (tests/test_distributed_cpu.py pins it).  The key's ordering needs a non-negative value (depths of Gaussians beyond
    the near plane; an empty render gives 0); a NaN local maximum makes the group maximum NaN on every rank.  One collective: the ranks MAX-reduce the int64 key
    (float bits << 32) | (world - rank); non-negative IEEE floats order like their bit patterns, so the winning key
    carries the maximum in its high word and the winner's rank in its low word."""

    @staticmethod
    def forward(ctx, local_max, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lm = local_max.detach().to(torch.float32).reshape(1)
        if not lm.is_cuda and bool(lm < 0):            # (on the GPU the check would be a host synchronisation: clamp instead)
            raise ValueError("group maximum of a negative value: the bit-pattern ordering needs a non-negative float")
        # NaN stays observable, like depths.max() of the single-process step: it is encoded as the canonical positive quiet NaN
        # (bits 0x7fc00000), which sorts above every finite non-negative float and +inf, so the group maximum becomes NaN on
        # EVERY rank (and the loss with it) — not another rank's finite maximum.  Negative values cannot occur (depths of
        # Gaussians beyond the near plane; an empty render gives 0) and are clamped.
        lm = torch.where(torch.isnan(lm), torch.full_like(lm, float("nan")), lm.clamp_min(0.0))
        bits = lm.view(torch.int32).to(torch.int64)
        key = (bits << 32) | (world - rank)
        dist.all_reduce(key, op=dist.ReduceOp.MAX, group=group)
        gmax = (key >> 32).to(torch.int32).view(torch.float32).reshape(local_max.shape).to(local_max.dtype)
        ctx.group = group
        ctx.save_for_backward((key & 0xFFFFFFFF) == (world - rank))
        return gmax

    @staticmethod
    def backward(ctx, g):
        (wins,) = ctx.saved_tensors
        total = g.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=ctx.group)
        return total * wins.reshape(total.shape).to(total.dtype), None


class ViewSharding:
    """BASELINE.json configs[3]: the 4 views of an optimizer step sharded over the ranks of one SEED GROUP, whole groups
    replicated for independent seeds (2 GPUs = 2 views each, 4 GPUs = 1 view each, 8 GPUs = 4 views x 2
    seeds; launch.py:80 offsets the seed by the rank).  Ranks [g * size, (g + 1) * size) form seed group g and exchange
    only among themselves (their own process group; different seed groups never communicate in the step).

    Equivalence with the single-process step (tests/test_gpu_sharded_step.py): every rank computes its SHARE of the global
    loss — the loss of its local views times len(views) / n_views, since the reference's loss is a mean over the batch
    (loss_sds / batch_size, ipa_guidance.py:653; the opacity term's .mean(), GaussianIP.py:387) — so that
      * parameter gradients       = SUM over the group  (== average of the unscaled local gradients for equal shards)
      * view-space gradient sum   = SUM over the group of the local sums (GaussianIP.py:451-454)
      * radii                     = MAX over the group   (:165-168, :456)
      * depth normaliser          = MAX over the group   (:225)
    densify / prune then runs on every rank from identical statistics with an identically seeded generator."""

    def __init__(self, n_views: int = 4, rank: Optional[int] = None, world: Optional[int] = None, make_groups: bool = True,
                 group_size: Optional[int] = None):
        """`group_size` (default min(world, n_views): BASELINE configs[3]'s 4 views x 2 seeds at 8 GPUs): ranks per seed group.
        A smaller divisor of n_views trades view sharding for seed replication — 8 GPUs as 4 seed groups of 2 ranks x 2 views
        keep every rank's networks at batch 6 instead of 3 (the one-GPU proxy of bench.py prices both)."""
        on = dist.is_available() and dist.is_initialized()
        self.world = world if world is not None else (dist.get_world_size() if on else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if on else 0)
        self.n_views = n_views
        if group_size is None:
            if self.world > n_views and self.world % n_views != 0:
                # e.g. 6 ranks x 4 views: ranks 4 and 5 would re-render views 0 and 1 inside the one group and count twice
                raise ValueError("ViewSharding: world size %d is neither <= n_views nor a multiple of n_views = %d" % (self.world, n_views))
            group_size = min(self.world, n_views)
        if group_size < 1 or group_size > n_views or self.world % group_size:
            raise ValueError("ViewSharding: group_size %d must be <= n_views = %d and divide the world size %d" % (group_size, n_views, self.world))
        self.group_size = group_size
        self.n_seed_groups = max(1, self.world // self.group_size)
        self.seed_id = self.rank // self.group_size
        self.local_rank = self.rank % self.group_size
        self.views = shard_views(n_views, self.local_rank, self.group_size)
        self.share = len(self.views) / float(n_views)
        self.group = None
        if on and make_groups and self.n_seed_groups > 1 and self.group_size > 1:
            for g in range(self.n_seed_groups):          # every rank creates every group (torch.distributed contract)
                ranks = list(range(g * self.group_size, (g + 1) * self.group_size))
                grp = dist.new_group(ranks)
                if g == self.seed_id:
                    self.group = grp

    @property
    def active(self):
        return self.group_size > 1 and dist.is_available() and dist.is_initialized()

    def depth_max(self, x: torch.Tensor) -> torch.Tensor:
        """Group-wide maximum of the local depth maximum (GaussianIP.py:225), with the reference's gradient: every
        rank's opacity depends on it, and all of that gradient reaches the pixel that holds the maximum."""
        return _GroupMax.apply(x, self.group) if self.active else x

    def exchange(self, stage) -> None:
        """Hook for system.StageOneStep.optimizer_step(exchange=...): call after backward.  The local loss must already be
        the rank's share (StageOneStep does that when `sharding` is set)."""
        vs = stage.viewspace_points.grad.sum(dim=0)
        if self.active:
            params = [g["params"][0] for g in stage.gaussian.optimizer.param_groups]
            exchange_max(stage.radii, None, self.group)
            exchange_sum(params, vs, self.group, average=False)
        stage.viewspace_grad_sum = vs
        # the mask of StageOneStep.forward (hand-region exclusion included), from the group-wide radii
        stage.visibility_filter = stage.visibility(stage.radii) if hasattr(stage, "visibility") else stage.radii > 0
