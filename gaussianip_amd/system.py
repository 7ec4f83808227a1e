"""Stage-1 step glue of the GaussianIP system, restated around the batched renderer.

Reference: threestudio/systems/GaussianIP.py — forward :144-230 (loop over the batch's cameras, stack, running radii
max, opacity := depth / (depth.max() + 1e-5)), training_step :362-395 (loss assembly), on_before_optimizer_step
:446-475 (densification statistics + densify / prune schedule), Adam set-up :569-575.  Config values are those of
configs/exp.yaml:66-75,131-138,163-168.  The Lightning plumbing, prompt processor and OpenPose drawing are the
caller's (out of scope): pose maps and prompt embeddings are inputs here.

One behavioural difference, by design: the cameras of a step are rendered in ONE launch set (`render_views`) instead
of sequentially; per-view results are identical (tests/test_gpu_pipeline.py).
"""
import contextlib
import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .renderer import render_views
from .scene.cameras import Camera

_POSE_STREAM = True      # False: pose maps on the main stream (the same-box A/B of DESIGN §4d; tests flip the attribute)
_FUSED_LOSS = True       # False: the sparsity term as the reference's op chain


class _SparsityTerm(torch.autograd.Function):
    """mean(sqrt((depth / (max(depth) + 1e-5))^2 + 0.01)) of a step's depth maps (GaussianIP.py:225, :377-380) in three launches
    forward and two backward (include/gip_model.h: gip_sparsity_loss_*); the op chain it replaces is ~20 launches on 4 M
    elements."""

    @staticmethod
    def forward(ctx, depth):
        from . import _lib
        lib = _lib.model_lib()
        # a workspace per call (12 KB): it carries the maximum and the tie count to THIS call's backward, whatever runs in between
        ws = torch.empty(lib.gip_sparsity_workspace_bytes() // 4, dtype=torch.float32, device=depth.device)
        rc = lib.gip_sparsity_loss_forward(ctypes.c_void_p(depth.data_ptr()), depth.numel(), ctypes.c_void_p(ws.data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream(depth.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_sparsity_loss_forward failed with status %d" % rc)
        ctx.save_for_backward(depth)
        ctx.ws = ws
        return ws[1]

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        depth, = ctx.saved_tensors
        g_depth = torch.empty_like(depth)
        g = g.to(torch.float32).reshape(1)
        rc = _lib.model_lib().gip_sparsity_loss_backward(
            ctypes.c_void_p(depth.data_ptr()), depth.numel(), ctypes.c_void_p(g.data_ptr()), 1.0, ctypes.c_void_p(ctx.ws.data_ptr()),
            ctypes.c_void_p(g_depth.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(depth.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_sparsity_loss_backward failed with status %d" % rc)
        return g_depth


class _StepOutputs(dict):
    """forward()'s dict; "opacity" = depth / (max + 1e-5) (GaussianIP.py:225-226), "scale" and "visibility_filter" are made when
    somebody reads them: the fused sparsity term works on the depth maps directly and the [B, H, W, 1] quotient is otherwise a
    dead 16 MB tensor per step.  The thunks live in `_lazy`, OFF the mapping: iteration, copies, `{**out}` and `dict(out)` show
    reference keys with tensor values only (a lazy key appears once it has been read; `materialize()` reads them all)."""

    _LAZY_KEYS = ("opacity", "scale", "visibility_filter")

    def __init__(self, mapping, lazy=None):
        super().__init__(mapping)
        self._lazy = dict(lazy or {})

    def __missing__(self, key):
        if key == "opacity" and "dmax" in self._lazy:
            v = self[key] = self["depth"] / (self._lazy["dmax"]() + 1e-5)
            return v
        if key == "scale" and "scale" in self._lazy:           # "scale": the activated scaling (GaussianIP.py:228), made when read
            v = self[key] = self._lazy["scale"]()
            return v
        if key == "visibility_filter" and dict.__contains__(self, "radii"):
            v = self[key] = self["radii"] > 0
            return v
        raise KeyError(key)

    def _available(self, key):
        return (key == "opacity" and "dmax" in self._lazy) or (key == "scale" and "scale" in self._lazy) or \
            (key == "visibility_filter" and dict.__contains__(self, "radii"))

    def __contains__(self, key):
        return dict.__contains__(self, key) or self._available(key)

    def get(self, key, default=None):           # dict.get does not go through __missing__
        return self[key] if key in self else default

    def materialize(self):
        """Every lazy key read once: afterwards this is a plain dict with the reference's keys."""
        for k in self._LAZY_KEYS:
            if self._available(k):
                self[k]
        return self


@dataclass
class StageOneConfig:
    # configs/exp.yaml values
    densify_prune_start_step: int = 200
    densify_prune_end_step: int = 1700
    densify_prune_interval: int = 500
    densify_prune_min_opacity: float = 0.04
    densify_prune_screen_size_threshold: int = 20
    densify_prune_screen_size_threshold_fix_step: int = 1500
    densify_prune_world_size_threshold: float = 0.015
    max_grad: float = 0.0002
    prune_only_start_step: int = 1700
    prune_only_end_step: int = 1900
    prune_only_interval: int = 300
    prune_opacity_threshold: float = 0.04
    prune_world_size_threshold: float = 0.015
    refine_start_step: int = 2400
    cameras_extent: float = 4.0
    lambda_sds: float = 1.0
    lambda_sparsity: float = 1.0
    lambda_opaque: float = 0.0
    disable_hand_densification: bool = False
    hand_radius: float = 0.05
    # refine (VCR) orbit and stage 3, configs/exp.yaml:150-160
    refine_n_views: int = 32
    refine_elevation: float = 17.0
    refine_camera_distance: float = 1.5
    refine_fovy_deg: float = 70.0
    refine_train_bs: int = 4
    lambda_l1: float = 10.0
    lambda_lpips: float = 15.0
    extra: dict = None

    @classmethod
    def from_dict(cls, d: dict) -> "StageOneConfig":
        """Build from the `system` section of the reference's YAML (configs/exp.yaml:122-170): scheduling / threshold keys
        become fields, `loss.{lambda_sds, lambda_sparsity, lambda_opaque}` are lifted, everything else is kept in `extra`."""
        from dataclasses import fields
        names = {f.name for f in fields(cls)} - {"extra"}
        own = {k: v for k, v in d.items() if k in names}
        for k, v in (d.get("loss") or {}).items():
            if k in names:
                own[k] = v
        return cls(**own, extra={k: v for k, v in d.items() if k not in names})


def binary_cross_entropy(inp, target):
    """threestudio/utils/ops.py:295-300 (no clamping of the log like F.binary_cross_entropy)."""
    return -(target * torch.log(inp) + (1 - target) * torch.log(1 - inp)).mean()


class StageOneStep:
    def __init__(self, gaussian, pipe, background: torch.Tensor, cfg: Optional[StageOneConfig] = None,
                 hand_centers: Optional[torch.Tensor] = None, skeleton=None, pose_height: int = 512, pose_width: int = 512,
                 head_offset: float = 0.65):
        """`skeleton`: a gaussianip_amd.poser.Skeleton; when given and the batch carries `mvp_mtx`, forward() also
        draws the ControlNet pose maps of all views (one HIP launch) and returns `pose` / `all_vis_all` like
        GaussianIP.forward (:175-196, 218-222); `pose_height/width` = cfg.height / cfg.width (512, configs/exp.yaml)."""
        self.gaussian, self.pipe, self.background = gaussian, pipe, background
        self.cfg = cfg or StageOneConfig()
        self.hand_centers = hand_centers
        self.skeleton, self.pose_hw, self.head_offset = skeleton, (pose_height, pose_width), head_offset
        self.viewspace_points = None
        self.viewspace_grad_sum = None      # [P,3], set by a multi-GPU exchange hook (sum over ALL ranks' views)
        self.depth_max_reduce = None        # callable(0-d tensor) -> all-reduced maximum, set for view-sharded runs
        self.sharding = None                # parallel.ViewSharding: this rank renders batch views sharding.views only
        self.radii = None
        self.visibility_filter = None

    _index_cache = None

    def _take(self, t, ids):
        """t[ids] along dim 0 without a host synchronisation (index tensors are cached per device)."""
        if self._index_cache is None:
            self._index_cache = {}
        key = (t.device, tuple(ids))
        idx = self._index_cache.get(key)
        if idx is None:
            idx = self._index_cache[key] = torch.as_tensor(list(ids), dtype=torch.long).to(t.device)
        return t.index_select(0, idx)

    def visibility(self, radii: torch.Tensor) -> torch.Tensor:
        """The densification mask of GaussianIP.forward (:212-216): radii > 0, minus the Gaussians within `hand_radius` of a
        hand centre when `disable_hand_densification` is set.  One definition for the single-GPU step and for the
        multi-GPU exchange, which re-derives the mask from the group-wide radii maximum."""
        vis = radii > 0
        if self.cfg.disable_hand_densification and self.hand_centers is not None:
            dist = torch.norm(self.gaussian.get_xyz[:, None, :] - self.hand_centers[None, :, :], dim=-1)
            vis = vis & ~(dist.min(dim=-1).values < self.cfg.hand_radius)
        return vis

    # GaussianIP.forward
    def forward(self, batch: Dict, renderbackground=None) -> Dict:
        bg = self.background if renderbackground is None else renderbackground
        B = batch["c2w"].shape[0]
        ids = list(range(B)) if self.sharding is None else list(self.sharding.views)
        cams: List[Camera] = [Camera(c2w=batch["c2w"][i], FoVy=batch["fovy"][i], height=batch["height"], width=batch["width"])
                              for i in ids]
        pose_job = self._start_pose_maps(batch, ids, batch["c2w"].device if torch.is_tensor(batch["c2w"]) else None)
        pkg = render_views(cams, self.gaussian, self.pipe, bg)
        self.viewspace_points = pkg["viewspace_points"]              # [B,P,3]; .grad after backward
        self.viewspace_grad_sum = None
        self.radii = pkg["radii"].max(dim=0).values                  # running max over the views (:165-168)
        self.visibility_filter = self.visibility(self.radii)
        images = pkg["render"].permute(0, 2, 3, 1)                   # [B,H,W,3]
        depths = pkg["depth_3dgs"].permute(0, 2, 3, 1)               # [B,H,W,1]
        local_max = not (self.sharding is not None and self.sharding.active) and self.depth_max_reduce is None
        if local_max and _FUSED_LOSS and depths.is_cuda and depths.dtype == torch.float32 and pkg["depth_3dgs"].is_contiguous():
            # one process, no exchange: the loss takes the sparsity term straight from the depth maps (loss()); the quotient
            # itself is only materialised if somebody asks for it
            out = _StepOutputs({**pkg, "comp_rgb": images, "depth": depths},
                               lazy={"dmax": lambda d_=depths: d_.max(), "scale": lambda g_=self.gaussian: g_.get_scaling})
        else:
            dmax = depths.max()                                      # batch-global maximum (:225)
            if self.sharding is not None and self.sharding.active:
                dmax = self.sharding.depth_max(dmax)                 # over all ranks' views, differentiable
            elif self.depth_max_reduce is not None:
                # replicated batches: the maximum over ALL ranks' views; its gradient flows on the rank that holds it
                gmax = self.depth_max_reduce(dmax.detach().clone())
                dmax = torch.where(dmax.detach() == gmax, dmax, gmax)
            out = {**pkg, "visibility_filter": pkg["visibility_filter"], "comp_rgb": images, "depth": depths,
                   "opacity": depths / (dmax + 1e-5), "scale": self.gaussian.get_scaling}
        if pose_job is not None:
            out["pose"], out["all_vis_all"] = self._finish_pose_maps(pose_job, images.device)
        return out

    # the ControlNet pose maps depend on the batch only (GaussianIP.forward :175-196): they are drawn on a second HIP stream while
    # the rasterizer runs (the span kernel is one wave per limb, 0.15 ms of latency that needs 68 of the chip's 1024 SIMDs)
    _pose_stream = None

    def _start_pose_maps(self, batch, ids, _unused):
        if self.skeleton is None or "mvp_mtx" not in batch:
            return None
        dev = self.skeleton.device
        az, cent, mvp = torch.as_tensor(batch["azimuth"]), torch.as_tensor(batch["center"]), batch["mvp_mtx"]
        if self.sharding is not None:           # (indexing with a Python list would synchronise: cached index tensors)
            az, cent, mvp = self._take(az, ids), self._take(cent, ids), self._take(mvp, ids)
        side = None
        if dev.type == "cuda" and _POSE_STREAM and not torch.cuda.is_current_stream_capturing():
            if self._pose_stream is None:
                self._pose_stream = torch.cuda.Stream(device=dev)
            side = self._pose_stream
            side.wait_stream(torch.cuda.current_stream(dev))        # device-side batch tensors were produced on the main stream
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if not (mvp.device.type == "cpu" and az.device.type == "cpu" and cent.device.type == "cpu"):
                az, cent = az.to(dev, non_blocking=True), cent.to(dev, non_blocking=True)       # device-side batch: stay on the device
            head_zoom = (cent == self.head_offset) & (az > 0)        # :176
            pose, all_vis, _ = self.skeleton.openpose_draw(mvp, self.pose_hw[0], self.pose_hw[1], az, head_zoom, True)
        return pose, all_vis, side

    def _finish_pose_maps(self, job, dev):
        pose, all_vis, side = job
        if side is not None:
            main = torch.cuda.current_stream(pose.device)
            main.wait_stream(side)
            for t_ in (pose, all_vis):
                if torch.is_tensor(t_) and t_.is_cuda:
                    t_.record_stream(main)                           # allocated on the side stream, consumed on the main one
        return pose, all_vis

    # GaussianIP.training_step (stage 1 branch, :362-395)
    def training_step(self, step: int, batch: Dict, guidance, prompt_utils, use_pose_controlnet: bool = True):
        """update_learning_rate -> forward (render + pose maps) -> guidance(...) -> loss.  `prompt_utils` = prompt_processor()."""
        self.gaussian.update_learning_rate(step)
        out = self.forward(batch)
        per_view = {k: v for k, v in batch.items() if k not in ("height", "width")}
        if self.sharding is not None:       # this rank's views only; the loss below is its share of the batch mean
            ids = list(self.sharding.views)
            per_view = {k: (self._take(v, ids) if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == batch["c2w"].shape[0] else v)
                        for k, v in per_view.items()}
        guidance_out = guidance(step, out["comp_rgb"], out["pose"], prompt_utils, use_pose_controlnet, out["all_vis_all"], **per_view)
        loss = self.loss(out, guidance_out)
        if self.sharding is not None:
            loss = loss * self.sharding.share
        return loss, out, guidance_out

    def optimizer_step(self, loss, step: int, scaler=None, exchange=None) -> Optional[str]:
        """backward -> (unscale) -> on_before_optimizer_step -> optimizer.step, in Lightning's order.

        `scaler`: a torch.amp.GradScaler reproduces the reference's `precision: 16-mixed` (configs/exp.yaml:193): the loss
        is multiplied by the scaler's scale before backward and the PARAMETER gradients are unscaled before the hook —
        but `viewspace_points.grad` is a retained non-leaf gradient that the scaler never sees, so the densification
        statistics (and the `max_grad` = 0.0002 threshold, GaussianIP.py:452-462) see gradients that are still
        multiplied by the scale (65536 at the start).  Reference behaviour, reproduced rather than fixed;
        with scaler=None the statistics are unscaled.  `exchange(self)`: multi-GPU gradient / statistics exchange.  It
        runs on the still-SCALED gradients, before `unscale_`: the scaler's inf / NaN check then sees the group-wide sums,
        so every rank of a seed group reaches the same `found_inf` verdict, skips (or takes) the same Adam step and
        keeps the same scale (an overflow on one rank's local gradients would otherwise poison the other ranks' step
        through the SUM while only that rank skips it)."""
        opt = self.gaussian.optimizer
        opt.zero_grad(set_to_none=True)
        if scaler is not None:
            scaler.scale(loss).backward()
        else:
            loss.backward()
        if exchange is not None:
            exchange(self)
        if scaler is not None:
            scaler.unscale_(opt)
        action = self.on_before_optimizer_step(step)
        if scaler is not None:
            scaler.step(opt)
            scaler.update()
        else:
            opt.step()
        return action

    # GaussianIP.training_step (loss assembly)
    def loss(self, out: Dict, guidance_out: Dict) -> torch.Tensor:
        c = self.cfg
        loss = guidance_out["loss_sds"] * c.lambda_sds
        if isinstance(out, _StepOutputs) and not dict.__contains__(out, "opacity") and torch.is_grad_enabled():
            loss = loss + _SparsityTerm.apply(out["depth_3dgs"]) * c.lambda_sparsity
        else:
            loss = loss + (out["opacity"] ** 2 + 0.01).sqrt().mean() * c.lambda_sparsity
        if c.lambda_opaque != 0:
            oc = out["opacity"].clamp(1.0e-3, 1.0 - 1.0e-3)
            loss = loss + binary_cross_entropy(oc, oc) * c.lambda_opaque
        return loss

    def _fused_stats(self, vis) -> bool:
        """accumulate() below as one launch (include/gip_model.h: gip_densify_stats) when everything lives on the GPU in the stock
        dtypes; False = not applicable, the op chain runs."""
        g = self.gaussian
        vg = self.viewspace_grad_sum if self.viewspace_grad_sum is not None else self.viewspace_points.grad
        if not (_FUSED_LOSS and vg is not None and vg.is_cuda and vg.dtype == torch.float32 and vg.is_contiguous() and vis.dtype == torch.bool and
                vis.is_contiguous() and self.radii.dtype == torch.int32 and self.radii.is_contiguous() and
                all(t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda for t in (g.max_radii2D, g.xyz_gradient_accum, g.denom))):
            return False
        from . import _lib
        P = vis.shape[0]
        V = vg.shape[0] if vg.dim() == 3 else 1
        p_ = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        rc = _lib.model_lib().gip_densify_stats(p_(vg), V, P, p_(vis), p_(self.radii), p_(g.max_radii2D), p_(g.xyz_gradient_accum), p_(g.denom),
                                                ctypes.c_void_p(torch.cuda.current_stream(vis.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_densify_stats failed with status %d" % rc)
        return True

    # GaussianIP.on_before_optimizer_step (stage 1 branch)
    @torch.no_grad()
    def on_before_optimizer_step(self, step: int) -> Optional[str]:
        c, g = self.cfg, self.gaussian
        if step >= c.refine_start_step:
            return None
        action = None

        def accumulate():
            # sum of the per-view grads (:451-454); a multi-GPU exchange leaves the all-reduced sum in viewspace_grad_sum
            vis = self.visibility_filter
            if self._fused_stats(vis):
                return
            grad = self.viewspace_grad_sum if self.viewspace_grad_sum is not None else self.viewspace_points.grad.sum(dim=0)
            # same values as the reference's masked assignment (:456), without nonzero() = without a host synchronisation
            g.max_radii2D = torch.where(vis, torch.max(g.max_radii2D, self.radii.to(g.max_radii2D.dtype)), g.max_radii2D)
            g.add_densification_stats(grad, vis)

        if step < c.densify_prune_end_step:
            accumulate()
            min_opacity = c.densify_prune_min_opacity if step > 1900 else 0.05
            if step > c.densify_prune_start_step and step % c.densify_prune_interval == 0:
                screen = c.densify_prune_screen_size_threshold if step > c.densify_prune_screen_size_threshold_fix_step else None
                g.densify_and_prune(c.max_grad, min_opacity, c.cameras_extent, screen, c.densify_prune_world_size_threshold)
                action = "densify_and_prune"
        if c.prune_only_start_step < step < c.prune_only_end_step:
            accumulate()
            if step % c.prune_only_interval == 0:
                g.prune_only(min_opacity=c.prune_opacity_threshold, max_world_size=c.prune_world_size_threshold)
                action = "prune_only"
        return action


class StageThreeStep:
    """Stage 3 (GaussianIP.py:424-436): RGB-only reconstruction of the refined orbit images.  Renders a random subset of
    the refine cameras, crops [60:890, 220:800], halves the resolution (bilinear) and takes
    lambda_l1 * L1 + lambda_lpips * perceptual against the equally prepared refined images.  `perceptual` is either
    a `guidance.perceptual.LPIPSVGG` (the reference's `lpips.LPIPS(net='vgg')`; the target images' features are then
    computed once and kept on the device) or any callable (a, b) -> distances; None omits the term.  The pretrained
    VGG / LPIPS weights are not shippable, so benchmarks use `LPIPSVGG().init_for_benchmark()`."""

    CROP = (slice(60, 890), slice(220, 800))

    def __init__(self, gaussian, pipe, background, cameras, refined_rgbs=None, view_idx_all=None, lambda_l1=1.0,
                 lambda_lpips=0.0, perceptual=None, train_bs=4, cfg: Optional[StageOneConfig] = None,
                 refined_rgbs_small=None):
        """Either `refined_rgbs` [n, H, W, 3] in refinement order with `view_idx_all` (the refiner's return values), or
        `refined_rgbs_small` [n, 3, h, w] in orbit order as stored in after_refine.pth (`load_after_refine`)."""
        self.gaussian, self.pipe, self.background, self.cameras = gaussian, pipe, background, cameras
        self.lambda_l1, self.lambda_lpips, self.perceptual, self.train_bs = lambda_l1, lambda_lpips, perceptual, train_bs
        self.cfg = cfg or StageOneConfig()
        self.viewspace_points = self.refine_radii = self.refine_visibility_filter = None
        if refined_rgbs_small is not None:
            self.gt_small = refined_rgbs_small.to(gaussian.get_xyz.device)
            self.orbit_ids = list(range(self.gt_small.shape[0]))
        else:
            self.gt_small, order = prepare_refined_targets(refined_rgbs, view_idx_all, self.CROP)
            self.orbit_ids = torch.as_tensor(view_idx_all)[order].tolist()
        self.gt_feats = None
        if perceptual is not None and lambda_lpips and hasattr(perceptual, "target_features"):
            self.gt_feats = perceptual.target_features(self.gt_small, normalize=True)

    def training_step(self, id_list=None, generator=None, step: Optional[int] = None):
        """`step` = the stage's own true_global_step (starts at 0): when given, the position learning rate follows
        `update_learning_rate(step + refine_start_step)` (GaussianIP.py:424)."""
        import random
        import torch.nn.functional as F
        from .renderer import render_views
        if step is not None:
            self.gaussian.update_learning_rate(step + self.cfg.refine_start_step)
        if id_list is None:
            id_list = random.sample(range(len(self.orbit_ids)), min(self.train_bs, len(self.orbit_ids)))
        cams = [self.cameras[self.orbit_ids[i]] for i in id_list]
        pkg = render_views(cams, self.gaussian, self.pipe, self.background)
        # render_refine_rgb bookkeeping (GaussianIP.py:304-313, 343): grad carriers, radii maximum over the views
        self.viewspace_points = pkg["viewspace_points"]
        self.refine_radii = pkg["radii"].max(dim=0).values
        self.refine_visibility_filter = self.refine_radii > 0.0
        img = pkg["render"]
        img = img[:, :, self.CROP[0], self.CROP[1]] if img.shape[2] >= 890 and img.shape[3] >= 800 else img
        small = F.interpolate(img, scale_factor=0.5, mode="bilinear", align_corners=False)
        gt = self.gt_small[torch.as_tensor(id_list, device=self.gt_small.device)]
        loss = self.lambda_l1 * (small - gt).abs().mean()
        if self.perceptual is not None and self.lambda_lpips:
            if self.gt_feats is not None:
                sel = torch.as_tensor(id_list, device=self.gt_small.device)
                d = self.perceptual.distance_to_features(small, [f[sel] for f in self.gt_feats], normalize=True)
            elif hasattr(self.perceptual, "target_features"):
                d = self.perceptual(small, gt, normalize=True)
            else:
                d = self.perceptual(small, gt)
            loss = loss + self.lambda_lpips * d.mean()
        return {"loss": loss, "render_pkg": pkg, "id_list": id_list}

    # GaussianIP.on_before_optimizer_step (stage 3 branch, GaussianIP.py:476-506), quirks included
    @torch.no_grad()
    def on_before_optimizer_step(self, step: int) -> Optional[str]:
        c, g = self.cfg, self.gaussian
        gstep = step + c.refine_start_step
        action = None

        def accumulate():
            grad = self.viewspace_points.grad.sum(dim=0)
            vis = self.refine_visibility_filter
            g.max_radii2D = torch.where(vis, torch.max(g.max_radii2D, self.refine_radii.to(g.max_radii2D.dtype)), g.max_radii2D)
            g.add_densification_stats(grad, vis)

        if gstep < 10000:
            if step == 0:
                # "When stage 3 starts, the loaded gaussians don't have max_radii2D" (:483-485)
                g.max_radii2D = self.refine_radii.to(g.max_radii2D.dtype).clone()
            accumulate()
            if gstep == 2500:
                # the threshold test uses the STAGE's step (100 here), so the screen-size limit is off (:491)
                screen = c.densify_prune_screen_size_threshold if step > c.densify_prune_screen_size_threshold_fix_step else None
                g.densify_and_prune(c.max_grad, 0.05, c.cameras_extent, screen, c.densify_prune_world_size_threshold)
                action = "densify_and_prune"
                return action          # the statistics tensors were rebuilt for the new point count (:492 is the last use)
        if 2500 < gstep < 3000:
            accumulate()               # second accumulation of the same step inside this window, as the reference does
            # operator precedence of the reference's test (:504): step + (refine_start_step % interval) == 0
            if step + c.refine_start_step % c.prune_only_interval == 0:
                g.prune_only(min_opacity=c.prune_opacity_threshold, max_world_size=c.prune_world_size_threshold)
                action = "prune_only"
        return action


def prepare_refined_targets(refined_rgbs, view_idx_all, crop=(slice(60, 890), slice(220, 800))):
    """refine.py:307-311: refinement order -> orbit order (idx_mapper = argsort(view_idx_all)), NCHW, crop
    [60:890, 220:800], bilinear half resolution.  Returns (refined_rgbs_small [n,3,h,w], order)."""
    import torch.nn.functional as F
    order = torch.argsort(torch.as_tensor(view_idx_all))
    gt = refined_rgbs[order.to(refined_rgbs.device)].permute(0, 3, 1, 2)
    gt = gt[:, :, crop[0], crop[1]] if gt.shape[2] >= 890 and gt.shape[3] >= 800 else gt
    return F.interpolate(gt, scale_factor=0.5, mode="bilinear", align_corners=False), order


# ---- hand-off files between the stages (GaussianIP.py:404, refine.py:271-274, 315, GaussianIP.py:359) ----
def save_before_refine(path, images, control_images):
    """stage 1 -> stage 2: {'images': [n,H,W,3], 'control_images': [n,H,W,3]} on the CPU."""
    torch.save({"images": images.detach().to("cpu"), "control_images": control_images.detach().to("cpu")}, path)


def load_before_refine(path, device=None):
    d = torch.load(path, map_location="cpu")
    images, control = d["images"], d["control_images"]
    return (images.to(device), control.to(device)) if device is not None else (images, control)


def save_after_refine(path, refined_rgbs, view_idx_all):
    """stage 2 -> stage 3: {'refined_rgbs_small': [n,3,415,290]} (orbit order, cropped, half resolution) on the CPU."""
    small, _ = prepare_refined_targets(refined_rgbs, view_idx_all)
    torch.save({"refined_rgbs_small": small.detach().cpu()}, path)


def load_after_refine(path, device=None):
    small = torch.load(path, map_location="cpu")["refined_rgbs_small"]
    return small.to(device) if device is not None else small


def create_refine_batch(n_views=32, elevation_deg=17.0, camera_distance=1.5, fovy_deg=70.0, height=1024, width=1024):
    """The fixed orbit of the refine stage (GaussianIP.create_refine_batch, GaussianIP.py:232-281): azimuth
    linspace(-180, 180, n + 1)[:n], constant elevation / distance / fovy, look-at the origin with +z up.  Returns the
    reference's dict (c2w [n,4,4], azimuth, elevation, fovy in radians, height, width, center); mvp matrices are left to
    the caller's projection helper, as are the Camera objects (`[Camera(c2w=b["c2w"][i], FoVy=b["fovy"][i], ...)]`)."""
    import math
    import torch.nn.functional as F
    az_deg = torch.linspace(-180.0, 180.0, n_views + 1)[:n_views]
    az = az_deg * math.pi / 180
    el_deg = torch.full_like(az_deg, float(elevation_deg))
    el = el_deg * math.pi / 180
    dist = torch.full_like(az_deg, float(camera_distance))
    pos = torch.stack([dist * torch.cos(el) * torch.cos(az), dist * torch.cos(el) * torch.sin(az), dist * torch.sin(el)], dim=-1)
    center = torch.zeros_like(pos)
    up = torch.as_tensor([0.0, 0.0, 1.0])[None, :].repeat(n_views, 1)
    lookat = F.normalize(center - pos, dim=-1)
    right = F.normalize(torch.cross(lookat, up, dim=-1), dim=-1)
    up = F.normalize(torch.cross(right, lookat, dim=-1), dim=-1)
    c2w3x4 = torch.cat([torch.stack([right, up, -lookat], dim=-1), pos[:, :, None]], dim=-1)
    c2w = torch.cat([c2w3x4, torch.zeros_like(c2w3x4[:, :1])], dim=1)
    c2w[:, 3, 3] = 1.0
    fovy = torch.full_like(az_deg, float(fovy_deg)) * math.pi / 180
    return {"c2w": c2w, "center": center[:, 2], "elevation": el_deg, "azimuth": az_deg, "height": height, "width": width, "fovy": fovy}
